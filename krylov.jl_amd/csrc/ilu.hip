// ilu.hip -- ILU(0) / IC(0) preconditioner resident on the device: factorisation and the two sparse
// triangular solves of M^{-1} = U^{-1} L^{-1}  (SURVEY.md section 8f, row N1).
//
// What it replaces: the reference has no such code of its own -- its GPU examples and tests build the
// preconditioner with the vendor library (`ic02` / `ilu02` + `ldiv!` on triangular views,
// docs/src/gpu.md:74-163, test/gpu/nvidia.jl:37-100: IC(0)-CG on sparse_laplacian(16) must converge in
// <= 19 iterations) and hand it to cg!/bicgstab!/gmres! as the `M` / `N` operator (src/cg.jl:160,241).
// Here the same operator is a khip_operator (khip_ilu0_create) on the pattern of a khip_csr.
//
// Algorithm: ILU(0), IKJ variant (Saad, alg. 10.4); for SPD A its factors are L and U = D L^T, i.e. the
// IC(0) preconditioner in exact arithmetic.  Parallelism: LEVEL SCHEDULING.  Row i of the lower solve
// depends on the rows j < i of its pattern; level(i) = 1 + max level(j).  Rows of one level are
// independent: one kernel launch per level, one lane per row, and every lane walks its row in stored order
// with one rounded multiply and one rounded subtract per entry -- the serial CPU loop, so factors and
// solves are BIT-IDENTICAL to the serial restatement the tests compare against (ko_ilu0 / ko_ilu0_solve).
// The level analysis (setup, once per pattern) runs on the host from the downloaded index arrays;
// the whole apply (2 x #levels launches) is captured once into a hipGraph and replayed.
// A 7-point grid in natural ordering has n1+n2+n3-2 levels (hyperplanes): the solves are launch-latency
// bound, not HBM bound -- see DESIGN.md for measured numbers.
//
// Distributed handles: block-Jacobi ILU(0) of the rank's diagonal block (ghost columns are ignored), the
// usual domain-decomposition preconditioner; it needs no communication.
#include <algorithm>
#include <map>

#include "khip_internal.hpp"

namespace khip {

constexpr int kIluBlock = 256;

// Row i of the factor occupies lu[row_lo[i] .. row_hi[i]) with the diagonal at diag[i]; for a
// non-distributed handle these are rowptr[i], rowptr[i+1]; for a distributed one the owned, sorted part.
struct IluView {
  const int32_t *col;
  const int32_t *row_lo, *diag, *row_hi;
  double *lu;
};

__global__ __launch_bounds__(kIluBlock) void ilu0_factor_level_kernel(IluView v, const int32_t *perm, int64_t lo,
                                                                        int64_t hi, int *bad_row) {
  const int64_t idx = lo + (int64_t)blockIdx.x * kIluBlock + threadIdx.x;
  if (idx >= hi) return;
  const int32_t i = perm[idx];
  const int32_t re = v.row_hi[i], di = v.diag[i];
  for (int32_t kk = v.row_lo[i]; kk < di; ++kk) {
    const int32_t k = v.col[kk];
    const int32_t dk = v.diag[k];
    const double piv = v.lu[dk];
    if (piv == 0.0) { atomicMin(bad_row, (int)k); return; }
    const double lik = v.lu[kk] / piv;
    v.lu[kk] = lik;
    int32_t p = kk + 1;
    const int32_t ke = v.row_hi[k];
    for (int32_t q = dk + 1; q < ke; ++q) {
      const int32_t j = v.col[q];
      while (p < re && v.col[p] < j) ++p;
      if (p < re && v.col[p] == j) {
        const double t = lik * v.lu[q];
        v.lu[p] = v.lu[p] - t;
      }
    }
  }
  if (v.lu[di] == 0.0) atomicMin(bad_row, (int)i);
}

// y[i] = x[i] - sum_{q in [row_lo, diag)} lu[q] * y[col[q]]          (unit lower triangle)
__global__ __launch_bounds__(kIluBlock) void trsv_lower_level_kernel(IluView v, const int32_t *perm, int64_t lo,
                                                                       int64_t hi, const double *x, double *y) {
  const int64_t idx = lo + (int64_t)blockIdx.x * kIluBlock + threadIdx.x;
  if (idx >= hi) return;
  const int32_t i = perm[idx];
  double acc = x[i];
  const int32_t di = v.diag[i];
  for (int32_t q = v.row_lo[i]; q < di; ++q) {
    const double t = v.lu[q] * y[v.col[q]];
    acc = acc - t;
  }
  y[i] = acc;
}

// y[i] = (y[i] - sum_{q in (diag, row_hi)} lu[q] * y[col[q]]) / lu[diag]
__global__ __launch_bounds__(kIluBlock) void trsv_upper_level_kernel(IluView v, const int32_t *perm, int64_t lo,
                                                                       int64_t hi, double *y) {
  const int64_t idx = lo + (int64_t)blockIdx.x * kIluBlock + threadIdx.x;
  if (idx >= hi) return;
  const int32_t i = perm[idx];
  double acc = y[i];
  const int32_t di = v.diag[i], re = v.row_hi[i];
  for (int32_t q = di + 1; q < re; ++q) {
    const double t = v.lu[q] * y[v.col[q]];
    acc = acc - t;
  }
  y[i] = acc / v.lu[di];
}

// Runs of consecutive SMALL levels (<= kIluBlock rows each: the tips of a stencil's wavefront pyramid, or every level of
// a chain-like operator such as a tridiagonal matrix) are executed by ONE workgroup that steps through them with a
// workgroup barrier in between: one launch instead of one per level.  lvl = device copy of the level pointers.
template <int KIND>     // 0: factorisation, 1: lower solve, 2: upper solve
__global__ __launch_bounds__(kIluBlock) void ilu_small_levels_kernel(IluView v, const int32_t *perm, const int64_t *lvl, int l0,
                                                                      int l1, const double *x, double *y, int *bad_row) {
  for (int l = l0; l < l1; ++l) {
    const int64_t idx = lvl[l] + threadIdx.x;
    if (idx < lvl[l + 1]) {
      const int32_t i = perm[idx];
      const int32_t re = v.row_hi[i], di = v.diag[i];
      if (KIND == 0) {
        bool dead = false;
        for (int32_t kk = v.row_lo[i]; kk < di && !dead; ++kk) {
          const int32_t k = v.col[kk];
          const int32_t dk = v.diag[k];
          const double piv = v.lu[dk];
          if (piv == 0.0) { atomicMin(bad_row, (int)k); dead = true; break; }
          const double lik = v.lu[kk] / piv;
          v.lu[kk] = lik;
          int32_t p = kk + 1;
          const int32_t ke = v.row_hi[k];
          for (int32_t q = dk + 1; q < ke; ++q) {
            const int32_t j = v.col[q];
            while (p < re && v.col[p] < j) ++p;
            if (p < re && v.col[p] == j) {
              const double t = lik * v.lu[q];
              v.lu[p] = v.lu[p] - t;
            }
          }
        }
        if (!dead && v.lu[di] == 0.0) atomicMin(bad_row, (int)i);
      } else if (KIND == 1) {
        double acc = x[i];
        for (int32_t q = v.row_lo[i]; q < di; ++q) {
          const double t = v.lu[q] * y[v.col[q]];
          acc = acc - t;
        }
        y[i] = acc;
      } else {
        double acc = y[i];
        for (int32_t q = di + 1; q < re; ++q) {
          const double t = v.lu[q] * y[v.col[q]];
          acc = acc - t;
        }
        y[i] = acc / v.lu[di];
      }
    }
    __threadfence_block();
    __syncthreads();
  }
}

}  // namespace khip

using namespace khip;

struct khip_ilu0 {
  khip_ctx *ctx = nullptr;
  const khip_csr *A = nullptr;
  int64_t n = 0, nnz = 0;
  double *lu = nullptr;
  int32_t *row_lo = nullptr, *diag = nullptr, *row_hi = nullptr;
  int32_t *perm_lo = nullptr, *perm_up = nullptr;
  std::vector<int64_t> lvl_lo, lvl_up;       // level pointers into perm_lo / perm_up
  int64_t *d_lvl_lo = nullptr, *d_lvl_up = nullptr;   // device copies (for the batched small levels)
  int *bad_row = nullptr;
  // cached hipGraph of one application, keyed by the (x, y) pointers it was captured with
  hipGraphExec_t graph = nullptr;
  const double *gx = nullptr;
  double *gy = nullptr;
  bool use_graph = true;
};

namespace {

IluView view_of(const khip_ilu0 *P) { return IluView{P->A->col, P->row_lo, P->diag, P->row_hi, P->lu}; }

unsigned grid_for(int64_t rows) { return (unsigned)((rows + kIluBlock - 1) / kIluBlock); }

// all levels of one triangle: runs of small levels in one single-workgroup launch, every other level its own launch
template <int KIND>
int enqueue_levels(khip_ilu0 *P, const std::vector<int64_t> &lvl, const int64_t *d_lvl, const int32_t *perm, const double *x,
                   double *y) {
  khip_ctx *ctx = P->ctx;
  const IluView v = view_of(P);
  const int nl = (int)lvl.size() - 1;
  int l = 0;
  while (l < nl) {
    int e = l;
    while (e < nl && lvl[e + 1] - lvl[e] <= kIluBlock) ++e;
    if (e - l >= 2) {                                  // a run of at least two small levels
      hipLaunchKernelGGL((ilu_small_levels_kernel<KIND>), dim3(1), dim3(kIluBlock), 0, ctx->stream, v, perm, d_lvl, l, e, x, y,
                         P->bad_row);
      l = e;
      continue;
    }
    const int64_t lo = lvl[l], hi = lvl[l + 1];
    if (KIND == 0) hipLaunchKernelGGL(ilu0_factor_level_kernel, dim3(grid_for(hi - lo)), dim3(kIluBlock), 0, ctx->stream, v, perm, lo, hi, P->bad_row);
    else if (KIND == 1) hipLaunchKernelGGL(trsv_lower_level_kernel, dim3(grid_for(hi - lo)), dim3(kIluBlock), 0, ctx->stream, v, perm, lo, hi, x, y);
    else hipLaunchKernelGGL(trsv_upper_level_kernel, dim3(grid_for(hi - lo)), dim3(kIluBlock), 0, ctx->stream, v, perm, lo, hi, y);
    ++l;
  }
  KHIP_CHECK_HIP(hipGetLastError());
  return KHIP_OK;
}

int enqueue_solve(khip_ilu0 *P, const double *x, double *y) {
  KHIP_TRY((enqueue_levels<1>(P, P->lvl_lo, P->d_lvl_lo, P->perm_lo, x, y)));
  return enqueue_levels<2>(P, P->lvl_up, P->d_lvl_up, P->perm_up, x, y);
}

int ilu0_apply(void *self, const double *x, double *y) {
  khip_ilu0 *P = static_cast<khip_ilu0 *>(self);
  if (P->n == 0) return KHIP_OK;
  khip_ctx *ctx = P->ctx;
  const size_t launches = P->lvl_lo.size() + P->lvl_up.size();
  if (!P->use_graph || launches < 8) return enqueue_solve(P, x, y);
  if (!P->graph || P->gx != x || P->gy != y) {
    if (P->graph) { (void)hipGraphExecDestroy(P->graph); P->graph = nullptr; }
    hipGraph_t g = nullptr;
    KHIP_CHECK_HIP(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    int rc = enqueue_solve(P, x, y);
    hipError_t e = hipStreamEndCapture(ctx->stream, &g);
    if (rc != KHIP_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
    KHIP_CHECK_HIP(e);
    e = hipGraphInstantiate(&P->graph, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    KHIP_CHECK_HIP(e);
    P->gx = x;
    P->gy = y;
  }
  KHIP_CHECK_HIP(hipGraphLaunch(P->graph, ctx->stream));
  return KHIP_OK;
}

void ilu0_free(khip_ilu0 *P) {
  if (!P) return;
  if (P->graph) (void)hipGraphExecDestroy(P->graph);
  for (void *p : {(void *)P->lu, (void *)P->row_lo, (void *)P->diag, (void *)P->row_hi, (void *)P->perm_lo,
                  (void *)P->perm_up, (void *)P->bad_row, (void *)P->d_lvl_lo, (void *)P->d_lvl_up})
    if (p) (void)hipFree(p);
  delete P;
}

// counting sort of the rows by level -> perm, level pointers
void sort_by_level(const std::vector<int32_t> &level, int32_t nlev, std::vector<int32_t> &perm, std::vector<int64_t> &ptr) {
  const int64_t n = (int64_t)level.size();
  ptr.assign((size_t)nlev + 1, 0);
  for (int64_t i = 0; i < n; ++i) ptr[(size_t)level[i] + 1]++;
  for (int32_t l = 0; l < nlev; ++l) ptr[(size_t)l + 1] += ptr[l];
  std::vector<int64_t> cur(ptr.begin(), ptr.end() - 1);
  perm.resize((size_t)n);
  for (int64_t i = 0; i < n; ++i) perm[(size_t)cur[level[i]]++] = (int32_t)i;    // rows stay in increasing order inside a level
}

template <typename T>
int upload(khip_ctx *ctx, const std::vector<T> &h, T **dev) {
  KHIP_CHECK_HIP(hipMalloc(dev, sizeof(T) * std::max<size_t>(h.size(), 1)));
  if (!h.empty()) KHIP_CHECK_HIP(hipMemcpy(*dev, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
  return KHIP_OK;
}

}  // namespace

extern "C" {

int khip_ilu0_create(khip_ctx *ctx, const khip_csr *A, khip_operator *op_out) {
  KHIP_REQUIRE(ctx && A && op_out, "ilu0_create: null argument");
  KHIP_REQUIRE(A->m == A->n || A->dist, "ilu0_create: the operator must be square");
  const int64_t n = A->m, nnz = A->nnz;
  khip_ilu0 *P = new khip_ilu0();
  struct Guard { khip_ilu0 *p; ~Guard() { if (p) ilu0_free(p); } } guard{P};      // any early return frees P
  P->ctx = ctx; P->A = A; P->n = n; P->nnz = nnz;
  P->use_graph = true;
  // ---- host analysis of the pattern (index arrays only) ---------------------------------
  std::vector<int32_t> rowptr((size_t)n + 1), col((size_t)std::max<int64_t>(nnz, 1));
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  KHIP_CHECK_HIP(hipMemcpy(rowptr.data(), A->rowptr, sizeof(int32_t) * (size_t)(n + 1), hipMemcpyDeviceToHost));
  if (nnz) KHIP_CHECK_HIP(hipMemcpy(col.data(), A->col, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToHost));
  std::vector<int32_t> row_lo((size_t)n), diag((size_t)n), row_hi((size_t)n), lev_lo((size_t)n), lev_up((size_t)n);
  int32_t nlev_lo = 0, nlev_up = 0;
  for (int64_t i = 0; i < n; ++i) {
    // owned part of the row: columns < n, contiguous and sorted (ghost columns of a distributed handle,
    // renumbered >= n, sit before / after it)
    int32_t a = rowptr[i], b = rowptr[i + 1];
    while (a < b && col[a] >= n) ++a;
    while (b > a && col[b - 1] >= n) --b;
    int32_t d = -1, lv = 0;
    for (int32_t q = a; q < b; ++q) {
      const int32_t j = col[q];
      if (j >= n || (q > a && col[q - 1] >= j)) {
        set_error("ilu0_create: row %lld: column indices must be sorted and unique", (long long)i);
        return KHIP_ERR_INVALID;
      }
      if (j == i) d = q;
      if (j < i) lv = std::max(lv, lev_lo[j] + 1);
    }
    if (d < 0) {
      set_error("ilu0_create: row %lld has no diagonal entry (structurally zero pivot)", (long long)i);
      return KHIP_ERR_NUMERIC;
    }
    row_lo[i] = a; diag[i] = d; row_hi[i] = b; lev_lo[i] = lv;
    nlev_lo = std::max(nlev_lo, lv + 1);
  }
  for (int64_t i = n - 1; i >= 0; --i) {
    int32_t lv = 0;
    for (int32_t q = diag[i] + 1; q < row_hi[i]; ++q) lv = std::max(lv, lev_up[col[q]] + 1);
    lev_up[i] = lv;
    nlev_up = std::max(nlev_up, lv + 1);
  }
  std::vector<int32_t> perm_lo, perm_up;
  sort_by_level(lev_lo, n ? nlev_lo : 0, perm_lo, P->lvl_lo);
  sort_by_level(lev_up, n ? nlev_up : 0, perm_up, P->lvl_up);
  // ---- device state ------------------------------------------------------------------------
  int rc = upload(ctx, row_lo, &P->row_lo);
  if (!rc) rc = upload(ctx, diag, &P->diag);
  if (!rc) rc = upload(ctx, row_hi, &P->row_hi);
  if (!rc) rc = upload(ctx, perm_lo, &P->perm_lo);
  if (!rc) rc = upload(ctx, perm_up, &P->perm_up);
  if (!rc) rc = upload(ctx, P->lvl_lo, &P->d_lvl_lo);
  if (!rc) rc = upload(ctx, P->lvl_up, &P->d_lvl_up);
  if (rc) return rc;
  hipError_t e = hipMalloc(&P->lu, sizeof(double) * (size_t)std::max<int64_t>(nnz, 1));
  if (e == hipSuccess) e = hipMalloc(&P->bad_row, sizeof(int));
  if (e != hipSuccess) { set_error("ilu0_create: %s", hipGetErrorString(e)); return KHIP_ERR_HIP; }
  const int none = 0x7fffffff;
  KHIP_CHECK_HIP(hipMemcpyAsync(P->bad_row, &none, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  if (nnz) KHIP_CHECK_HIP(hipMemcpyAsync(P->lu, A->val, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToDevice, ctx->stream));
  // ---- numeric factorisation, level by level (same dependency graph as the lower solve) ------
  KHIP_TRY((enqueue_levels<0>(P, P->lvl_lo, P->d_lvl_lo, P->perm_lo, nullptr, nullptr)));
  int bad = none;
  e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(&bad, P->bad_row, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) { set_error("ilu0_create: %s", hipGetErrorString(e)); return KHIP_ERR_HIP; }
  if (bad != none) {
    set_error("ilu0_create: zero pivot in row %d", bad);
    return KHIP_ERR_NUMERIC;
  }
  guard.p = nullptr;
  op_out->csr = nullptr;
  op_out->apply = ilu0_apply;
  op_out->self = P;
  return KHIP_OK;
}

int khip_ilu0_destroy(khip_operator *op) {
  if (!op || op->apply != ilu0_apply || !op->self) return KHIP_OK;
  ilu0_free(static_cast<khip_ilu0 *>(op->self));
  op->self = nullptr;
  return KHIP_OK;
}

int khip_ilu0_info(const khip_operator *op, int64_t *levels_lower, int64_t *levels_upper, const double **lu_dev) {
  KHIP_REQUIRE(op && op->apply == ilu0_apply && op->self, "ilu0_info: not an ILU(0) operator");
  const khip_ilu0 *P = static_cast<const khip_ilu0 *>(op->self);
  if (levels_lower) *levels_lower = (int64_t)P->lvl_lo.size() - 1;
  if (levels_upper) *levels_upper = (int64_t)P->lvl_up.size() - 1;
  if (lu_dev) *lu_dev = P->lu;
  return KHIP_OK;
}

int khip_ilu0_set_graph(khip_operator *op, int enable) {
  KHIP_REQUIRE(op && op->apply == ilu0_apply && op->self, "ilu0_set_graph: not an ILU(0) operator");
  static_cast<khip_ilu0 *>(op->self)->use_graph = enable != 0;
  return KHIP_OK;
}

}  // extern "C"
