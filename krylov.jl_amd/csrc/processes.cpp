// Krylov processes on the device: hermitian_lanczos, arnoldi, golub_kahan.
//
// Reference: src/krylov_processes.jl:28-102 (hermitian_lanczos), :250-296 (arnoldi), :323-398 (golub_kahan).
// Same order of kmul! / kdotr / kaxpy! / knorm / kdivcopy! as the reference, on the same device primitives the
// solvers use: the orthogonalisation runs through the MGS cascade (khip_mgs) whose coefficients stay on the
// device until the one host read a step needs (the breakdown test of :92 / :288 is a host branch in the reference
// too).  Bases are dense column-major n x (k+1) blocks in HBM with a leading dimension (`M(undef, n, k+1)`, :52);
// the small matrices are returned on the host in the reference's own storage: T and L as the `nzval` of the
// SparseMatrixCSC the reference builds (:35-48, :331-347), H dense (k+1) x k column-major (:260).
// Row-partitioned operators work unchanged: n is then the local row count and every dot / norm is all-reduced
// inside the primitives.
#include <cmath>
#include <cstring>
#include <vector>

#include "khip_internal.hpp"

using namespace khip;

namespace {

int apply(khip_ctx *ctx, const khip_operator *op, const double *x, double *y) {
  if (op->apply) {
    const int rc = op->apply(op->self, x, y);
    if (rc != 0) { set_error("user operator returned %d", rc); return KHIP_ERR_INVALID; }
    return KHIP_OK;
  }
  if (!op->csr) { set_error("operator has neither a CSR handle nor an apply callback"); return KHIP_ERR_INVALID; }
  return khip_spmv(ctx, op->csr, x, y);
}

bool basis_ok(const double *V, int64_t n, int64_t ld) {
  return V && ld >= n && (ld % 2 == 0) && (reinterpret_cast<uintptr_t>(V) % 16 == 0);
}

// v <- b / ||b||  or  v <- 0 on an exact breakdown (:61-68, :264-271, :355-367)
int first_vector(khip_ctx *ctx, int64_t n, const double *b, double *v, int allow_breakdown, const char *what,
                 double *beta_out) {
  double beta = 0.0;
  KHIP_TRY(khip_nrm2(ctx, n, b, &beta));
  *beta_out = beta;
  if (beta == 0.0) {
    if (!allow_breakdown) { set_error("Exact breakdown %s == 0.", what); return KHIP_ERR_NUMERIC; }
    return khip_fill(ctx, n, v, 0.0);
  }
  return khip_divcopy(ctx, n, v, b, beta);
}

// q <- q / s in place, or q <- 0 with the reference's error when s == 0 and breakdowns are not allowed
int normalise(khip_ctx *ctx, int64_t n, double *q, double s, int allow_breakdown, const char *what, int it) {
  if (s == 0.0) {
    if (!allow_breakdown) { set_error("Exact breakdown %s == 0 at iteration i = %d.", what, it); return KHIP_ERR_NUMERIC; }
    return khip_fill(ctx, n, q, 0.0);
  }
  return khip_divcopy(ctx, n, q, q, s);
}

}  // namespace

extern "C" {

int khip_hermitian_lanczos(khip_ctx *ctx, const khip_operator *A, int64_t n, const double *b, int k,
                           int allow_breakdown, int reorthogonalization, double *V, int64_t ldv,
                           double *beta1_host, double *T_nzval_host) {
  KHIP_REQUIRE(ctx && A && b && beta1_host && T_nzval_host && n >= 0 && k >= 1, "hermitian_lanczos: bad argument");
  KHIP_REQUIRE(basis_ok(V, n, ldv), "hermitian_lanczos: V must be 16-byte aligned with an even leading dimension >= n");
  double *nz = T_nzval_host;
  memset(nz, 0, sizeof(double) * (size_t)(3 * k - 1));
  int pa = 0;                                                    // position of alpha_i in nzval (:54)
  for (int i = 0; i < k; ++i) {
    double *vi = V + (int64_t)i * ldv;
    double *q = V + (int64_t)(i + 1) * ldv;
    double *vim1 = i > 0 ? V + (int64_t)(i - 1) * ldv : nullptr;
    if (i == 0) KHIP_TRY(first_vector(ctx, n, b, vi, allow_breakdown, "β₁", beta1_host));   // :60-68
    KHIP_TRY(apply(ctx, A, vi, q));                              // :70
    if (i >= 1) {                                                // :71-76
      const double beta_i = nz[pa - 2];
      nz[pa - 1] = beta_i;
      KHIP_TRY(khip_axpy(ctx, n, -beta_i, vim1, q));
    }
    double alpha = 0.0, beta_next = 0.0;
    const double *one[1] = {vi};
    if (!reorthogonalization) {
      KHIP_TRY(khip_mgs(ctx, n, 1, one, q, &alpha, &beta_next, 0));            // :77-78 and :91 in one cascade
    } else {
      KHIP_TRY(khip_mgs(ctx, n, 1, one, q, &alpha, nullptr, 0));               // :77-78
      if (i >= 1) {                                                            // :80-85 then :86-88
        const double *two[2] = {vim1, vi};
        double t[2];
        KHIP_TRY(khip_mgs(ctx, n, 2, two, q, t, &beta_next, 0));
        nz[pa - 2] += t[0];
        nz[pa - 1] += t[0];
        alpha += t[1];
      } else {
        double t;
        KHIP_TRY(khip_mgs(ctx, n, 1, one, q, &t, &beta_next, 0));              // :86-88
        alpha += t;
      }
    }
    nz[pa] = alpha;                                                            // :90
    KHIP_TRY(normalise(ctx, n, q, beta_next, allow_breakdown, "βᵢ₊₁", i + 1));   // :92-97
    nz[pa + 1] = beta_next;                                                    // :98
    pa += 3;
  }
  return KHIP_OK;
}

int khip_arnoldi(khip_ctx *ctx, const khip_operator *A, int64_t n, const double *b, int k, int allow_breakdown,
                 int reorthogonalization, double *V, int64_t ldv, double *beta_host, double *H_host) {
  KHIP_REQUIRE(ctx && A && b && beta_host && H_host && n >= 0 && k >= 1, "arnoldi: bad argument");
  KHIP_REQUIRE(basis_ok(V, n, ldv), "arnoldi: V must be 16-byte aligned with an even leading dimension >= n");
  const int ldh = k + 1;
  memset(H_host, 0, sizeof(double) * (size_t)ldh * k);           // :260
  std::vector<const double *> cols((size_t)k + 1);
  for (int j = 0; j <= k; ++j) cols[j] = V + (int64_t)j * ldv;
  for (int j = 0; j < k; ++j) {
    double *vj = V + (int64_t)j * ldv;
    double *q = V + (int64_t)(j + 1) * ldv;
    if (j == 0) KHIP_TRY(first_vector(ctx, n, b, vj, allow_breakdown, "β", beta_host));   // :265-272
    KHIP_TRY(apply(ctx, A, vj, q));                              // :274
    double *h = H_host + (size_t)j * ldh;
    double hn = 0.0;
    KHIP_TRY(khip_mgs(ctx, n, j + 1, cols.data(), q, h, reorthogonalization ? nullptr : &hn, 0));   // :275-279
    if (reorthogonalization) KHIP_TRY(khip_mgs(ctx, n, j + 1, cols.data(), q, h, &hn, 1));          // :280-286
    h[j + 1] = hn;                                               // :287
    KHIP_TRY(normalise(ctx, n, q, hn, allow_breakdown, "Hᵢ₊₁.ᵢ", j + 1));   // :288-293
  }
  return KHIP_OK;
}

int khip_golub_kahan(khip_ctx *ctx, const khip_operator *A, const khip_operator *At, int64_t m, int64_t n,
                     const double *b, int k, int allow_breakdown, double *V, int64_t ldv, double *U, int64_t ldu,
                     double *beta1_host, double *L_nzval_host) {
  KHIP_REQUIRE(ctx && A && At && b && beta1_host && L_nzval_host && m >= 0 && n >= 0 && k >= 1, "golub_kahan: bad argument");
  KHIP_REQUIRE(basis_ok(V, n, ldv) && basis_ok(U, m, ldu),
               "golub_kahan: V and U must be 16-byte aligned with even leading dimensions >= n and >= m");
  double *nz = L_nzval_host;
  memset(nz, 0, sizeof(double) * (size_t)(2 * k + 1));
  int pa = 0;
  for (int i = 0; i < k; ++i) {
    double *ui = U + (int64_t)i * ldu, *vi = V + (int64_t)i * ldv;
    double *q = U + (int64_t)(i + 1) * ldu, *p = V + (int64_t)(i + 1) * ldv;
    if (i == 0) {                                                // :359-377
      KHIP_TRY(first_vector(ctx, m, b, ui, allow_breakdown, "β₁", beta1_host));
      KHIP_TRY(apply(ctx, At, ui, vi));
      double alpha1 = 0.0;
      KHIP_TRY(khip_nrm2(ctx, n, vi, &alpha1));
      if (alpha1 == 0.0 && !allow_breakdown) { set_error("Exact breakdown α₁ == 0."); return KHIP_ERR_NUMERIC; }
      KHIP_TRY(normalise(ctx, n, vi, alpha1, 1, "", 0));
      nz[pa] = alpha1;
    }
    KHIP_TRY(apply(ctx, A, vi, q));                              // :378
    const double alpha = nz[pa];
    double sq = 0.0;
    KHIP_TRY(khip_axpy_sqnorm(ctx, m, -alpha, ui, q, &sq));      // :380-381 in one pass
    const double beta_next = std::sqrt(sq);
    KHIP_TRY(normalise(ctx, m, q, beta_next, allow_breakdown, "βᵢ₊₁", i + 1));   // :382-387
    KHIP_TRY(apply(ctx, At, q, p));                              // :388
    KHIP_TRY(khip_axpy_sqnorm(ctx, n, -beta_next, vi, p, &sq));  // :389-390
    const double alpha_next = std::sqrt(sq);
    KHIP_TRY(normalise(ctx, n, p, alpha_next, allow_breakdown, "αᵢ₊₁", i + 1));  // :391-396
    nz[pa + 1] = beta_next;                                      // :397-398
    nz[pa + 2] = alpha_next;
    pa += 2;
  }
  return KHIP_OK;
}

}  // extern "C"
