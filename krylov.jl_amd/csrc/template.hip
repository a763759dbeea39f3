// template.hip -- row-template compression of a CSR handle ("CSR-T"): detection and table construction.
//
// Stencil-type operators repeat a handful of ROW TEMPLATES: the sequence of (column - row, value) pairs of
// a row.  get_div_grad(n1,n2,n3) (test/get_div_grad.jl:8-25) has 27 of them (interior + faces / edges /
// corners), however large the grid.  When a handle has at most 1024 distinct templates of at most 32
// entries, khip_csr_compress stores ONE 16-bit template id per row next to the CSR arrays and the SpMV reads
// 2 bytes of matrix data per row instead of 12 per nonzero + 4 per row: 18 B/row instead of 104 B/row for the
// 7-point operator.  The kernel (spmv_template_kernel, spmv.hip) walks the template out of LDS in the
// stored order with the same rounded multiply + rounded add per entry, so y is BIT-IDENTICAL to the CSR
// kernels and to the serial CPU loop; the format is an internal representation of the same operator, the
// boundary still takes plain CSR.  This is the "value / index dictionary" idea of CSR-VI / CSR-DU (Kourtis,
// Goumas, Koziris 2008) taken to whole rows; SURVEY.md section 8f lists the matrix-free stencil operator as
// next-row N4.  It is OPT-IN (khip_csr_compress); the headline CSR numbers never use it.
//
// Detection, all on the device: 64-bit hash of every row -> open-addressing table (atomicCAS) with the
// smallest row index per distinct hash as representative -> ids in order of representative row (host
// sorts <= 1024 entries, deterministic) -> table built from the representative rows -> every row is
// compared ENTRY BY ENTRY with its template (a hash collision or any mismatch aborts the compression).
#include <algorithm>

#include "spmv_common.hpp"

namespace khip {

constexpr int kTmplMaxLen = 32;        // entries per row
constexpr int kTmplMax = 1024;         // distinct templates
constexpr int kTmplHash = 8192;        // hash table slots (power of two)
constexpr size_t kTmplLdsMax = 60 * 1024;

enum TmplFail { TMPL_OK = 0, TMPL_ROW_TOO_LONG = 1, TMPL_TOO_MANY = 2, TMPL_MISMATCH = 3 };

__device__ __forceinline__ unsigned long long mix64(unsigned long long h, unsigned long long v) {
  h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
  h *= 0xff51afd7ed558ccdull;
  h ^= h >> 33;
  return h;
}

__device__ __forceinline__ unsigned long long row_hash(const int32_t *rowptr, const int32_t *col, const double *val,
                                                        int64_t row, int *fail) {
  const int32_t s = rowptr[row], e = rowptr[row + 1];
  if (e - s > kTmplMaxLen) { atomicMax(fail, (int)TMPL_ROW_TOO_LONG); return 1ull; }
  unsigned long long h = mix64(0x243f6a8885a308d3ull, (unsigned long long)(e - s));
  for (int32_t q = s; q < e; ++q) {
    h = mix64(h, (unsigned long long)(unsigned int)(col[q] - (int32_t)row));
    h = mix64(h, (unsigned long long)__double_as_longlong(val[q]));
  }
  return h | 1ull;      // 0 marks an empty slot
}

__global__ __launch_bounds__(kBlock) void tmpl_insert_kernel(const int32_t *rowptr, const int32_t *col, const double *val,
                                                             int64_t m, unsigned long long *keys, int *rep, int *count,
                                                             int *fail) {
  const int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (row >= m || *fail) return;
  const unsigned long long h = row_hash(rowptr, col, val, row, fail);
  unsigned slot = (unsigned)(h >> 20) & (kTmplHash - 1);
  for (int probe = 0; probe < 256; ++probe) {
    const unsigned long long prev = atomicCAS(&keys[slot], 0ull, h);
    if (prev == 0ull || prev == h) {
      if (prev == 0ull && atomicAdd(count, 1) >= kTmplMax) atomicMax(fail, (int)TMPL_TOO_MANY);
      atomicMin(&rep[slot], (int)row);
      return;
    }
    slot = (slot + 1) & (kTmplHash - 1);
  }
  atomicMax(fail, (int)TMPL_TOO_MANY);
}

// table[t] <- entries of the representative row of template t
__global__ void tmpl_build_kernel(const int32_t *rowptr, const int32_t *col, const double *val, const int *rep_of_id, int T,
                                  int K, int32_t *t_off, double *t_val, int32_t *t_cnt) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const int64_t row = rep_of_id[t];
  const int32_t s = rowptr[row], e = rowptr[row + 1];
  t_cnt[t] = e - s;
  for (int k = 0; k < K; ++k) {
    const bool in = k < e - s;
    t_off[(size_t)t * K + k] = in ? col[s + k] - (int32_t)row : 0;
    t_val[(size_t)t * K + k] = in ? val[s + k] : 0.0;
  }
}

// id of every row + exact verification against its template
__global__ __launch_bounds__(kBlock) void tmpl_assign_kernel(const int32_t *rowptr, const int32_t *col, const double *val,
                                                             int64_t m, const unsigned long long *keys, const int *id_of_slot,
                                                             int K, const int32_t *t_off, const double *t_val,
                                                             const int32_t *t_cnt, uint16_t *tmpl_id, int *fail) {
  const int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (row >= m) return;
  const unsigned long long h = row_hash(rowptr, col, val, row, fail);
  unsigned slot = (unsigned)(h >> 20) & (kTmplHash - 1);
  int t = -1;
  for (int probe = 0; probe < 256; ++probe) {
    if (keys[slot] == h) { t = id_of_slot[slot]; break; }
    slot = (slot + 1) & (kTmplHash - 1);
  }
  if (t < 0) { atomicMax(fail, (int)TMPL_MISMATCH); return; }
  const int32_t s = rowptr[row], e = rowptr[row + 1];
  bool same = (e - s) == t_cnt[t];
  for (int32_t q = s; same && q < e; ++q)
    same = (col[q] - (int32_t)row == t_off[(size_t)t * K + (q - s)]) &&
           (__double_as_longlong(val[q]) == __double_as_longlong(t_val[(size_t)t * K + (q - s)]));
  if (!same) { atomicMax(fail, (int)TMPL_MISMATCH); return; }
  tmpl_id[row] = (uint16_t)t;
}

void csr_free_templates(khip_csr *A) {
  (void)hipFree(A->tmpl_id); (void)hipFree(A->tmpl_off); (void)hipFree(A->tmpl_val); (void)hipFree(A->tmpl_cnt);
  A->tmpl_id = nullptr; A->tmpl_off = nullptr; A->tmpl_val = nullptr; A->tmpl_cnt = nullptr;
  A->tmpl_T = 0; A->tmpl_K = 0;
}

}  // namespace khip

using namespace khip;

extern "C" int khip_csr_compress(khip_ctx *ctx, khip_csr *A, int *templates_out) {
  KHIP_REQUIRE(ctx && A, "csr_compress: null argument");
  if (templates_out) *templates_out = 0;
  csr_free_templates(A);
  const int64_t m = A->m;
  if (m == 0 || A->max_row_nnz > kTmplMaxLen || A->max_row_nnz < 1) return KHIP_OK;       // not compressible: stays CSR
  unsigned long long *keys = nullptr;
  int *rep = nullptr, *cnt_fail = nullptr;
  KHIP_CHECK_HIP(hipMalloc(&keys, sizeof(unsigned long long) * kTmplHash));
  KHIP_CHECK_HIP(hipMalloc(&rep, sizeof(int) * kTmplHash));
  KHIP_CHECK_HIP(hipMalloc(&cnt_fail, sizeof(int) * 2));
  int *d_id_of_slot = nullptr, *d_rep_of_id = nullptr;
  bool keep_templates = false;
  struct Scratch {                       // frees the scratch (and a half-built table) on every path out of this function
    unsigned long long *&keys; int *&rep; int *&cnt_fail; int *&a; int *&b; khip_csr *A; bool &keep;
    ~Scratch() {
      (void)hipFree(keys); (void)hipFree(rep); (void)hipFree(cnt_fail); (void)hipFree(a); (void)hipFree(b);
      if (!keep) csr_free_templates(A);
    }
  } scratch{keys, rep, cnt_fail, d_id_of_slot, d_rep_of_id, A, keep_templates};
  KHIP_CHECK_HIP(hipMemsetAsync(keys, 0, sizeof(unsigned long long) * kTmplHash, ctx->stream));
  KHIP_CHECK_HIP(hipMemsetAsync(rep, 0x7f, sizeof(int) * kTmplHash, ctx->stream));
  KHIP_CHECK_HIP(hipMemsetAsync(cnt_fail, 0, sizeof(int) * 2, ctx->stream));
  const unsigned grid = (unsigned)((m + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(tmpl_insert_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, A->rowptr, A->col, A->val, m, keys, rep,
                     cnt_fail, cnt_fail + 1);
  int cf[2] = {0, 0};
  std::vector<unsigned long long> hkeys(kTmplHash);
  std::vector<int> hrep(kTmplHash);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(cf, cnt_fail, sizeof(cf), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) { set_error("csr_compress: %s", hipGetErrorString(e)); return KHIP_ERR_HIP; }
  if (cf[1] != TMPL_OK) return KHIP_OK;                                   // too many templates / rows too long
  KHIP_CHECK_HIP(hipMemcpy(hkeys.data(), keys, sizeof(unsigned long long) * kTmplHash, hipMemcpyDeviceToHost));
  KHIP_CHECK_HIP(hipMemcpy(hrep.data(), rep, sizeof(int) * kTmplHash, hipMemcpyDeviceToHost));
  // ids in order of the representative (smallest) row: deterministic
  std::vector<std::pair<int, int>> occ;      // (representative row, slot)
  for (int s = 0; s < kTmplHash; ++s) if (hkeys[s]) occ.emplace_back(hrep[s], s);
  std::sort(occ.begin(), occ.end());
  const int T = (int)occ.size();
  const int K = (int)A->max_row_nnz;
  if (T == 0 || T > kTmplMax || (size_t)T * K * 12 + (size_t)T * 4 > kTmplLdsMax) return KHIP_OK;
  std::vector<int> id_of_slot(kTmplHash, -1), rep_of_id((size_t)T);
  for (int t = 0; t < T; ++t) { id_of_slot[occ[t].second] = t; rep_of_id[t] = occ[t].first; }
  KHIP_CHECK_HIP(hipMalloc(&d_id_of_slot, sizeof(int) * kTmplHash));
  KHIP_CHECK_HIP(hipMalloc(&d_rep_of_id, sizeof(int) * (size_t)T));
  KHIP_CHECK_HIP(hipMemcpy(d_id_of_slot, id_of_slot.data(), sizeof(int) * kTmplHash, hipMemcpyHostToDevice));
  KHIP_CHECK_HIP(hipMemcpy(d_rep_of_id, rep_of_id.data(), sizeof(int) * (size_t)T, hipMemcpyHostToDevice));
  KHIP_CHECK_HIP(hipMalloc(&A->tmpl_id, sizeof(uint16_t) * (size_t)(m + 8)));
  KHIP_CHECK_HIP(hipMalloc(&A->tmpl_off, sizeof(int32_t) * (size_t)T * K));
  KHIP_CHECK_HIP(hipMalloc(&A->tmpl_val, sizeof(double) * (size_t)T * K));
  KHIP_CHECK_HIP(hipMalloc(&A->tmpl_cnt, sizeof(int32_t) * (size_t)T));
  hipLaunchKernelGGL(tmpl_build_kernel, dim3((T + 63) / 64), dim3(64), 0, ctx->stream, A->rowptr, A->col, A->val, d_rep_of_id,
                     T, K, A->tmpl_off, A->tmpl_val, A->tmpl_cnt);
  hipLaunchKernelGGL(tmpl_assign_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, A->rowptr, A->col, A->val, m, keys,
                     d_id_of_slot, K, A->tmpl_off, A->tmpl_val, A->tmpl_cnt, A->tmpl_id, cnt_fail + 1);
  e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(cf, cnt_fail, sizeof(cf), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) { set_error("csr_compress: %s", hipGetErrorString(e)); return KHIP_ERR_HIP; }
  if (cf[1] != TMPL_OK) return KHIP_OK;                                                  // hash collision: stay with CSR
  keep_templates = true;
  A->tmpl_T = T;
  A->tmpl_K = K;
  if (templates_out) *templates_out = T;
  return KHIP_OK;
}
