// colcode.hip -- dictionary-coded column indices for the staged SpMV (spmv_code_kernel, spmv.hip).
//
// A banded / stencil operator touches few distinct DIAGONALS d = column - row: get_div_grad(n1,n2,n3)
// (test/get_div_grad.jl:8-25) has 7 whatever the grid size, the kron-unsymmetric operator 7, the 27-point operator 27,
// and a rank's renumbered [owned | ghost] slab two more (every ghost plane sits at one constant offset from the rows
// that reference it).  When all entries of a handle lie on at most 256 diagonals the handle keeps, NEXT TO its CSR
// arrays, one byte per entry: the rank of d in the ascending table of distinct diagonals (two bytes up to 2048
// diagonals).  The staged SpMV then streams 8 + 1 bytes per entry instead of 8 + 4 and rebuilds col = row + tab[code]
// with exact integer arithmetic, so nothing about the product changes -- not the order of the entries, not a bit of y.
// The boundary (khip_csr_create, src/krylov_utils.jl:305 kmul!) still takes and keeps plain CSR; every other kernel
// (SpMM, transpose, ILU, halo plan) reads the int32 columns.  Operators with more diagonals simply stay on the int32 stream.
//
// Construction, on the device, at the first product that can use it: every row inserts its offsets into a 4096-slot
// open-addressing table (read first, atomicCAS only on an empty slot -> after the first few workgroups the pass is a
// read-only stream of col), the host sorts the <= 2048 keys (deterministic), a second pass writes rank(d) per entry.
#include <algorithm>

#include "spmv_common.hpp"

namespace khip {

constexpr int kCodeHash = 4096;            // slots (power of two)
constexpr int kCodeMax = 2048;             // distinct diagonals at most (8 KB of LDS per workgroup)
constexpr int32_t kCodeEmpty = INT32_MIN;  // never a valid offset (|col - row| < 2^31 - 1)
constexpr int kCodePad = 64;               // zeroed codes behind the last entry (lane loads may overrun)

__device__ __forceinline__ unsigned code_hash(int32_t d) { return ((unsigned)d * 2654435761u) >> 20; }   // 12 bits

__global__ __launch_bounds__(kBlock) void code_collect_kernel(const int32_t *rowptr, const int32_t *col, int64_t m,
                                                              int32_t *keys, int *count_fail) {
  const int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (row >= m) return;
  if (__atomic_load_n(&count_fail[1], __ATOMIC_RELAXED)) return;
  int32_t last = kCodeEmpty;
  for (int32_t q = rowptr[row]; q < rowptr[row + 1]; ++q) {
    const int32_t d = col[q] - (int32_t)row;
    if (d == last) continue;
    last = d;
    unsigned slot = code_hash(d) & (kCodeHash - 1);
    bool done = false;
    for (int probe = 0; probe < kCodeHash && !done; ++probe) {
      int32_t cur = __atomic_load_n(&keys[slot], __ATOMIC_RELAXED);
      if (cur == kCodeEmpty) {
        cur = atomicCAS(&keys[slot], kCodeEmpty, d);
        if (cur == kCodeEmpty) {
          if (atomicAdd(&count_fail[0], 1) >= kCodeMax) atomicMax(&count_fail[1], 1);
          cur = d;
        }
      }
      if (cur == d) done = true;
      else slot = (slot + 1) & (kCodeHash - 1);
    }
    if (!done) { atomicMax(&count_fail[1], 1); return; }
  }
}

template <typename CODE>
__global__ __launch_bounds__(kBlock) void code_assign_kernel(const int32_t *rowptr, const int32_t *col, int64_t m,
                                                             const int32_t *tab, int T, CODE *code, int *fail) {
  extern __shared__ int32_t s_tab[];
  for (int i = threadIdx.x; i < T; i += kBlock) s_tab[i] = tab[i];
  __syncthreads();
  const int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (row >= m) return;
  for (int32_t q = rowptr[row]; q < rowptr[row + 1]; ++q) {
    const int32_t d = col[q] - (int32_t)row;
    int lo = 0, hi = T - 1;
    while (lo < hi) {                      // first entry >= d
      const int mid = (lo + hi) >> 1;
      if (s_tab[mid] < d) lo = mid + 1; else hi = mid;
    }
    if (s_tab[lo] != d) { atomicMax(fail, 1); return; }
    code[q] = (CODE)lo;
  }
}

void csr_free_codes(khip_csr *A) {
  (void)hipFree(A->code); (void)hipFree(A->code_tab);
  A->code = nullptr; A->code_tab = nullptr;
  A->code_T = 0; A->code_bits = 0; A->code_state = 0;
}

int csr_build_codes(khip_ctx *ctx, khip_csr *A) {
  csr_free_codes(A);
  A->code_state = -1;
  const int64_t m = A->m;
  if (m == 0 || A->nnz == 0) return KHIP_OK;
  int32_t *keys = nullptr;
  int *cf = nullptr;
  KHIP_CHECK_HIP(hipMalloc(&keys, sizeof(int32_t) * kCodeHash));
  KHIP_CHECK_HIP(hipMalloc(&cf, sizeof(int) * 2));
  bool keep = false;
  struct Scratch {
    int32_t *&keys; int *&cf; khip_csr *A; bool &keep;
    ~Scratch() {
      (void)hipFree(keys); (void)hipFree(cf);
      if (!keep) { csr_free_codes(A); A->code_state = -1; }
    }
  } scratch{keys, cf, A, keep};
  std::vector<int32_t> init(kCodeHash, kCodeEmpty);
  KHIP_CHECK_HIP(hipMemcpyAsync(keys, init.data(), sizeof(int32_t) * kCodeHash, hipMemcpyHostToDevice, ctx->stream));
  KHIP_CHECK_HIP(hipMemsetAsync(cf, 0, sizeof(int) * 2, ctx->stream));
  const unsigned grid = (unsigned)((m + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(code_collect_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, A->rowptr, A->col, m, keys, cf);
  KHIP_CHECK_HIP(hipGetLastError());
  int h[2] = {0, 0};
  std::vector<int32_t> hk(kCodeHash);
  KHIP_CHECK_HIP(hipMemcpyAsync(h, cf, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  KHIP_CHECK_HIP(hipMemcpyAsync(hk.data(), keys, sizeof(int32_t) * kCodeHash, hipMemcpyDeviceToHost, ctx->stream));
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  if (h[1] || h[0] > kCodeMax) return KHIP_OK;                       // too many diagonals: stays on the int32 stream
  std::vector<int32_t> tab;
  for (int32_t k : hk) if (k != kCodeEmpty) tab.push_back(k);
  std::sort(tab.begin(), tab.end());
  const int T = (int)tab.size();
  if (T == 0 || T > kCodeMax) return KHIP_OK;
  const int bits = (T <= 256 && ctx->tune.spmv_codes != 16) ? 8 : 16;
  KHIP_CHECK_HIP(hipMalloc(&A->code_tab, sizeof(int32_t) * (size_t)T));
  KHIP_CHECK_HIP(hipMemcpyAsync(A->code_tab, tab.data(), sizeof(int32_t) * (size_t)T, hipMemcpyHostToDevice, ctx->stream));
  const size_t bytes = (size_t)(A->nnz + kCodePad) * (size_t)(bits / 8);
  KHIP_CHECK_HIP(hipMalloc(&A->code, bytes));
  KHIP_CHECK_HIP(hipMemsetAsync((char *)A->code + (size_t)A->nnz * (size_t)(bits / 8), 0, (size_t)kCodePad * (size_t)(bits / 8),
                                ctx->stream));
  KHIP_CHECK_HIP(hipMemsetAsync(cf, 0, sizeof(int) * 2, ctx->stream));
  if (bits == 8)
    hipLaunchKernelGGL(code_assign_kernel<uint8_t>, dim3(grid), dim3(kBlock), sizeof(int32_t) * (size_t)T, ctx->stream, A->rowptr,
                       A->col, m, A->code_tab, T, (uint8_t *)A->code, cf + 1);
  else
    hipLaunchKernelGGL(code_assign_kernel<uint16_t>, dim3(grid), dim3(kBlock), sizeof(int32_t) * (size_t)T, ctx->stream,
                       A->rowptr, A->col, m, A->code_tab, T, (uint16_t *)A->code, cf + 1);
  KHIP_CHECK_HIP(hipGetLastError());
  KHIP_CHECK_HIP(hipMemcpyAsync(h, cf, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));      // also keeps `tab` alive until the upload is done
  if (h[1]) return KHIP_OK;
  keep = true;
  A->code_T = T;
  A->code_bits = bits;
  A->code_state = 1;
  return KHIP_OK;
}

}  // namespace khip
