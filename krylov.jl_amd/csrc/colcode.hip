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

// ---------------------------------------------------------------- sliced form (spmv_sell_kernel) ----
// The coded kernel reads the CSR value / code streams of a row block coalesced, parks them in LDS and lets lane i walk row i out of
// LDS: rowptr -> window loads -> barrier -> LDS reads -> gathers is one dependent chain per workgroup.  Here the SAME entries, in the
// SAME order per row, are stored a second time transposed per 64-row slice, so that lane l reads entry k of its own row with a
// coalesced 8-byte load (512 contiguous bytes per wave instruction): slice s starts at unit off[s] (a unit = 64 words of 8 bytes);
// its first W = ceil(L / 8) units hold, per lane, eight 1-byte codes per word (0xFF = the row has no such entry), the next L units
// the values (L = longest row of the slice).  64 B per row for the 7-point operator (7 values + one code word) where CSR + codes
// + row pointer is 67.  y is bit-identical (stored order, one rounded multiply and one rounded add per entry).
constexpr double kSellMaxPad = 1.20;       // bytes of the sliced copy / bytes of the CSR stream it replaces above which it is not built
constexpr double kSellUniformPad = 1.02;   // padding every slice to the longest one may cost this much (then there is no offset array)

__global__ __launch_bounds__(kBlock) void sell_units_kernel(const int32_t *rowptr, int64_t m, int64_t slices, int32_t *units, int mode) {
  const int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (s >= slices) return;
  int L = 0;
  const int64_t r0 = s * 64, r1 = (r0 + 64 < m) ? r0 + 64 : m;
  int32_t prev = rowptr[r0];
  for (int64_t r = r0; r < r1; ++r) {
    const int32_t nxt = rowptr[r + 1];
    L = (nxt - prev > L) ? nxt - prev : L;
    prev = nxt;
  }
  const int Lp = L + (L & 1);                // pair layouts: an even number of value words ...
  const int h4 = (Lp / 2) + ((Lp / 2) & 1), h5 = ((L + 7) / 8) + (((L + 7) / 8) & 1);      // ... behind an even number of column / code words
  units[s] = L > 0 ? (mode == 5 ? Lp + h5 : (mode == 4 ? Lp + h4 : (mode == 3 ? 2 * ((L + 2) / 2) : L + (mode == 2 ? 0 : (mode == 1 ? (L + 1) / 2 : (L + 7) / 8))))) : 0;      // mode 0: 8-bit code words, 1: int32 column words, 2: values only (narrow codes live in their own array), 3: one code word + values, an even number of words (16-byte pairs; L <= 8), 4 / 5: int32 column words / code words padded to an even count, then the values padded to an even count (16-byte pairs, any L)
}

// words per row ahead of the values for T units of a slice: code words (T = L + ceil(L / 8)) or column words (T = L + ceil(L / 2))
__host__ __device__ __forceinline__ int sell_head_words(int T, int mode) {
  if (T <= 0) return 0;
  if (mode == 5) return T <= 18 ? 2 : (T <= 36 ? 4 : (T <= 54 ? 6 : 8));       // T = Lp + even(ceil(Lp / 8)), Lp even <= 64: disjoint ranges
  if (mode == 4) return 2 * ((T - 4) / 6) + 2;                                 // T = Lp + even(Lp / 2)
  return mode == 3 ? 1 : (mode == 2 ? 0 : (mode == 1 ? (T + 2) / 3 : (T + 8) / 9));
}

__global__ __launch_bounds__(kBlock) void sell_fill_kernel(const int32_t *rowptr, const double *val, const uint8_t *code, const int32_t *col, int64_t m,
                                                           int64_t slices, const uint32_t *off, int uniform_units,
                                                           unsigned long long *sell, uint32_t *c4, int mode) {
  const int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x;      // rows of the last slice beyond m are written too (no entry)
  if (row >= slices * 64) return;
  const int64_t s = row >> 6;
  const int lane = (int)(row & 63);
  const int64_t o0 = uniform_units ? s * uniform_units : (int64_t)off[s];
  const int T = uniform_units ? uniform_units : (int)(off[s + 1] - off[s]);
  const int cols32 = col != nullptr;
  const bool pair = mode >= 3;
  const int32_t q0 = row < m ? rowptr[row] : 0;
  const int len = row < m ? rowptr[row + 1] - q0 : 0;
  if (c4) {                                   // eight 4-bit codes, 0xF = no entry
    uint32_t word = 0;
    for (int u = 0; u < 8; ++u) word |= (u < len ? (uint32_t)code[q0 + u] : 0xFu) << (4 * u);
    c4[row] = word;
  }
  if (T == 0) return;
  const int W = sell_head_words(T, mode), L = T - W;
  unsigned long long *base = sell + (size_t)o0 * 64 + lane;
  // word w of this lane: plain layout (w * 64 + lane); pair layout ((w / 2) * 64 + lane) * 2 + (w & 1)
  auto slot = [&](int w) -> unsigned long long & { return pair ? sell[(size_t)o0 * 64 + ((size_t)(w >> 1) * 64 + lane) * 2 + (w & 1)] : base[(size_t)w * 64]; };
  for (int w = 0; w < W; ++w) {
    unsigned long long word = 0;
    if (cols32) {
      const unsigned long long c0 = 2 * w < len ? (unsigned long long)(uint32_t)col[q0 + 2 * w] : 0xFFFFFFFFull;
      const unsigned long long c1 = 2 * w + 1 < len ? (unsigned long long)(uint32_t)col[q0 + 2 * w + 1] : 0xFFFFFFFFull;
      word = c0 | (c1 << 32);
    } else {
      for (int u = 0; u < 8; ++u) {
        const int k = 8 * w + u;
        const unsigned long long c = k < len ? (unsigned long long)code[q0 + k] : 0xFFull;
        word |= c << (8 * u);
      }
    }
    slot(w) = word;
  }
  for (int k = 0; k < L; ++k) slot(W + k) = k < len ? (unsigned long long)__double_as_longlong(val[q0 + k]) : 0ull;
}

void csr_free_sell(khip_csr *A) {
  (void)hipFree(A->sell); (void)hipFree(A->sell_off); (void)hipFree(A->sell_c4);
  A->sell = nullptr; A->sell_off = nullptr; A->sell_c4 = nullptr; A->sell_pair = 0;
  A->sell_units = 0; A->sell_total_units = 0; A->sell_state = 0;
}

void csr_free_sell32(khip_csr *A) {
  (void)hipFree(A->sell32); (void)hipFree(A->sell32_off);
  A->sell32 = nullptr; A->sell32_off = nullptr;
  A->sell32_units = 0; A->sell32_total_units = 0; A->sell32_state = 0; A->sell32_pair = 0;
}

// cols32 = false: code words (needs the 8-bit codes); true: int32 column words
static int build_sell_form(khip_ctx *ctx, khip_csr *A, bool cols32) {
  if (cols32) csr_free_sell32(A); else csr_free_sell(A);
  int &state = cols32 ? A->sell32_state : A->sell_state;
  unsigned long long *&words = cols32 ? A->sell32 : A->sell;
  uint32_t *&offs = cols32 ? A->sell32_off : A->sell_off;
  state = -1;
  const int64_t m = A->m;
  if (m == 0 || A->nnz == 0 || A->max_row_nnz > 64) return KHIP_OK;
  if (!cols32 && (A->code_state != 1 || A->code_bits != 8 || A->code_T > 255)) return KHIP_OK;
  const bool narrow = !cols32 && ctx->tune.spmv_sell_narrow && A->code_T <= 15 && A->max_row_nnz <= 8;
  const bool pair = !narrow && ctx->tune.spmv_sell_pair;
  // int32 columns: the pair layout costs 96 instead of 88 B per 7-point row and measured no faster (plain 2.36 against 2.27 ms, fused 2.44
  // against 2.41 at 512^3, profiles/r06aw): only with spmv_sell_pair = 2
  const int mode = cols32 ? ((pair && ctx->tune.spmv_sell_pair >= 2) ? 4 : 1) : (narrow ? 2 : (pair ? (A->max_row_nnz <= 8 ? 3 : 5) : 0));
  const int64_t slices = (m + 63) / 64;
  int32_t *units_d = nullptr;
  KHIP_CHECK_HIP(hipMalloc(&units_d, sizeof(int32_t) * (size_t)slices));
  struct Scratch { int32_t *&p; ~Scratch() { (void)hipFree(p); } } scratch{units_d};
  hipLaunchKernelGGL(sell_units_kernel, dim3((unsigned)((slices + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, A->rowptr, m, slices, units_d, mode);
  KHIP_CHECK_HIP(hipGetLastError());
  std::vector<int32_t> units((size_t)slices);
  KHIP_CHECK_HIP(hipMemcpyAsync(units.data(), units_d, sizeof(int32_t) * (size_t)slices, hipMemcpyDeviceToHost, ctx->stream));
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  int64_t total = 0, slots = 0;
  int umax = 0;
  for (int32_t u : units) { total += u; const int W = sell_head_words(u, mode); slots += (int64_t)(u - W) * 64; umax = u > umax ? u : umax; }
  (void)slots;
  const double ref_bytes = (cols32 ? 12.0 : 9.0) * (double)A->nnz + 4.0 * (double)m;    // the CSR stream this copy replaces
  if (512.0 * (double)total > kSellMaxPad * ref_bytes + 65536.0) return KHIP_OK;       // too much padding: stays on the CSR stream
  const bool uniform = (double)umax * (double)slices <= kSellUniformPad * (double)total + 8.0;
  if (uniform) total = (int64_t)umax * slices;
  if (total >= ((int64_t)1 << 32)) return KHIP_OK;
  bool keep = false;
  struct Guard { khip_csr *A; bool cols32; int &state; bool &keep;
                 ~Guard() { if (!keep) { if (cols32) csr_free_sell32(A); else csr_free_sell(A); state = -1; } } } guard{A, cols32, state, keep};
  if (!uniform) {
    std::vector<uint32_t> off((size_t)slices + 1);
    uint32_t run = 0;
    for (int64_t s = 0; s < slices; ++s) { off[(size_t)s] = run; run += (uint32_t)units[(size_t)s]; }
    off[(size_t)slices] = run;
    KHIP_CHECK_HIP(hipMalloc(&offs, sizeof(uint32_t) * ((size_t)slices + 1)));
    KHIP_CHECK_HIP(hipMemcpyAsync(offs, off.data(), sizeof(uint32_t) * ((size_t)slices + 1), hipMemcpyHostToDevice, ctx->stream));
    KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));                                  // `off` dies with this scope
  }
  KHIP_CHECK_HIP(hipMalloc(&words, sizeof(unsigned long long) * 64 * (size_t)(total + 1)));
  if (narrow) KHIP_CHECK_HIP(hipMalloc(&A->sell_c4, sizeof(uint32_t) * 64 * (size_t)slices));
  hipLaunchKernelGGL(sell_fill_kernel, dim3((unsigned)((slices * 64 + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, A->rowptr, A->val,
                     (const uint8_t *)A->code, cols32 ? A->col : (const int32_t *)nullptr, m, slices, offs, uniform ? umax : 0, words, narrow ? A->sell_c4 : (uint32_t *)nullptr, mode);
  KHIP_CHECK_HIP(hipGetLastError());
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  keep = true;
  (cols32 ? A->sell32_units : A->sell_units) = uniform ? umax : 0;
  (cols32 ? A->sell32_total_units : A->sell_total_units) = total;
  if (cols32) A->sell32_pair = mode == 4 ? 2 : 0; else A->sell_pair = mode == 3 ? 1 : (mode == 5 ? 2 : 0);
  state = 1;
  return KHIP_OK;
}

int csr_build_sell(khip_ctx *ctx, khip_csr *A) { return build_sell_form(ctx, A, false); }
int csr_build_sell32(khip_ctx *ctx, khip_csr *A) { return build_sell_form(ctx, A, true); }

void csr_free_codes(khip_csr *A) {
  csr_free_sell(A);                          // the sliced form carries the codes
  csr_free_sell32(A);                        // ... and its int32 twin the columns (called wherever the columns change, and by destroy)
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
