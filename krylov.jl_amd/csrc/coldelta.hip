// coldelta.hip -- block-delta column stream for the stream SpMV (spmv_delta_kernel, spmv.hip).
//
// The dictionary codes of colcode.hip need an operator with at most 2048 distinct diagonals -- a stencil.  This is the
// column stream for everything else with BAND LOCALITY (the SuiteSparse-type operators of benchmark/cg_bmark.jl:29-54; here
// the "banded + random" benchmark operator): rows are taken in blocks of R (the row block of the stream kernel), every
// block has a base column, and an entry whose column lies in [base, base + 2^bits - 2] keeps only the difference --
// one byte (bits = 8) or two (bits = 16) instead of the int32.  The entries that do not fit (long-range links, dense
// rows) are ESCAPES: their code is the all-ones value and the block owns a list of (position in the block, int32
// column) pairs, 6 bytes each, which the kernel patches in after the main pass.  col = base + code is exact integer
// arithmetic and the products are summed per row in stored order exactly as before: y stays BIT-IDENTICAL to the
// serial loop for any matrix.  The handle still takes and keeps plain CSR (src/krylov_utils.jl:305 kmul! boundary);
// this is only how the stream kernel reads the columns: 8 + 1 (2) bytes per entry + 6 per escape instead of 12.
//
// base = max(0, first row of the block - H) with H = (2^bits - 1 - R) / 2 (band centred on the diagonal) -- stored per
// block, so the kernel does not depend on the rule.  Construction on the device at the first product that can use it:
// one workgroup per block counts its escapes for the 8-bit and the 16-bit form, the host takes the prefix sums and the
// cheaper form (or neither, when it would not save at least a sixth of the column bytes), a second pass writes codes
// and escape lists.  Deterministic: no atomics on data.
#include <vector>

#include "spmv_common.hpp"

namespace khip {

constexpr int kDeltaPad = 64;      // zeroed codes behind the last entry (16-byte lane loads may overrun)

__host__ __device__ inline int32_t delta_base(int64_t r0, int rows, int bits) {
  const int64_t H = (((int64_t)1 << bits) - 1 - rows) / 2;
  const int64_t b = r0 - (H > 0 ? H : 0);
  return (int32_t)(b > 0 ? b : 0);
}

// escapes of every row block for the (rows8, 8-bit) and the (rows16, 16-bit) candidate; one thread per row
__global__ __launch_bounds__(kBlock) void delta_count_kernel(const int32_t *rowptr, const int32_t *col, int64_t m, int rows8,
                                                             int rows16, int *cnt8, int *cnt16, int *too_long) {
  const int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (row >= m) return;
  const int64_t r8 = row / rows8 * rows8, r16 = row / rows16 * rows16;
  const int32_t b8 = delta_base(r8, rows8, 8), b16 = delta_base(r16, rows16, 16);
  int e8 = 0, e16 = 0;
  for (int32_t q = rowptr[row]; q < rowptr[row + 1]; ++q) {
    const int64_t c = col[q];
    e8 += (c < b8 || c - b8 >= 255);
    e16 += (c < b16 || c - b16 >= 65535);
  }
  if (e8) atomicAdd(&cnt8[row / rows8], e8);          // integer counts: the order of the additions does not matter
  if (e16) atomicAdd(&cnt16[row / rows16], e16);
  // a block's entries must be addressable by 16 bits (esc_pos): the first row of every block checks its block
  if (row % rows8 == 0) {
    const int64_t hi = row + rows8 < m ? row + rows8 : m;
    if (rowptr[hi] - rowptr[row] > 65535) atomicMax(too_long, 1);
  }
  if (row % rows16 == 0) {
    const int64_t hi = row + rows16 < m ? row + rows16 : m;
    if (rowptr[hi] - rowptr[row] > 65535) atomicMax(too_long, 1);
  }
}

// codes + escape lists; one workgroup (`rows` threads) per row block, thread t = row t of the block
template <typename CODE>
__global__ void delta_assign_kernel(const int32_t *rowptr, const int32_t *col, int64_t m, int rows, const int32_t *esc_ptr,
                                    CODE *code, int32_t *base_out, uint16_t *esc_pos, int32_t *esc_col) {
  constexpr int BITS = 8 * (int)sizeof(CODE);
  constexpr int64_t ESC = ((int64_t)1 << BITS) - 1;
  __shared__ int s_cnt[kBlock + 1];
  const int t = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows, row = r0 + t;
  const int32_t base = delta_base(r0, rows, BITS);
  const bool live = t < rows && row < m;
  int e = 0;
  if (live)
    for (int32_t q = rowptr[row]; q < rowptr[row + 1]; ++q) {
      const int64_t c = col[q];
      e += (c < base || c - base >= ESC);
    }
  s_cnt[t + 1] = e;
  if (t == 0) { s_cnt[0] = 0; base_out[blockIdx.x] = base; }
  __syncthreads();
  if (t == 0) for (int i = 1; i <= rows; ++i) s_cnt[i] += s_cnt[i - 1];   // <= 256 rows: a serial scan is fine for a one-off pass
  __syncthreads();
  if (!live) return;
  int64_t out = (int64_t)esc_ptr[blockIdx.x] + s_cnt[t];
  const int32_t s0 = rowptr[r0];
  for (int32_t q = rowptr[row]; q < rowptr[row + 1]; ++q) {
    const int64_t c = col[q];
    if (c < base || c - base >= ESC) {
      code[q] = (CODE)ESC;
      esc_pos[out] = (uint16_t)(q - s0);
      esc_col[out] = (int32_t)c;
      ++out;
    } else {
      code[q] = (CODE)(c - base);
    }
  }
}

void csr_free_delta(khip_csr *A) {
  (void)hipFree(A->dcode); (void)hipFree(A->dbase); (void)hipFree(A->desc_ptr); (void)hipFree(A->desc_pos); (void)hipFree(A->desc_col);
  A->dcode = nullptr; A->dbase = nullptr; A->desc_ptr = nullptr; A->desc_pos = nullptr; A->desc_col = nullptr;
  A->delta_state = 0; A->delta_bits = 0; A->delta_rows = 0; A->delta_esc = 0;
}

// rows: the row block the stream kernel uses for this operator (32 .. 256).  Sets A->delta_state to 1 or -1.
int csr_build_delta(khip_ctx *ctx, khip_csr *A, int rows) {
  csr_free_delta(A);
  A->delta_state = -1;
  const int64_t m = A->m;
  if (m == 0 || A->nnz == 0 || rows < 32 || rows > 256) return KHIP_OK;
  const int rows16 = rows, rows8 = rows < 64 ? rows : 64;       // 8 bits reach 255 columns: keep the block narrow
  const int64_t nb8 = (m + rows8 - 1) / rows8, nb16 = (m + rows16 - 1) / rows16;
  int *cnt = nullptr;
  KHIP_CHECK_HIP(hipMalloc(&cnt, sizeof(int) * (size_t)(nb8 + nb16 + 1)));
  bool keep = false;
  struct Scratch {
    int *&cnt; khip_csr *A; bool &keep;
    ~Scratch() { (void)hipFree(cnt); if (!keep) { csr_free_delta(A); A->delta_state = -1; } }
  } scratch{cnt, A, keep};
  KHIP_CHECK_HIP(hipMemsetAsync(cnt, 0, sizeof(int) * (size_t)(nb8 + nb16 + 1), ctx->stream));
  const unsigned grid = (unsigned)((m + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(delta_count_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, A->rowptr, A->col, m, rows8, rows16, cnt,
                     cnt + nb8, cnt + nb8 + nb16);
  KHIP_CHECK_HIP(hipGetLastError());
  std::vector<int> h((size_t)(nb8 + nb16 + 1));
  KHIP_CHECK_HIP(hipMemcpyAsync(h.data(), cnt, sizeof(int) * h.size(), hipMemcpyDeviceToHost, ctx->stream));
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  if (h[(size_t)(nb8 + nb16)]) return KHIP_OK;                  // a block with more than 65535 entries
  int64_t esc8 = 0, esc16 = 0;
  for (int64_t i = 0; i < nb8; ++i) esc8 += h[(size_t)i];
  for (int64_t i = 0; i < nb16; ++i) esc16 += h[(size_t)(nb8 + i)];
  // column bytes of the three forms; ctx option spmv_delta: 8 / 16 force a width, 2 = take the cheaper one whatever it saves
  const int64_t by8 = A->nnz + 6 * esc8, by16 = 2 * A->nnz + 6 * esc16, by32 = 4 * A->nnz;
  int bits = by8 <= by16 ? 8 : 16;
  if (ctx->tune.spmv_delta == 8 || ctx->tune.spmv_delta == 16) bits = ctx->tune.spmv_delta;
  const int64_t by = bits == 8 ? by8 : by16, esc = bits == 8 ? esc8 : esc16;
  if (ctx->tune.spmv_delta < 2 && 6 * by > 5 * by32) return KHIP_OK;             // saves less than a sixth: stay on int32
  if (esc >= ((int64_t)1 << 31) - 1) return KHIP_OK;
  const int R = bits == 8 ? rows8 : rows16;
  const int64_t nb = bits == 8 ? nb8 : nb16;
  std::vector<int32_t> ptr((size_t)nb + 1, 0);
  const int *c = h.data() + (bits == 8 ? 0 : nb8);
  for (int64_t i = 0; i < nb; ++i) ptr[(size_t)i + 1] = ptr[(size_t)i] + c[i];
  KHIP_CHECK_HIP(hipMalloc(&A->desc_ptr, sizeof(int32_t) * (size_t)(nb + 1)));
  KHIP_CHECK_HIP(hipMalloc(&A->dbase, sizeof(int32_t) * (size_t)nb));
  KHIP_CHECK_HIP(hipMalloc(&A->desc_pos, sizeof(uint16_t) * (size_t)(esc + 8)));
  KHIP_CHECK_HIP(hipMalloc(&A->desc_col, sizeof(int32_t) * (size_t)(esc + 8)));
  const size_t cb = (size_t)(A->nnz + kDeltaPad) * (size_t)(bits / 8);
  KHIP_CHECK_HIP(hipMalloc(&A->dcode, cb));
  KHIP_CHECK_HIP(hipMemsetAsync((char *)A->dcode + (size_t)A->nnz * (size_t)(bits / 8), 0, (size_t)kDeltaPad * (size_t)(bits / 8), ctx->stream));
  KHIP_CHECK_HIP(hipMemcpyAsync(A->desc_ptr, ptr.data(), sizeof(int32_t) * ptr.size(), hipMemcpyHostToDevice, ctx->stream));
  if (bits == 8)
    hipLaunchKernelGGL(delta_assign_kernel<uint8_t>, dim3((unsigned)nb), dim3(R < 64 ? 64 : R), 0, ctx->stream, A->rowptr, A->col, m, R,
                       A->desc_ptr, (uint8_t *)A->dcode, A->dbase, A->desc_pos, A->desc_col);
  else
    hipLaunchKernelGGL(delta_assign_kernel<uint16_t>, dim3((unsigned)nb), dim3(R < 64 ? 64 : R), 0, ctx->stream, A->rowptr, A->col, m, R,
                       A->desc_ptr, (uint16_t *)A->dcode, A->dbase, A->desc_pos, A->desc_col);
  KHIP_CHECK_HIP(hipGetLastError());
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));            // also keeps `ptr` alive until the upload is done
  keep = true;
  A->delta_bits = bits;
  A->delta_rows = R;
  A->delta_esc = esc;
  A->delta_state = 1;
  return KHIP_OK;
}

}  // namespace khip
