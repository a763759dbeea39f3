// panel.hip -- tall-skinny panel kernels of block_gmres! (src/block_gmres.jl:244-247,259,324-326) on
// the FP64 matrix cores of gfx950.
//
// Panel layout in HBM: n_pad x p ROW-MAJOR (p contiguous doubles per row, n_pad = n rounded up to 16,
// padding rows are zero).  With p = 16 a panel row is one 128-byte line: the SpMM gather of a row of
// X is a single line and four consecutive rows are exactly one v_mfma_f64_16x16x4_f64 operand (64
// lanes x 8 B, fully coalesced).  The reference keeps n x p column-major Matrix{Float64}; conversion
// happens once at the boundary (khip_panel_from/to_colmajor).
//
// All three panel products are HBM-bound (AI = p/8 = 2 flop/B at p = 16): V^T Q reads two panels,
// Q - V Psi reads two and writes one.  The MFMA only keeps the FP64 pipe from co-limiting:
//   Psi = V^T Q : D(16x16) += A(16x4) B(4x16) with A[i][k] = V[r0+k][i], B[k][j] = Q[r0+k][j]
//                 => lane l feeds V[r0*p + l] and Q[r0*p + l] (p = 16): one MFMA per 4 rows.
//   (the two are also fused into one pass per Gram-Schmidt step: panel_nn_tn_kernel / khip_panel_mgs)
//   Q += a V Psi: D(16 rows x 16 cols) += A(16 rows x 4) B(4 x 16): A[i][k] = V[r0+i][4kk+k],
//                 B[k][j] = Psi[4kk+k][j] held in registers; C/D lane layout of the f64 MFMA:
//                 col = lane & 15, row = (lane >> 4) + 4 * reg  (cdna_hip_programming.md section 3).
// p up to 32 is handled with 2 x 2 tiles of 16 columns; p that is not a multiple of 16 zero-fills.
#include <vector>

#include "device_reduce.hpp"

namespace khip {

typedef double dbl4 __attribute__((ext_vector_type(4)));
typedef double dbl2 __attribute__((ext_vector_type(2)));

// Panels far larger than the caches are streams: non-temporal accesses lift a streaming kernel here from ~5.7 to 6.3-6.7 TB/s
// (DESIGN.md 3.1).  `nt` is wave-uniform (a launch parameter): the branch costs a scalar compare.
template <typename T>
__device__ __forceinline__ T pld(const T *p, bool nt) { return nt ? __builtin_nontemporal_load(p) : *p; }
template <typename T>
__device__ __forceinline__ void pst(T *p, T v, bool nt) { if (nt) __builtin_nontemporal_store(v, p); else *p = v; }

#ifndef KHIP_ROWS_PER_WAVE_TN
#define KHIP_ROWS_PER_WAVE_TN 256      // experiment builds: -DKHIP_ROWS_PER_WAVE_TN=64 / 128 (tools/panel_front_ab.sh)
#endif
constexpr int kRowsPerWaveTN = KHIP_ROWS_PER_WAVE_TN;   // rows folded by one wave of the V^T Q kernel (64 MFMAs per tile pair)

// ---------------------------------------------------------------- layout conversion ------
// in: column-major n x p (ld = n); out: row-major n_pad x p (ld = p).  256 rows per workgroup via LDS.
__global__ __launch_bounds__(kBlock) void panel_from_colmajor_kernel(int64_t n, int p, const double *in, double *out) {
  extern __shared__ double s_tile[];                 // [256][p + 1]
  const int64_t r0 = (int64_t)blockIdx.x * kBlock;
  const int tid = threadIdx.x;
  for (int c = 0; c < p; ++c) {
    const int64_t r = r0 + tid;
    s_tile[tid * (p + 1) + c] = (r < n) ? in[(int64_t)c * n + r] : 0.0;
  }
  __syncthreads();
  const int64_t base = r0 * p;
  for (int i = tid; i < kBlock * p; i += kBlock) {
    const int rl = i / p, c = i % p;
    if (r0 + rl < n) out[base + i] = s_tile[rl * (p + 1) + c];
  }
}

__global__ __launch_bounds__(kBlock) void panel_to_colmajor_kernel(int64_t n, int p, const double *in, double *out) {
  extern __shared__ double s_tile[];
  const int64_t r0 = (int64_t)blockIdx.x * kBlock;
  const int tid = threadIdx.x;
  const int64_t base = r0 * p;
  for (int i = tid; i < kBlock * p; i += kBlock) {
    const int rl = i / p, c = i % p;
    s_tile[rl * (p + 1) + c] = (r0 + rl < n) ? in[base + i] : 0.0;
  }
  __syncthreads();
  for (int c = 0; c < p; ++c) {
    const int64_t r = r0 + tid;
    if (r < n) out[(int64_t)c * n + r] = s_tile[tid * (p + 1) + c];
  }
}

// The NT x NT accumulator tiles of a workgroup's four waves -> ONE partial tile per workgroup: the waves park their tiles in
// LDS and thread t adds element t of the four in wave order (a fixed order: deterministic), then stores it.  A quarter of the
// partial traffic and of the first level of panel_tn_reduce_kernel (per-wave tiles were 80 MB per product at 10 M rows, 46 us
// of reduction behind every 0.3-0.9 ms kernel).  Every thread of the workgroup must call it (barrier).
template <int NT>
__device__ __forceinline__ void tn_publish(const dbl4 (&tn)[NT][NT], double *partials) {
  __shared__ double s_tn[kWavesPerBlock][NT * NT * 256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // tile (a, b), register g of lane l holds Psi[16a + (l >> 4) + 4g][16b + (l & 15)]
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) s_tn[w][(a * NT + b) * 256 + g * 64 + lane] = tn[a][b][g];
  __syncthreads();
  double *out = partials + (size_t)blockIdx.x * (NT * NT * 256);
#pragma unroll
  for (int q = 0; q < NT * NT; ++q) {
    const int e = q * 256 + threadIdx.x;
    double sum = s_tn[0][e];
#pragma unroll
    for (int v = 1; v < kWavesPerBlock; ++v) sum = sum + s_tn[v][e];
    out[e] = sum;
  }
}

// ---------------------------------------------------------------- Psi = V^T Q -------------
// Each wave folds kRowsPerWaveTN consecutive rows into NT x NT accumulator tiles; the workgroup's four are added in wave order
// (tn_publish) into partials[workgroup][NT*NT][256]; panel_tn_reduce_kernel sums those in workgroup order.
template <int NT>
__global__ __launch_bounds__(kBlock) void panel_gemm_tn_kernel(int64_t n_pad, int p, const double *V, const double *Q,
                                                               double *partials) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t row_begin = wid * kRowsPerWaveTN;
  const int i = lane & 15, k = lane >> 4;
  dbl4 acc[NT][NT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = dbl4{0.0, 0.0, 0.0, 0.0};
  if (row_begin < n_pad) {
    const int64_t row_end = (row_begin + kRowsPerWaveTN < n_pad) ? row_begin + kRowsPerWaveTN : n_pad;
    constexpr int UN = 8;                            // MFMA steps whose loads are issued together
    for (int64_t r = row_begin; r < row_end; r += 4 * UN) {
      double va[UN][NT], qb[UN][NT];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int64_t row = r + 4 * u + k;
        const bool rok = row < row_end;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int col = 16 * t + i;
          const bool ok = rok && col < p;
          va[u][t] = ok ? V[row * p + col] : 0.0;
          qb[u][t] = ok ? Q[row * p + col] : 0.0;
        }
      }
#pragma unroll
      for (int u = 0; u < UN; ++u)
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
          for (int b = 0; b < NT; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[u][a], qb[u][b], acc[a][b], 0, 0, 0);
    }
  }
  tn_publish<NT>(acc, partials);
}

// Fixed-order tree over the per-workgroup tiles: each workgroup (x = group, y = tile pair) sums up to
// kTnFan consecutive tiles; repeated until one is left, which is scattered into Psi_dev
// (p x p column-major).  The order depends only on the wave count: deterministic.
constexpr int kTnFan = 64;
template <int NT>
__global__ __launch_bounds__(kBlock) void panel_tn_reduce_kernel(int64_t count, int p, const double *in, double *out,
                                                                 double *Psi_dev) {
  const int tile = blockIdx.y;                      // a * NT + b
  const int t = threadIdx.x;                        // = g * 64 + lane
  const int64_t w0 = (int64_t)blockIdx.x * kTnFan;
  const int64_t w1 = (w0 + kTnFan < count) ? w0 + kTnFan : count;
  double s = 0.0;
  // sixteen loads in flight, added in tile order (the sum is the same left-to-right chain whatever the unrolling); one load
  // at a time made a level of 64 tiles cost 64 dependent round trips (33 us at 10 M rows)
  const double *src = in + (size_t)tile * 256 + t;
  int64_t w = w0;
  for (; w + 16 <= w1; w += 16) {
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = src[(size_t)(w + u) * (NT * NT * 256)];
#pragma unroll
    for (int u = 0; u < 16; ++u) s += v[u];
  }
  for (; w < w1; ++w) s += src[(size_t)w * (NT * NT * 256)];
  if (Psi_dev == nullptr) {
    out[(size_t)blockIdx.x * (NT * NT * 256) + tile * 256 + t] = s;
  } else {                                           // last level (gridDim.x == 1)
    const int a = tile / NT, b = tile % NT;
    const int g = t >> 6, lane = t & 63;
    const int row = 16 * a + (lane >> 4) + 4 * g, col = 16 * b + (lane & 15);
    if (row < p && col < p) Psi_dev[(size_t)col * p + row] = s;
  }
}

// ---------------------------------------------------------------- Q = beta Q + alpha V Psi --
// One wave per 16-row tile of the panel; V may alias Q (in-place Q <- Q * Psi when beta == 0):
// every A fragment of the tile is loaded before the first store.
template <int NT>
__global__ __launch_bounds__(kBlock) void panel_gemm_nn_kernel(int64_t n_pad, int p, double alpha, const double *V,
                                                               const double *Psi_dev, double beta, double *Q) {
  const int lane = threadIdx.x & 63;
  const int64_t tile = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t r0 = tile * 16;
  if (r0 >= n_pad) return;
  const int i = lane & 15, k = lane >> 4;
  constexpr int KK = NT * 4;                         // k-steps of 4 covering up to 16 * NT columns of V
  // B fragments: Psi[4kk + k][16b + i]  (column-major p x p)
  double bf[KK][NT];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk)
#pragma unroll
    for (int b = 0; b < NT; ++b) {
      const int prow = 4 * kk + k, pcol = 16 * b + i;
      bf[kk][b] = (prow < p && pcol < p) ? Psi_dev[(size_t)pcol * p + prow] : 0.0;
    }
  // A fragments: V[r0 + i][4kk + k]
  double af[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    const int vcol = 4 * kk + k;
    af[kk] = (vcol < p) ? V[(r0 + i) * p + vcol] : 0.0;
  }
  dbl4 cin[NT], acc[NT];
#pragma unroll
  for (int b = 0; b < NT; ++b) {
    acc[b] = dbl4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int col = 16 * b + i;
      const int64_t row = r0 + k + 4 * g;
      cin[b][g] = (beta != 0.0 && col < p) ? Q[row * p + col] : 0.0;
    }
  }
#pragma unroll
  for (int kk = 0; kk < KK; ++kk)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk], bf[kk][b], acc[b], 0, 0, 0);
#pragma unroll
  for (int b = 0; b < NT; ++b)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int col = 16 * b + i;
      const int64_t row = r0 + k + 4 * g;
      if (col < p) Q[row * p + col] = fma(alpha, acc[b][g], beta * cin[b][g]);
    }
}

// ---------------------------------------------------------------- Q -= V_i Psi_i fused with Psi_{i+1} = V_{i+1}^T Q ----
// One step of the block Gram-Schmidt sweep (src/block_gmres.jl:244-247) in one pass: the D tile of the first product
// -- lane (i, k), register g = Q_new[r0 + k + 4g][16b + i] -- is exactly the B operand the second product needs for
// its k-step u = g, so the updated rows never leave the registers between the two.  A wave covers the same
// kRowsPerWaveTN rows, in the same order, as a wave of panel_gemm_tn_kernel and the update is the expression of
// panel_gemm_nn_kernel: Q and Psi_{i+1} are bit-identical to the two separate kernels; four panel passes instead of five.
// SELF: the second product is the Gram matrix of the updated panel itself (Vn unused): round 1 of the CholeskyQR
// (Q <- Q R^-1) fused with G = Q^T Q of round 2.
#ifndef KHIP_NN_TN_WAVES
#define KHIP_NN_TN_WAVES 1             // experiment builds: minimum waves per SIMD the register allocation must leave room for
#endif
template <int NT, int UNT, bool SELF>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(KHIP_NN_TN_WAVES, 8))) void panel_nn_tn_kernel(int64_t n_pad, int p, double alpha, const double *Vi,
                                                             const double *Psi_dev, double beta, const double *Vn, double *Q,
                                                             double *partials, int a_stage) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t row_begin = wid * kRowsPerWaveTN;
  const int i = lane & 15, k = lane >> 4;
  constexpr int KK = NT * 4;
  __shared__ __attribute__((aligned(16))) char a_lds[NT == 1 ? kWavesPerBlock * UNT * 2048 : 16];
  const bool a_via_lds = NT == 1 && p == 16 && (a_stage & 1) != 0;
  const bool snt = (a_stage & 2) != 0;                    // streaming (non-temporal) accesses to the panels
  dbl4 tn[NT][NT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) tn[a][b] = dbl4{0.0, 0.0, 0.0, 0.0};
  if (row_begin < n_pad) {
    const int64_t row_end = (row_begin + kRowsPerWaveTN < n_pad) ? row_begin + kRowsPerWaveTN : n_pad;
    double bf[KK][NT];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
      for (int b = 0; b < NT; ++b) {
        const int prow = 4 * kk + k, pcol = 16 * b + i;
        bf[kk][b] = (prow < p && pcol < p) ? Psi_dev[(size_t)pcol * p + prow] : 0.0;
      }
    for (int64_t r = row_begin; r < row_end; r += 16 * UNT) {
      double af[UNT][KK], va[UNT][4][NT];
      dbl4 cin[UNT][NT];
#pragma unroll
      for (int t = 0; t < UNT; ++t) {
        const int64_t r0 = r + 16 * t;
        const bool tok = r0 < row_end;
        if (NT == 1 && a_via_lds) {
          // The A operand wants V_i[r0 + i][4 kk + k]: 16 lanes of one k read 16 DIFFERENT rows, i.e. every one of the four
          // loads touches all 16 lines of the 2 KB tile.  Instead the tile is fetched as what it is in memory -- 128
          // consecutive 16-byte pieces, two fully coalesced loads per lane -- parked in the wave's own LDS slice with the pieces
          // of row i rotated by i (XOR swizzle: the column reads below are then free of bank conflicts) and read back in operand
          // order.  Same values into the same MFMAs: bit-identical.  Only this wave touches the slice: no barrier.
          char *slice = a_lds + (size_t)((threadIdx.x >> 6) * UNT + t) * 2048;
          if (tok) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int c16 = lane + 64 * h, row = c16 >> 3, c = c16 & 7;
              const dbl2 piece = pld(reinterpret_cast<const dbl2 *>(Vi + (r0 + row) * 16 + 2 * c), snt);
              *reinterpret_cast<dbl2 *>(slice + row * 128 + ((c ^ (row & 7)) << 4)) = piece;
            }
          }
#pragma unroll
          for (int kk = 0; kk < KK; ++kk) {
            const int vcol = 4 * kk + k;
            af[t][kk] = tok ? *reinterpret_cast<const double *>(slice + i * 128 + (((vcol >> 1) ^ (i & 7)) << 4) + ((vcol & 1) << 3)) : 0.0;
          }
        } else {
#pragma unroll
          for (int kk = 0; kk < KK; ++kk) {
            const int vcol = 4 * kk + k;
            af[t][kk] = (tok && vcol < p) ? pld(Vi + (r0 + i) * p + vcol, snt) : 0.0;
          }
        }
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int col = 16 * b + i;
            cin[t][b][g] = (tok && beta != 0.0 && col < p) ? pld(Q + (r0 + k + 4 * g) * p + col, snt) : 0.0;
          }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int a = 0; a < NT; ++a) {
            const int col = 16 * a + i;
            va[t][u][a] = (!SELF && tok && col < p) ? pld(Vn + (r0 + 4 * u + k) * p + col, snt) : 0.0;
          }
      }
#pragma unroll
      for (int t = 0; t < UNT; ++t) {
        const int64_t r0 = r + 16 * t;
        if (r0 >= row_end) break;
        dbl4 acc[NT], wnew[NT];
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[b] = dbl4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
          for (int b = 0; b < NT; ++b) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[t][kk], bf[kk][b], acc[b], 0, 0, 0);
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int col = 16 * b + i;
            wnew[b][g] = col < p ? fma(alpha, acc[b][g], beta * cin[t][b][g]) : 0.0;
            if (col < p) pst(Q + (r0 + k + 4 * g) * p + col, wnew[b][g], snt);
          }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b)
              tn[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(SELF ? wnew[a][u] : va[t][u][a], wnew[b][u], tn[a][b], 0, 0, 0);
      }
    }
  }
  tn_publish<NT>(tn, partials);
}

// ---------------------------------------------------------------- X = beta X + sum_i V_i Y_i ----
// The solution update of block_gmres! (src/block_gmres.jl:324-326: k products mul!(X, V_i, Y_i, 1, 1)) in one pass over X:
// per 16-row tile the k products are applied in the order i = 0..k-1 with the expression of panel_gemm_nn_kernel
// (x <- fma(1, acc_i, 1 * x); the first one with the caller's beta), so X is bit-identical to the k separate calls;
// k + 2 panel passes instead of 3 k.
constexpr int kMultiNN = 32;
struct MultiNNArgs { const double *v[kMultiNN]; };

template <int NT>
__global__ __launch_bounds__(kBlock) void panel_multi_nn_kernel(int64_t n_pad, int p, int k, MultiNNArgs V, const double *Y_dev,
                                                                double beta, double *X) {
  const int lane = threadIdx.x & 63;
  const int64_t tile = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t r0 = tile * 16;
  if (r0 >= n_pad) return;
  const int i = lane & 15, kq = lane >> 4;
  constexpr int KK = NT * 4;
  dbl4 x[NT];
#pragma unroll
  for (int b = 0; b < NT; ++b)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int col = 16 * b + i;
      x[b][g] = (beta != 0.0 && col < p) ? X[(r0 + kq + 4 * g) * p + col] : 0.0;
    }
  double bcur = beta;
  for (int j = 0; j < k; ++j) {
    const double *Vj = V.v[j];
    const double *Yj = Y_dev + (size_t)j * 1024;
    double af[KK], bf[KK][NT];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      const int vcol = 4 * kk + kq;
      af[kk] = (vcol < p) ? Vj[(r0 + i) * p + vcol] : 0.0;
#pragma unroll
      for (int b = 0; b < NT; ++b) {
        const int prow = 4 * kk + kq, pcol = 16 * b + i;
        bf[kk][b] = (prow < p && pcol < p) ? Yj[(size_t)pcol * p + prow] : 0.0;
      }
    }
    dbl4 acc[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[b] = dbl4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
      for (int b = 0; b < NT; ++b) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk], bf[kk][b], acc[b], 0, 0, 0);
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) x[b][g] = fma(1.0, acc[b][g], bcur * x[b][g]);
    bcur = 1.0;
  }
#pragma unroll
  for (int b = 0; b < NT; ++b)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int col = 16 * b + i;
      if (col < p) X[(r0 + kq + 4 * g) * p + col] = x[b][g];
    }
}

// Same update with the k factor blocks held in LDS and T consecutive 16-row tiles per wave.  The kernel above re-reads the
// 4 B-operand values of every factor from global memory for every tile: 8 k + 8 vector-memory instructions per 2 (k + 2) KB
// of panel traffic, and on this chip a wave's vector-memory instruction costs ~17 cycles of the CU's texture path whatever
// its width (tools/l2bench.hip: 8 B per lane reach half the bytes per clock of 16 B per lane) -- the kernel ran at 3.3 TB/s.
// Here every workgroup copies the factors once (transposed, so that the B-operand reads are lane-contiguous = free of bank
// conflicts) and each wave walks T tiles: 4 k + 8 instructions per tile.  Same A / B operand values into the same MFMA
// chain in the same order => X bit-identical to panel_multi_nn_kernel and to the k khip_panel_gemm_nn calls.
// With k = 1 and a general alpha it is also Q <- beta Q + alpha V Psi (panel_gemm_nn_kernel's expression; V may alias X:
// the A fragments of a tile are loaded before its first store).
template <int NT>
__global__ __launch_bounds__(kBlock) void panel_multi_nn_lds_kernel(int64_t n_pad, int p, int k, MultiNNArgs V, const double *Y_dev,
                                                                    double alpha, double beta, double *X, int T) {
  extern __shared__ double s_Y[];                    // [k][p][p]: s_Y[j][prow][pcol] = Y_j[pcol][prow] (Y_j column-major)
  const int pp = p * p;
  for (int idx = threadIdx.x; idx < k * pp; idx += kBlock) {
    const int j = idx / pp, r = idx - j * pp, prow = r / p, pcol = r - prow * p;
    s_Y[idx] = Y_dev[(size_t)j * 1024 + (size_t)pcol * p + prow];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int i = lane & 15, kq = lane >> 4;
  constexpr int KK = NT * 4;
  const int64_t tile0 = ((int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)) * T;
  // The A fragments of the NEXT factor (or of the next tile's first factor) are requested before the MFMA chain of the current
  // one (round 5): with k a run-time count the loop is not unrolled, and a wave used to have the 2 KB of ONE factor tile in flight
  // at a time -- k = 5 ran at 0.61 of peak where the one-factor form reaches the 1R + 1W floor (0.72).  Same operand values into
  // the same MFMA chain in the same order: X stays bit-identical.
  auto load_a = [&](int64_t r0, int j, double (&a)[KK]) {
    const double *Vj = V.v[j];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      const int vcol = 4 * kk + kq;
      a[kk] = (vcol < p) ? Vj[(r0 + i) * p + vcol] : 0.0;
    }
  };
  if (tile0 * 16 >= n_pad) return;
  double af[KK], afn[KK];
  load_a(tile0 * 16, 0, af);
  for (int t = 0; t < T; ++t) {
    const int64_t r0 = (tile0 + t) * 16;
    if (r0 >= n_pad) return;
    const bool next_tile = t + 1 < T && r0 + 16 < n_pad;
    dbl4 x[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = 16 * b + i;
        x[b][g] = (beta != 0.0 && col < p) ? X[(r0 + kq + 4 * g) * p + col] : 0.0;
      }
    double bcur = beta;
    for (int j = 0; j < k; ++j) {
      if (j + 1 < k) load_a(r0, j + 1, afn);
      else if (next_tile) load_a(r0 + 16, 0, afn);
      const double *Yj = s_Y + (size_t)j * pp;
      double bf[KK][NT];
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
        for (int b = 0; b < NT; ++b) {
          const int prow = 4 * kk + kq, pcol = 16 * b + i;
          bf[kk][b] = (prow < p && pcol < p) ? Yj[prow * p + pcol] : 0.0;
        }
      }
      dbl4 acc[NT];
#pragma unroll
      for (int b = 0; b < NT; ++b) acc[b] = dbl4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < KK; ++kk)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[kk], bf[kk][b], acc[b], 0, 0, 0);
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) x[b][g] = fma(alpha, acc[b][g], bcur * x[b][g]);
      bcur = 1.0;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) af[kk] = afn[kk];
    }
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = 16 * b + i;
        if (col < p) X[(r0 + kq + 4 * g) * p + col] = x[b][g];
      }
  }
}

}  // namespace khip

using namespace khip;

namespace {
int64_t pad16(int64_t n) { return (n + 15) & ~(int64_t)15; }

// Q <- beta Q + alpha V Psi (Psi_dev: p x p column-major in device memory): the LDS-factor kernel with k = 1 when
// "panel_multi_tiles" > 0 (same bits, fewer vector-memory instructions), else one tile per wave with the factor re-read.
void launch_gemm_nn(khip_ctx *ctx, int64_t np, int p, double alpha, const double *V, const double *Psi_dev, double beta, double *Q) {
  const int64_t tiles = np / 16;
  if (tiles == 0) return;
  const int T = ctx->tune.panel_multi_tiles;
  if (T > 0) {
    MultiNNArgs a;
    for (int i = 0; i < kMultiNN; ++i) a.v[i] = i == 0 ? V : nullptr;
    const int64_t per_wg = (int64_t)kWavesPerBlock * T;
    const unsigned g = (unsigned)((tiles + per_wg - 1) / per_wg);
    const size_t lds = sizeof(double) * (size_t)p * p;
    { ProfScope prof_scope(ctx, kProfPanelNn);
    if (p <= 16) hipLaunchKernelGGL((panel_multi_nn_lds_kernel<1>), dim3(g), dim3(kBlock), lds, ctx->stream, np, p, 1, a, Psi_dev, alpha, beta, Q, T);
    else hipLaunchKernelGGL((panel_multi_nn_lds_kernel<2>), dim3(g), dim3(kBlock), lds, ctx->stream, np, p, 1, a, Psi_dev, alpha, beta, Q, T);
    }
    return;
  }
  const unsigned g = (unsigned)((tiles + kWavesPerBlock - 1) / kWavesPerBlock);
  { ProfScope prof_scope(ctx, kProfPanelNn);
  if (p <= 16) hipLaunchKernelGGL((panel_gemm_nn_kernel<1>), dim3(g), dim3(kBlock), 0, ctx->stream, np, p, alpha, V, Psi_dev, beta, Q);
  else hipLaunchKernelGGL((panel_gemm_nn_kernel<2>), dim3(g), dim3(kBlock), 0, ctx->stream, np, p, alpha, V, Psi_dev, beta, Q);
  }
}

constexpr int kPsiSlots = 64;    // staging ring for the p x p factors of Q += V Psi
struct PanelScratch {            // scratch for the V^T Q partial tiles and the Psi staging
  double *partials = nullptr;    // two ping-pong regions
  size_t partial_elems = 0;
  double *psi_dev = nullptr;     // kPsiSlots x 32 x 32
  double *psi_pinned = nullptr;
  int next_slot = 0;
};
// scratch is owned by the context (khip_ctx::panel_scratch, freed by khip_ctx_destroy)
#define g_ps (*static_cast<PanelScratch *>(ctx->panel_scratch))

// streaming accesses for panels of at least tune.nt_min_elems doubles (the BLAS-1 kernels' threshold); panel_nt = 0 / 2 force off / on
bool panel_nt(khip_ctx *ctx, int64_t np, int p) {
  if (ctx->tune.panel_nt == 0) return false;
  if (ctx->tune.panel_nt >= 2) return true;
  return np * (int64_t)p >= (int64_t)ctx->tune.nt_min_elems;
}

int ensure_panel_scratch(khip_ctx *ctx, size_t elems) {
  if (!ctx->panel_scratch) ctx->panel_scratch = new PanelScratch();
  if (!g_ps.psi_dev) {
    KHIP_CHECK_HIP(hipMalloc(&g_ps.psi_dev, sizeof(double) * 32 * 32 * kPsiSlots));
    KHIP_CHECK_HIP(hipHostMalloc(reinterpret_cast<void **>(&g_ps.psi_pinned), sizeof(double) * 32 * 32 * kPsiSlots,
                                 hipHostMallocDefault));
  }
  if (elems > g_ps.partial_elems) {
    KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    if (g_ps.partials) KHIP_CHECK_HIP(hipFree(g_ps.partials));
    KHIP_CHECK_HIP(hipMalloc(&g_ps.partials, sizeof(double) * elems));
    g_ps.partial_elems = elems;
  }
  return KHIP_OK;
}

// ---------------------------------------------------------------- TSQR: the R factor of a tall panel ----
// Communication-avoiding QR (Demmel, Grigori, Hoemmen, Langou 2012), the R factor only: every workgroup
// runs an unblocked Householder QR of one block of rows out of LDS and keeps its p x p upper triangle; the stacked triangles
// are a new tall matrix with p / kTsqrRows as many rows, reduced the same way until one block is left.  Backward stable
// whatever the conditioning of the panel (the Cholesky of the Gram matrix squares the condition number).  Fixed reduction
// order: deterministic.  Thread layout per block: 16 row groups x 16 column lanes; P = columns rounded up to 16 or 32.
constexpr int kTsqrRows = 256;           // rows of a block for P = 16 (P = 32: half of it)

template <int P>
__global__ __launch_bounds__(kBlock) void panel_block_qr_kernel(const double *A, int64_t rows, int p, double *Rout) {
  constexpr int B = P == 16 ? kTsqrRows : kTsqrRows / 2;
  constexpr int LD = P + 1;                                  // padded leading dimension: column walks without bank conflicts
  extern __shared__ double tsqr_a[];                         // [B][LD] block, then [16][P] partial sums, then [P] w
  double *part = tsqr_a + B * LD;
  double *wv = part + 16 * P;
  const int tid = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * B;
  for (int q = tid; q < B * P; q += kBlock) {
    const int i = q / P, c = q % P;
    const int64_t r = r0 + i;
    tsqr_a[i * LD + c] = (r < rows && c < p) ? A[r * p + c] : 0.0;
  }
  __syncthreads();
  const int cl = tid & 15, rg = tid >> 4;                    // column lane, row group
  for (int j = 0; j < p; ++j) {
    // sigma = sum_{i > j} a[i][j]^2 : every row group sums its rows, lane 0 of the groups publishes, thread 0 folds in order
    if (cl == 0) {
      double sg = 0.0;
      for (int i = rg; i < B; i += 16) if (i > j) sg += tsqr_a[i * LD + j] * tsqr_a[i * LD + j];
      part[rg] = sg;
    }
    __syncthreads();
    if (tid == 0) {
      double sigma = 0.0;
      for (int g = 0; g < 16; ++g) sigma += part[g];
      const double alpha = tsqr_a[j * LD + j];
      const double nrm = sqrt(alpha * alpha + sigma);
      double beta = 0.0, tau = 0.0, scale = 0.0;
      if (sigma != 0.0) {                                    // DLARFG: beta = -sign(alpha) norm, tau = (beta - alpha) / beta, v = x / (alpha - beta)
        beta = alpha >= 0.0 ? -nrm : nrm;
        tau = (beta - alpha) / beta;
        scale = 1.0 / (alpha - beta);
      } else {
        beta = alpha;                                        // H = I
      }
      wv[0] = beta; wv[1] = tau; wv[2] = scale;
    }
    __syncthreads();
    const double beta = wv[0], tau = wv[1], scale = wv[2];
    __syncthreads();
    if (tau != 0.0) {
      // w_c = a[j][c] + sum_{i > j} v_i a[i][c] for the columns c > j (v_i = a[i][j] * scale); 16 lanes x P / 16 columns each
      for (int c = cl; c < p; c += 16) {
        double acc = 0.0;
        if (c > j)
          for (int i = rg; i < B; i += 16) if (i > j) acc += tsqr_a[i * LD + j] * scale * tsqr_a[i * LD + c];
        part[rg * P + c] = acc;
      }
      __syncthreads();
      if (tid < p && tid > j) {
        double acc = tsqr_a[j * LD + tid];
        for (int g = 0; g < 16; ++g) acc += part[g * P + tid];
        wv[4 + tid] = acc;
      }
      __syncthreads();
      for (int c = cl; c < p; c += 16) {
        if (c > j) {
          const double tw = tau * wv[4 + c];
          for (int i = rg; i < B; i += 16) {
            if (i > j) tsqr_a[i * LD + c] -= tsqr_a[i * LD + j] * scale * tw;
            else if (i == j) tsqr_a[i * LD + c] -= tw;
          }
        }
      }
      __syncthreads();
    }
    if (tid == 0) tsqr_a[j * LD + j] = beta;
    __syncthreads();
  }
  // the block's R: rows blockIdx.x * p .. + p of the next level's matrix (zeros below the diagonal)
  for (int q = tid; q < p * p; q += kBlock) {
    const int i = q / p, c = q % p;
    Rout[((int64_t)blockIdx.x * p + i) * p + c] = c >= i ? tsqr_a[i * LD + c] : 0.0;
  }
}

// R (p x p, row-major, as the last block leaves it: arbitrary diagonal signs) of the n x p row-major panel Q -> R_host
int panel_tsqr_r_impl(khip_ctx *ctx, int64_t n, int p, const double *Q, double *R_host_rowmajor) {
  if (p < 1 || p > 32) { set_error("panel_tsqr_r: 1 <= p <= 32 required"); return KHIP_ERR_INVALID; }
  const int P = p <= 16 ? 16 : 32;
  const int B = P == 16 ? kTsqrRows : kTsqrRows / 2;
  const size_t lds = sizeof(double) * ((size_t)B * (P + 1) + 16 * P + 4 + P);
  int64_t rows = n;
  int64_t nb = (rows + B - 1) / B;
  if (nb < 1) nb = 1;
  // two ping-pong regions for the stacked triangles of the levels
  const size_t lvl1 = (size_t)nb * p * p;
  KHIP_TRY(ensure_panel_scratch(ctx, 2 * lvl1 + 64));
  double *ping = g_ps.partials, *pong = g_ps.partials + lvl1 + 32;
  const double *src = Q;
  for (;;) {
    nb = (rows + B - 1) / B;
    if (nb < 1) nb = 1;
    if (P == 16) hipLaunchKernelGGL((panel_block_qr_kernel<16>), dim3((unsigned)nb), dim3(kBlock), lds, ctx->stream, src, rows, p, ping);
    else hipLaunchKernelGGL((panel_block_qr_kernel<32>), dim3((unsigned)nb), dim3(kBlock), lds, ctx->stream, src, rows, p, ping);
    KHIP_CHECK_HIP(hipGetLastError());
    if (nb == 1) break;
    rows = nb * p;
    src = ping;
    std::swap(ping, pong);
  }
  KHIP_CHECK_HIP(hipMemcpyAsync(R_host_rowmajor, ping, sizeof(double) * (size_t)p * p, hipMemcpyDeviceToHost, ctx->stream));
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return KHIP_OK;
}
}  // namespace

namespace khip {
// Columns `mask` of the n x p row-major panel <- scale * u(row, column), u a fixed hash into [-1, 1): the stand-in directions
// of khip_panel_qr_tau for columns that lie in the span of the columns before them (what LAPACK's reflectors complete the
// basis with is just as arbitrary).  Rows past n (the panel's padding) stay as they are.
__global__ __launch_bounds__(kBlock) void panel_fill_columns_kernel(int64_t n, int p, double *Q, unsigned mask, double scale,
                                                                    unsigned long long seed) {
  const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (idx >= n * p) return;
  const int c = (int)(idx % p);
  if (!((mask >> c) & 1u)) return;
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(idx + 1);       // splitmix64
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  Q[idx] = scale * ((double)(long long)(z >> 11) * (1.0 / 4503599627370496.0) - 1.0);         // 53 bits -> [-1, 1)
}
int panel_fill_columns(khip_ctx *ctx, int64_t n, int p, double *Q, unsigned mask, double scale, unsigned long long seed) {
  if (n <= 0 || mask == 0) return KHIP_OK;
  const int64_t blocks = (n * p + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(panel_fill_columns_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, ctx->stream, n, p, Q, mask, scale, seed);
  KHIP_CHECK_HIP(hipGetLastError());
  return KHIP_OK;
}
int panel_tsqr_r(khip_ctx *ctx, int64_t n, int p, const double *Q, double *R_host_rowmajor) {
  return panel_tsqr_r_impl(ctx, n, p, Q, R_host_rowmajor);
}
void panel_scratch_destroy(khip_ctx *ctx) {
  if (!ctx->panel_scratch) return;
  PanelScratch *ps = static_cast<PanelScratch *>(ctx->panel_scratch);
  if (ps->partials) (void)hipFree(ps->partials);
  if (ps->psi_dev) (void)hipFree(ps->psi_dev);
  if (ps->psi_pinned) (void)hipHostFree(ps->psi_pinned);
  delete ps;
  ctx->panel_scratch = nullptr;
}
}  // namespace khip

extern "C" {

int khip_panel_rows(int64_t n, int64_t *n_pad) {
  KHIP_REQUIRE(n_pad, "panel_rows: null argument");
  *n_pad = pad16(n);
  return KHIP_OK;
}

int khip_panel_from_colmajor(khip_ctx *ctx, int64_t n, int p, const double *X_colmajor, double *P) {
  KHIP_REQUIRE(ctx && X_colmajor && P && p >= 1 && p <= 32, "panel_from_colmajor: bad argument (1 <= p <= 32)");
  const int64_t np = pad16(n);
  if (np > n) KHIP_CHECK_HIP(hipMemsetAsync(P + n * p, 0, sizeof(double) * (size_t)(np - n) * p, ctx->stream));
  if (n == 0) return KHIP_OK;
  const unsigned g = (unsigned)((n + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(panel_from_colmajor_kernel, dim3(g), dim3(kBlock), sizeof(double) * kBlock * (p + 1), ctx->stream, n, p,
                     X_colmajor, P);
  KHIP_CHECK_HIP(hipGetLastError());
  return KHIP_OK;
}

int khip_panel_to_colmajor(khip_ctx *ctx, int64_t n, int p, const double *P, double *X_colmajor) {
  KHIP_REQUIRE(ctx && X_colmajor && P && p >= 1 && p <= 32, "panel_to_colmajor: bad argument (1 <= p <= 32)");
  if (n == 0) return KHIP_OK;
  const unsigned g = (unsigned)((n + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(panel_to_colmajor_kernel, dim3(g), dim3(kBlock), sizeof(double) * kBlock * (p + 1), ctx->stream, n, p, P,
                     X_colmajor);
  KHIP_CHECK_HIP(hipGetLastError());
  return KHIP_OK;
}

// Launch geometry of the V^T Q kernels and the reduction tree that turns their per-wave tiles into Psi (p x p,
// column-major) at psi_out on the device.  `first` launches the kernel that fills the ping region.
struct TnPlan {
  int NT;
  int64_t nwaves;
  unsigned blocks;
  size_t tile_elems;
};
static TnPlan tn_plan(int64_t np, int p) {
  TnPlan t;
  t.NT = p <= 16 ? 1 : 2;
  const int64_t nwaves_used = (np + kRowsPerWaveTN - 1) / kRowsPerWaveTN;
  const int64_t nblocks = (nwaves_used + kWavesPerBlock - 1) / kWavesPerBlock;
  t.blocks = (unsigned)(nblocks > 0 ? nblocks : 1);
  t.nwaves = (int64_t)t.blocks * kWavesPerBlock;
  t.tile_elems = (size_t)t.NT * t.NT * 256;
  return t;
}
static void tn_reduce(khip_ctx *ctx, const TnPlan &t, int p, double *psi_out) {
  double *ping = g_ps.partials, *pong = g_ps.partials + (size_t)t.blocks * t.tile_elems;
  int64_t count = t.blocks;
  while (true) {
    const int64_t groups = (count + kTnFan - 1) / kTnFan;
    double *out = groups == 1 ? psi_out : nullptr;
    if (t.NT == 1) hipLaunchKernelGGL((panel_tn_reduce_kernel<1>), dim3((unsigned)groups, 1), dim3(kBlock), 0, ctx->stream, count, p, ping, pong, out);
    else hipLaunchKernelGGL((panel_tn_reduce_kernel<2>), dim3((unsigned)groups, 4), dim3(kBlock), 0, ctx->stream, count, p, ping, pong, out);
    if (groups == 1) break;
    count = groups;
    double *tmp = ping; ping = pong; pong = tmp;
  }
}
// Psi = V^T Q into psi_out (device), no host synchronisation
static int tn_enqueue(khip_ctx *ctx, int64_t np, int p, const double *V, const double *Q, double *psi_out) {
  const TnPlan t = tn_plan(np, p);
  KHIP_TRY(ensure_panel_scratch(ctx, 2 * (size_t)t.blocks * t.tile_elems));
  { ProfScope prof_scope(ctx, kProfPanelTn);
  if (t.NT == 1) hipLaunchKernelGGL((panel_gemm_tn_kernel<1>), dim3(t.blocks), dim3(kBlock), 0, ctx->stream, np, p, V, Q, g_ps.partials);
  else hipLaunchKernelGGL((panel_gemm_tn_kernel<2>), dim3(t.blocks), dim3(kBlock), 0, ctx->stream, np, p, V, Q, g_ps.partials);
  }
  tn_reduce(ctx, t, p, psi_out);
  KHIP_CHECK_HIP(hipGetLastError());
  return KHIP_OK;
}

int khip_panel_gemm_tn(khip_ctx *ctx, int64_t n, int p, const double *V, const double *Q, double *Psi_host) {
  KHIP_REQUIRE(ctx && V && Q && Psi_host && p >= 1 && p <= 32, "panel_gemm_tn: bad argument (1 <= p <= 32)");
  KHIP_TRY(ensure_panel_scratch(ctx, 1));
  KHIP_TRY(tn_enqueue(ctx, pad16(n), p, V, Q, g_ps.psi_dev));
  KHIP_CHECK_HIP(hipMemcpyAsync(g_ps.psi_pinned, g_ps.psi_dev, sizeof(double) * (size_t)p * p, hipMemcpyDeviceToHost, ctx->stream));
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  memcpy(Psi_host, g_ps.psi_pinned, sizeof(double) * (size_t)p * p);
  // row-partitioned panels: the block is the sum of the ranks' partial blocks (2 KB at p = 16), added in rank order
  return comm_allreduce_sum_host(ctx, Psi_host, p * p);
}

}  // extern "C"
// Q <- Q Ri in place and G = Q_new^T Q_new in the same pass (khip_panel_qr: the scaling of round 1 with the Gram matrix
// of round 2); bit-identical to khip_panel_gemm_nn followed by khip_panel_gemm_tn(Q, Q).  G is rank-summed like
// every V^T Q.
int khip::panel_scale_gram(khip_ctx *ctx, int64_t n, int p, double *Q, const double *Ri_host, double *G_host) {
  const int64_t np = pad16(n);
  const TnPlan t = tn_plan(np, p);
  KHIP_TRY(ensure_panel_scratch(ctx, 2 * (size_t)t.blocks * t.tile_elems));
  if (g_ps.next_slot == kPsiSlots || g_ps.next_slot == 0) {
    KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    g_ps.next_slot = 1;
  }
  const int slot = g_ps.next_slot++;
  double *psi_h = g_ps.psi_pinned + (size_t)slot * 1024, *psi_d = g_ps.psi_dev + (size_t)slot * 1024;
  memcpy(psi_h, Ri_host, sizeof(double) * (size_t)p * p);
  KHIP_CHECK_HIP(hipMemcpyAsync(psi_d, psi_h, sizeof(double) * (size_t)p * p, hipMemcpyHostToDevice, ctx->stream));
  { ProfScope prof_scope(ctx, kProfPanelQr);
  if (t.NT == 1) hipLaunchKernelGGL((panel_nn_tn_kernel<1, 2, true>), dim3(t.blocks), dim3(kBlock), 0, ctx->stream, np, p, 1.0, Q, psi_d, 0.0, nullptr, Q, g_ps.partials, (ctx->tune.panel_a_lds ? 1 : 0) | (panel_nt(ctx, np, p) ? 2 : 0));
  else hipLaunchKernelGGL((panel_nn_tn_kernel<2, 2, true>), dim3(t.blocks), dim3(kBlock), 0, ctx->stream, np, p, 1.0, Q, psi_d, 0.0, nullptr, Q, g_ps.partials, (ctx->tune.panel_a_lds ? 1 : 0) | (panel_nt(ctx, np, p) ? 2 : 0));
  }
  tn_reduce(ctx, t, p, g_ps.psi_dev);
  KHIP_CHECK_HIP(hipGetLastError());
  KHIP_CHECK_HIP(hipMemcpyAsync(g_ps.psi_pinned, g_ps.psi_dev, sizeof(double) * (size_t)p * p, hipMemcpyDeviceToHost, ctx->stream));
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  memcpy(G_host, g_ps.psi_pinned, sizeof(double) * (size_t)p * p);
  return comm_allreduce_sum_host(ctx, G_host, p * p);
}
extern "C" {

// the solution update of block_gmres! (src/block_gmres.jl:324-326) as one call; see include/krylov_hip.h
int khip_panel_multi_nn(khip_ctx *ctx, int64_t n, int p, int k, const double *const *V_host, const double *Y_host, double beta,
                        double *X) {
  KHIP_REQUIRE(ctx && X && p >= 1 && p <= 32 && k >= 0 && (k == 0 || (V_host && Y_host)), "panel_multi_nn: bad argument (1 <= p <= 32)");
  return khip::panel_multi_nn(ctx, n, p, k, V_host, Y_host, beta, X);
}

}  // extern "C"
// X <- beta X + sum_i V_i Y_i (Y_host: k blocks of p x p, column-major), products applied in the order i = 0..k-1:
// the k calls khip_panel_gemm_nn(1, V_i, Y_i, beta_i, X) with beta_0 = beta, beta_i = 1 of src/block_gmres.jl:324-326.
int khip::panel_multi_nn(khip_ctx *ctx, int64_t n, int p, int k, const double *const *V_host, const double *Y_host,
                         double beta, double *X) {
  const size_t pp = (size_t)p * p;
  if (ctx->tune.panel_fuse == 0 || k < 1 || k > kMultiNN || k + 1 >= kPsiSlots) {
    for (int i = 0; i < k; ++i) KHIP_TRY(khip_panel_gemm_nn(ctx, n, p, 1.0, V_host[i], Y_host + (size_t)i * pp, i == 0 ? beta : 1.0, X));
    return KHIP_OK;
  }
  const int64_t np = pad16(n);
  KHIP_TRY(ensure_panel_scratch(ctx, 1));
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));       // the Psi staging ring is ours from slot 1 on
  MultiNNArgs a;
  for (int i = 0; i < kMultiNN; ++i) a.v[i] = i < k ? V_host[i] : nullptr;
  for (int i = 0; i < k; ++i) memcpy(g_ps.psi_pinned + (size_t)(i + 1) * 1024, Y_host + (size_t)i * pp, sizeof(double) * pp);
  KHIP_CHECK_HIP(hipMemcpyAsync(g_ps.psi_dev + 1024, g_ps.psi_pinned + 1024, sizeof(double) * 1024 * (size_t)k, hipMemcpyHostToDevice, ctx->stream));
  g_ps.next_slot = k + 1;
  const int64_t tiles = np / 16;
  if (tiles == 0) return KHIP_OK;
  const size_t lds = sizeof(double) * (size_t)k * pp;
  const int T = ctx->tune.panel_multi_tiles;
  if (T > 0 && lds <= 48 * 1024) {            // factors in LDS, T tiles per wave
    const int64_t per_wg = (int64_t)kWavesPerBlock * T;
    const unsigned g = (unsigned)((tiles + per_wg - 1) / per_wg);
    { ProfScope prof_scope(ctx, kProfPanelMultiNn);
    if (p <= 16) hipLaunchKernelGGL((panel_multi_nn_lds_kernel<1>), dim3(g), dim3(kBlock), lds, ctx->stream, np, p, k, a, g_ps.psi_dev + 1024, 1.0, beta, X, T);
    else hipLaunchKernelGGL((panel_multi_nn_lds_kernel<2>), dim3(g), dim3(kBlock), lds, ctx->stream, np, p, k, a, g_ps.psi_dev + 1024, 1.0, beta, X, T);
    }
    KHIP_CHECK_HIP(hipGetLastError());
    return KHIP_OK;
  }
  const unsigned g = (unsigned)((tiles + kWavesPerBlock - 1) / kWavesPerBlock);
  { ProfScope prof_scope(ctx, kProfPanelMultiNn);
  if (p <= 16) hipLaunchKernelGGL((panel_multi_nn_kernel<1>), dim3(g), dim3(kBlock), 0, ctx->stream, np, p, k, a, g_ps.psi_dev + 1024, beta, X);
  else hipLaunchKernelGGL((panel_multi_nn_kernel<2>), dim3(g), dim3(kBlock), 0, ctx->stream, np, p, k, a, g_ps.psi_dev + 1024, beta, X);
  }
  KHIP_CHECK_HIP(hipGetLastError());
  return KHIP_OK;
}
extern "C" {

// Block Gram-Schmidt sweep of Q against the k panels V[0..k) in the reference's order (src/block_gmres.jl:244-247):
//   for i: Psi_i = V_i^T Q ; Q <- Q - V_i Psi_i.
// Single GPU: every Psi_i stays on the device (Psi_{i+1} comes out of the fused kernel that applies Psi_i), the host
// reads all k blocks after ONE synchronisation.  Row-partitioned panels need the rank sum of every Psi_i before it is
// applied, which goes through the host: the two-kernel sequence per step.  Either way the same bits.
int khip_panel_mgs(khip_ctx *ctx, int64_t n, int p, int k, const double *const *V_host, double *Q, double *Psi_host,
                   int accumulate) {
  return khip::panel_mgs_gram(ctx, n, p, k, V_host, Q, Psi_host, accumulate, nullptr, nullptr);
}
}  // extern "C"

// The sweep, and on request the Gram matrix G = Q^T Q of the swept panel with it: the last update Q -= V_k Psi_k and G come out
// of one pass (the SELF form of panel_nn_tn_kernel, as in panel_scale_gram), so the CholeskyQR that follows in block_gmres!
// (src/block_gmres.jl:259) starts without its own pass over the panel.  Same bits as khip_panel_gemm_tn(Q, Q) after the sweep.
// *have_gram says whether G_host was filled (single GPU, fused sweeps only).
int khip::panel_mgs_gram(khip_ctx *ctx, int64_t n, int p, int k, const double *const *V_host, double *Q, double *Psi_host,
                         int accumulate, double *G_host, bool *have_gram) {
  KHIP_REQUIRE(ctx && Q && p >= 1 && p <= 32 && k >= 0 && (k == 0 || (V_host && Psi_host)), "panel_mgs: bad argument (1 <= p <= 32)");
  if (have_gram) *have_gram = false;
  const size_t pp = (size_t)p * p;
  const bool fuse = ctx->tune.panel_fuse != 0 && comm_nranks(ctx) == 1 && k >= 1 && k + 2 < kPsiSlots;   // Psi_i lives in slot i + 1 of the ring, G in slot k + 1
  const bool want_gram = fuse && G_host != nullptr && have_gram != nullptr;
  if (!fuse) {
    std::vector<double> psi(pp);
    for (int i = 0; i < k; ++i) {
      KHIP_TRY(khip_panel_gemm_tn(ctx, n, p, V_host[i], Q, psi.data()));
      KHIP_TRY(khip_panel_gemm_nn(ctx, n, p, -1.0, V_host[i], psi.data(), 1.0, Q));
      for (size_t l = 0; l < pp; ++l) Psi_host[(size_t)i * pp + l] = accumulate ? Psi_host[(size_t)i * pp + l] + psi[l] : psi[l];
    }
    return KHIP_OK;
  }
  const int64_t np = pad16(n);
  const TnPlan t = tn_plan(np, p);
  KHIP_TRY(ensure_panel_scratch(ctx, 2 * (size_t)t.blocks * t.tile_elems));
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));       // the Psi staging ring is ours from slot 1 on
  g_ps.next_slot = 1;
  const int64_t tiles = np / 16;
  KHIP_TRY(tn_enqueue(ctx, np, p, V_host[0], Q, g_ps.psi_dev + 1024));
  for (int i = 0; i < k; ++i) {
    double *psi_i = g_ps.psi_dev + (size_t)(i + 1) * 1024;
    if (i + 1 < k) {
      double *psi_n = g_ps.psi_dev + (size_t)(i + 2) * 1024;
      { ProfScope prof_scope(ctx, kProfPanelNnTn);
      if (t.NT == 1) hipLaunchKernelGGL((panel_nn_tn_kernel<1, 2, false>), dim3(t.blocks), dim3(kBlock), 0, ctx->stream, np, p, -1.0, V_host[i], psi_i, 1.0, V_host[i + 1], Q, g_ps.partials, (ctx->tune.panel_a_lds ? 1 : 0) | (panel_nt(ctx, np, p) ? 2 : 0));
      else hipLaunchKernelGGL((panel_nn_tn_kernel<2, 2, false>), dim3(t.blocks), dim3(kBlock), 0, ctx->stream, np, p, -1.0, V_host[i], psi_i, 1.0, V_host[i + 1], Q, g_ps.partials, (ctx->tune.panel_a_lds ? 1 : 0) | (panel_nt(ctx, np, p) ? 2 : 0));
      }
      tn_reduce(ctx, t, p, psi_n);
    } else if (tiles > 0 && want_gram) {
      double *psi_g = g_ps.psi_dev + (size_t)(k + 1) * 1024;
      const int flags = (ctx->tune.panel_a_lds ? 1 : 0) | (panel_nt(ctx, np, p) ? 2 : 0);
      { ProfScope prof_scope(ctx, kProfPanelNnTn);
      if (t.NT == 1) hipLaunchKernelGGL((panel_nn_tn_kernel<1, 2, true>), dim3(t.blocks), dim3(kBlock), 0, ctx->stream, np, p, -1.0, V_host[i], psi_i, 1.0, nullptr, Q, g_ps.partials, flags);
      else hipLaunchKernelGGL((panel_nn_tn_kernel<2, 2, true>), dim3(t.blocks), dim3(kBlock), 0, ctx->stream, np, p, -1.0, V_host[i], psi_i, 1.0, nullptr, Q, g_ps.partials, flags);
      }
      tn_reduce(ctx, t, p, psi_g);
    } else if (tiles > 0) {
      launch_gemm_nn(ctx, np, p, -1.0, V_host[i], psi_i, 1.0, Q);
    }
  }
  KHIP_CHECK_HIP(hipGetLastError());
  const int nslots = k + ((want_gram && tiles > 0) ? 1 : 0);
  KHIP_CHECK_HIP(hipMemcpyAsync(g_ps.psi_pinned + 1024, g_ps.psi_dev + 1024, sizeof(double) * 1024 * (size_t)nslots, hipMemcpyDeviceToHost, ctx->stream));
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < k; ++i) {
    const double *src = g_ps.psi_pinned + (size_t)(i + 1) * 1024;
    double *dst = Psi_host + (size_t)i * pp;
    for (size_t l = 0; l < pp; ++l) dst[l] = accumulate ? dst[l] + src[l] : src[l];
  }
  if (want_gram && tiles > 0) {
    memcpy(G_host, g_ps.psi_pinned + (size_t)(k + 1) * 1024, sizeof(double) * pp);
    *have_gram = true;
  }
  g_ps.next_slot = nslots + 1;
  return KHIP_OK;
}
extern "C" {

int khip_panel_gemm_nn(khip_ctx *ctx, int64_t n, int p, double alpha, const double *V, const double *Psi_host, double beta,
                       double *Q) {
  KHIP_REQUIRE(ctx && V && Q && Psi_host && p >= 1 && p <= 32, "panel_gemm_nn: bad argument (1 <= p <= 32)");
  const int64_t np = pad16(n);
  KHIP_TRY(ensure_panel_scratch(ctx, 1));
  // Psi is uploaded through a ring of staging slots (the ring is drained before it wraps)
  if (g_ps.next_slot == kPsiSlots) {
    KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    g_ps.next_slot = 1;                      // slot 0 is the result slot of khip_panel_gemm_tn
  }
  if (g_ps.next_slot == 0) g_ps.next_slot = 1;
  const int slot = g_ps.next_slot++;
  double *psi_h = g_ps.psi_pinned + (size_t)slot * 1024, *psi_d = g_ps.psi_dev + (size_t)slot * 1024;
  memcpy(psi_h, Psi_host, sizeof(double) * (size_t)p * p);
  KHIP_CHECK_HIP(hipMemcpyAsync(psi_d, psi_h, sizeof(double) * (size_t)p * p, hipMemcpyHostToDevice, ctx->stream));
  launch_gemm_nn(ctx, np, p, alpha, V, psi_d, beta, Q);
  KHIP_CHECK_HIP(hipGetLastError());
  return KHIP_OK;
}

int khip_panel_norm(khip_ctx *ctx, int64_t n, int p, const double *Q, double *result_host) {
  KHIP_REQUIRE(ctx && Q && result_host && p >= 1, "panel_norm: bad argument");
  return khip_nrm2(ctx, pad16(n) * p, Q, result_host);     // padding rows are zero
}

}  // extern "C"
