// gatherbench.hip -- what does the x gather of a 7-point row walk cost on MI355X, and does staging x through LDS with
// 16-byte loads help?  (tuning aid, not product code)   hipcc --offload-arch=gfx950 -O3 tools/gatherbench.hip -o tools/gatherbench
//   mode 0: one lane per row, 7 global_load_dwordx2 at x[i + d_k] (what the staged SpMV kernels do)
//   mode 1: per 256-row tile the 7 ranges x[r0 + d_k .. + 256) go to LDS with 16-byte loads, rows read LDS
//   mode 2: as 1 with the ranges of d = -1, 0, +1 merged (5 ranges)
//   STREAM = bytes per row of an additional coalesced 16-byte stream (the matrix), read and folded into y
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double dbl2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// W > 0: plane sweep -- XCD p = b % 8 walks a column of W consecutive tiles through all planes (S tiles per plane)
__device__ __forceinline__ long tile_of(long b, long G, int S, int W) {
  if (W <= 0 || S % (8 * W) != 0) return b;
  const long K = G / S;
  if (b >= K * S) return b;
  const long p = b & 7, l = b >> 3;
  const long per = K * W;
  const long tt = l / per, rem = l - tt * per;
  const long k = rem / W, w = rem - k * W;
  return k * S + (tt * 8 + p) * W + w;
}

template <int MODE, int STREAM>
__global__ __launch_bounds__(256) void k(const double* __restrict__ x, const dbl2* __restrict__ st, double* __restrict__ y, long n, int n1, int W) {
  const long tile = tile_of(blockIdx.x, gridDim.x, n1 * n1 / 256, W);
  const long r0 = tile * 256;
  const int tid = threadIdx.x;
  const long i = r0 + tid;
  const long d[7] = {-(long)n1 * n1, -(long)n1, -1, 0, 1, (long)n1, (long)n1 * n1};
  double acc = 0.0;
  dbl2 sv[STREAM / 16 > 0 ? STREAM / 16 : 1];
  if (STREAM > 0) {
#pragma unroll
    for (int q = 0; q < STREAM / 16; ++q) sv[q] = st[(tile * (STREAM / 16) + q) * 256 + tid];
  }
  if (MODE == 0) {
    double xv[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) { long j = i + d[k]; xv[k] = (j >= 0 && j < n) ? x[j] : 0.0; }
#pragma unroll
    for (int k = 0; k < 7; ++k) acc += xv[k];
  } else {
    __shared__ double s_x[7][260];
    constexpr int NR = MODE == 1 ? 7 : 5;
    const long dr[7] = {d[0], d[1], MODE == 1 ? d[2] : -2, MODE == 1 ? d[3] : d[5], MODE == 1 ? d[4] : d[6], d[5], d[6]};
    // range r covers x[start_r .. start_r + 260) with start_r = (r0 + dr[r]) & ~1 ; 130 lanes x 16 B
    for (int t = tid; t < NR * 130; t += 256) {
      const int r = t / 130, l = t - r * 130;
      const long start = (r0 + dr[r]) & ~1L;
      const long j = start + 2 * l;
      dbl2 v = {0.0, 0.0};
      if (j >= 0 && j + 1 < n) v = *reinterpret_cast<const dbl2*>(x + j);
      *reinterpret_cast<dbl2*>(&s_x[r][2 * l]) = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      int r; long dd;
      if (MODE == 1) { r = k; dd = d[k]; }
      else { r = (k <= 1) ? k : (k <= 4 ? 2 : k - 2); dd = (k <= 1 || k >= 5) ? d[k] : -2; }
      const long start = (r0 + dd) & ~1L;
      const long j = i + d[k];
      acc += (j >= 0 && j < n) ? s_x[r][j - start] : 0.0;
    }
  }
  if (STREAM > 0) {
#pragma unroll
    for (int q = 0; q < STREAM / 16; ++q) acc += sv[q].x + sv[q].y;
  }
  if (i < n) y[i] = acc;
}

template <typename F> float timeit(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int r = 0; r < reps; ++r) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main(int argc, char** argv) {
  int n1 = argc > 1 ? atoi(argv[1]) : 512;
  long n = (long)n1 * n1 * n1;
  double *x, *y; dbl2* st;
  CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, n * 8)); CK(hipMalloc(&st, n * 96));
  CK(hipMemset(x, 0, n * 8)); CK(hipMemset(y, 0, n * 8)); CK(hipMemset(st, 0, n * 96));
  const unsigned G = (unsigned)((n + 255) / 256);
#define RUN(MODE, STREAM, W) do { float ms = timeit([&] { hipLaunchKernelGGL((k<MODE, STREAM>), dim3(G), dim3(256), 0, 0, x, st, y, n, n1, W); }, 10); \
    printf("mode %d stream %2d B/row  W %3d  %.3f ms  %.2f ps/row  HBM-algorithmic %.0f GB/s\n", MODE, STREAM, W, ms, ms * 1e9 / n, (16.0 + STREAM) * n / ms / 1e6); fflush(stdout); } while (0)
  for (int rep = 0; rep < 2; ++rep) {
    for (int W : {0, 8, 16, 32, 64, 128}) { RUN(0, 0, W); }
    for (int W : {0, 8, 16, 32, 64, 128}) { RUN(0, 64, W); }
    for (int W : {0, 16, 128}) { RUN(0, 96, W); }
    RUN(2, 64, 0); RUN(2, 64, 128);
  }
  return 0;
}
