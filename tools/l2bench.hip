// l2bench.hip -- what do the vector L1 / L2 deliver per CU for wave-contiguous reads of 8 vs 16 bytes per lane, from an
// L1-, L2-, MALL- or HBM-resident buffer?  (tuning aid)   hipcc --offload-arch=gfx950 -O3 tools/l2bench.hip -o tools/l2bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double dbl2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// every wave reads `iters` x U chunks; chunk c of wave w at element offset ((w * 977 + c * 131) * 64 * W8) mod n, lane-contiguous
template <int W8, int U>   // W8 = doubles per lane (1 or 2)
__global__ __launch_bounds__(256) void k(const double* __restrict__ x, long nmask, int iters, double* sink) {
  const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  double acc = 0.0;
  long c = 0;
  for (int it = 0; it < iters; ++it) {
    double v[U * W8];
#pragma unroll
    for (int u = 0; u < U; ++u, ++c) {
      const long off = (((w * 977 + c * 131) * 64 + lane) * W8) & nmask;
      if (W8 == 2) { dbl2 t = *reinterpret_cast<const dbl2*>(x + off); v[2 * u] = t.x; v[2 * u + 1] = t.y; }
      else v[u] = x[off];
    }
#pragma unroll
    for (int u = 0; u < U * W8; ++u) acc += v[u];
  }
  if (acc == 12345.678) *sink = acc;
}

int main() {
  const long nmax = 1L << 27;   // 1 GiB of doubles
  double *x, *sink;
  CK(hipMalloc(&x, nmax * 8)); CK(hipMalloc(&sink, 8)); CK(hipMemset(x, 0, nmax * 8));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int G = 256 * 8;   // 8 workgroups per CU
#define RUN(W8, U, BYTES, ITERS) do { const long n = (BYTES) / 8; \
    hipLaunchKernelGGL((k<W8, U>), dim3(G), dim3(256), 0, 0, x, n - 1, ITERS, sink); CK(hipDeviceSynchronize()); \
    CK(hipEventRecord(a)); hipLaunchKernelGGL((k<W8, U>), dim3(G), dim3(256), 0, 0, x, n - 1, ITERS, sink); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); \
    float ms; CK(hipEventElapsedTime(&ms, a, b)); const double bytes = (double)G * 256 * (double)(ITERS) * U * W8 * 8; \
    printf("%2d B/lane  U=%d  footprint %8.2f MiB : %.3f ms  %7.2f TB/s  %5.1f B/clk/CU at 2.4 GHz\n", W8 * 8, U, (BYTES) / 1048576.0, ms, bytes / ms / 1e9, \
           bytes / ms / 1e9 * 1e12 / 256 / 2.4e9 / 1e3 * 1e3 / 1e9 * 1e6); fflush(stdout); } while (0)
  for (long bytes : {16L << 10, 2L << 20, 16L << 20, 128L << 20, 1L << 30}) {
    RUN(1, 8, bytes, 400); RUN(2, 8, bytes, 400); RUN(2, 4, bytes, 800); RUN(1, 16, bytes, 200);
  }
  return 0;
}
