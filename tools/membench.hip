// membench.hip -- which access pattern reaches the MI355X HBM ceiling? (tuning aid, not product code)
// hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double dbl2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

enum Op { COPY = 0, READ = 1, FILL = 2, AXPY = 3 };

template <int OP, bool NT> __device__ __forceinline__ void body(const dbl2* x, dbl2* y, long i, dbl2& acc) {
  if (OP == COPY) { dbl2 v = NT ? __builtin_nontemporal_load(x + i) : x[i]; if (NT) __builtin_nontemporal_store(v, y + i); else y[i] = v; }
  if (OP == READ) { dbl2 v = NT ? __builtin_nontemporal_load(x + i) : x[i]; acc += v; }
  if (OP == FILL) { dbl2 v = {1.0, 2.0}; if (NT) __builtin_nontemporal_store(v, y + i); else y[i] = v; }
  if (OP == AXPY) { dbl2 a = x[i], b = y[i]; b.x = fma(0.5, a.x, b.x); b.y = fma(0.5, a.y, b.y); y[i] = b; }
}

// grid-stride, U independent accesses in flight
template <int OP, bool NT, int U> __global__ __launch_bounds__(256) void k_gs(const dbl2* x, dbl2* y, long nv, double* sink) {
  long stride = (long)gridDim.x * 256, i = (long)blockIdx.x * 256 + threadIdx.x;
  dbl2 acc = {0, 0};
  for (; i + (U - 1) * stride < nv; i += U * stride) {
    if (OP == AXPY || OP == COPY) {
      dbl2 a[U], b[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { a[u] = NT ? __builtin_nontemporal_load(x + i + u * stride) : x[i + u * stride]; if (OP == AXPY) b[u] = y[i + u * stride]; }
#pragma unroll
      for (int u = 0; u < U; ++u) { dbl2 o = a[u]; if (OP == AXPY) { o.x = fma(0.5, a[u].x, b[u].x); o.y = fma(0.5, a[u].y, b[u].y); }
        if (NT) __builtin_nontemporal_store(o, y + i + u * stride); else y[i + u * stride] = o; }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) body<OP, NT>(x, y, i + u * stride, acc);
    }
  }
  for (; i < nv; i += stride) body<OP, NT>(x, y, i, acc);
  if (OP == READ && acc.x + acc.y == 12345.678) *sink = acc.x;
}
// contiguous chunk per block
template <int OP, bool NT, int U> __global__ __launch_bounds__(256) void k_chunk(const dbl2* x, dbl2* y, long nv, double* sink) {
  long per = (nv + gridDim.x - 1) / gridDim.x;
  long lo = per * blockIdx.x, hi = lo + per < nv ? lo + per : nv;
  dbl2 acc = {0, 0};
  long i = lo + threadIdx.x;
  for (; i + (U - 1) * 256 < hi; i += U * 256) {
    if (OP == AXPY || OP == COPY) {
      dbl2 a[U], b[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { a[u] = NT ? __builtin_nontemporal_load(x + i + u * 256) : x[i + u * 256]; if (OP == AXPY) b[u] = y[i + u * 256]; }
#pragma unroll
      for (int u = 0; u < U; ++u) { dbl2 o = a[u]; if (OP == AXPY) { o.x = fma(0.5, a[u].x, b[u].x); o.y = fma(0.5, a[u].y, b[u].y); }
        if (NT) __builtin_nontemporal_store(o, y + i + u * 256); else y[i + u * 256] = o; }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) body<OP, NT>(x, y, i + u * 256, acc);
    }
  }
  for (; i < hi; i += 256) body<OP, NT>(x, y, i, acc);
  if (OP == READ && acc.x + acc.y == 12345.678) *sink = acc.x;
}

template <typename F> float timeit(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int r = 0; r < reps; ++r) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main(int argc, char** argv) {
  long n = argc > 1 ? atol(argv[1]) : 134217728L;
  long nv = n / 2;
  dbl2 *x, *y; double* sink;
  CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, n * 8)); CK(hipMalloc(&sink, 8));
  CK(hipMemset(x, 0, n * 8)); CK(hipMemset(y, 0, n * 8));
  const char* opn[] = {"copy", "read", "fill", "axpy"};
  const double bpe[] = {16, 8, 8, 24};
#define RUN(KER, OP, NT, U, G) do { float ms = timeit([&] { hipLaunchKernelGGL((KER<OP, NT, U>), dim3(G), dim3(256), 0, 0, x, y, nv, sink); }, 10); \
    printf("%-6s %-5s nt=%d U=%d grid=%-7ld %.3f ms %.0f GB/s\n", #KER, opn[OP], (int)NT, U, (long)(G), ms, bpe[OP] * n / ms / 1e6); fflush(stdout); } while (0)
  long flat = (nv + 255) / 256;
  for (long G : {256L, 512L, 1024L, 2048L, 4096L, 16384L, flat / 4, flat}) {
    RUN(k_gs, COPY, false, 1, G); RUN(k_gs, COPY, false, 4, G);
    RUN(k_gs, READ, false, 4, G); RUN(k_gs, AXPY, false, 4, G);
  }
  for (long G : {512L, 1024L, 2048L, 4096L, 16384L, 65536L}) {
    RUN(k_chunk, COPY, false, 4, G); RUN(k_chunk, READ, false, 4, G); RUN(k_chunk, AXPY, false, 4, G); RUN(k_chunk, AXPY, false, 8, G);
  }
  for (long G : {2048L, flat / 4}) {
    RUN(k_gs, COPY, true, 4, G); RUN(k_gs, READ, true, 4, G); RUN(k_gs, FILL, false, 4, G); RUN(k_gs, FILL, true, 4, G);
    RUN(k_gs, READ, false, 8, G); RUN(k_gs, READ, false, 2, G);
  }
  return 0;
}
