// ldsbench.hip -- what does one wave LDS instruction cost the CU, by width and access pattern?  (tuning aid for the SpMM window
// kernel, whose product loop reads (val: 8 B broadcast per row, slot: 2 B broadcast per row, panel piece: 16 B) per entry)
//   hipcc --offload-arch=gfx950 -O3 tools/ldsbench.hip -o tools/ldsbench
// Reports cycles of CU time per wave instruction = elapsed cycles / (instructions per wave x waves per CU), for 4 and 8
// waves per CU (one or two 256-thread workgroups), at the 2.4 GHz the clock-rate query reports.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double dbl2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

enum { B128_CONTIG, B128_ROWS8, B128_ROWS4x2, B64_CONTIG, B64_BCAST8, B64_BCAST4, B64_BCAST8_MASK, U16_BCAST8, U16_BCAST4, B32_CONTIG, B64_QUADLANE, NPAT };
static const char *names[NPAT] = {"b128 lane-contiguous", "b128 8 rows x 8 lanes (random 128-B rows)", "2 x b128 16 rows x 4 lanes (random 128-B rows)",
                                  "b64 lane-contiguous", "b64 broadcast per 8 lanes", "b64 broadcast per 4 lanes", "b64 one lane in 8 active",
                                  "u16 broadcast per 8 lanes", "u16 broadcast per 4 lanes", "b32 lane-contiguous", "b64 one entry per lane (quad-strided rows)"};

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// The loop body is address arithmetic of two VALU instructions per read and the read itself (inline asm, result unused), so
// that the LDS pipe and not the VALU is what the waves queue for.
template <int PAT>
__global__ __launch_bounds__(256) void k(int iters, unsigned seed, double *sink, int lds_bytes) {
  extern __shared__ dbl2 lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < lds_bytes / 16; i += 256) lds[i] = dbl2{(double)i, 1.0};
  __syncthreads();
  const unsigned lm = (unsigned)lds_bytes - 1;
  unsigned row, off, active = 1;                 // byte offset of the lane inside its row's 128 bytes; row hash per lane group
  switch (PAT) {
    case B128_CONTIG: row = 0; off = lane * 16; break;
    case B128_ROWS8: row = (lane >> 3) * 0x9E3779B1u; off = (lane & 7) * 16; break;
    case B128_ROWS4x2: row = (lane >> 2) * 0x9E3779B1u; off = (lane & 3) * 32; break;
    case B64_CONTIG: row = 0; off = lane * 8; break;
    case B64_BCAST8: row = 0; off = (lane >> 3) * 216; break;
    case B64_BCAST4: row = 0; off = (lane >> 2) * 216; break;
    case B64_BCAST8_MASK: row = 0; off = (lane >> 3) * 216; active = (lane & 7) == 0; break;
    case U16_BCAST8: row = 0; off = (lane >> 3) * 54; break;
    case U16_BCAST4: row = 0; off = (lane >> 2) * 54; break;
    case B32_CONTIG: row = 0; off = lane * 4; break;
    default: row = 0; off = ((lane >> 2) * 27 + (lane & 3)) * 8; break;
  }
  unsigned addr = ((row >> 8) * 128 + off + (unsigned)(tid >> 6) * 4096 + seed * 128) & lm;
  const unsigned step = 128 * 37 + (row >> 20) * 128;      // whole rows: alignment and the lane pattern inside a row stay
  const unsigned am = PAT == U16_BCAST8 || PAT == U16_BCAST4 ? lm & ~1u : (PAT == B32_CONTIG ? lm & ~3u : lm & ~15u & ~(PAT >= B64_CONTIG ? 0u : 0u));
  u32x4 v4; u32x2 v2; unsigned v1;
  if (active) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        addr = (addr + step) & lm;
        const unsigned a8 = addr & ~7u;
        if (PAT == B128_CONTIG || PAT == B128_ROWS8) asm volatile("ds_read_b128 %0, %1" : "=v"(v4) : "v"(addr & ~15u));
        else if (PAT == B128_ROWS4x2) {
          asm volatile("ds_read_b128 %0, %1" : "=v"(v4) : "v"(addr & ~15u));
          asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(v4) : "v"(addr & ~15u));
        } else if (PAT == U16_BCAST8 || PAT == U16_BCAST4) asm volatile("ds_read_u16 %0, %1" : "=v"(v1) : "v"(addr & ~1u));
        else if (PAT == B32_CONTIG) asm volatile("ds_read_b32 %0, %1" : "=v"(v1) : "v"(addr & ~3u));
        else asm volatile("ds_read_b64 %0, %1" : "=v"(v2) : "v"(a8));
      }
      asm volatile("s_waitcnt lgkmcnt(0)");
    }
  }
  (void)am;
  if (addr == 0xFFFFFFFFu) *sink = 1.0;
}

template <int PAT>
static void run(int wgs_per_cu, double *sink, int clk_khz) {
  const int lds_bytes = 64 * 1024;                 // <= 64 KB: two workgroups per CU fit, one if wgs_per_cu == 1 (grid = #CU)
  const int iters = 2000;
  const int grid = 256 * wgs_per_cu;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((k<PAT>), dim3(grid), dim3(256), lds_bytes, 0, 10, 1u, sink, lds_bytes); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((k<PAT>), dim3(grid), dim3(256), lds_bytes, 0, iters, 1u, sink, lds_bytes);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double cycles = (double)ms * 1e-3 * clk_khz * 1e3;
  const double per = cycles / ((double)iters * 8 * 4 * wgs_per_cu) / (PAT == B128_ROWS4x2 ? 2 : 1);
  printf("%-52s %d waves/CU: %.3f ms  %6.2f cycles of CU time per wave instruction\n", names[PAT], 4 * wgs_per_cu, ms, per);
  fflush(stdout);
}

int main() {
  double *sink; CK(hipMalloc(&sink, 8));
  int clk = 0; CK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
  printf("clock %d kHz\n", clk);
#define ALL(P) run<P>(1, sink, clk); run<P>(2, sink, clk);
  ALL(B128_CONTIG) ALL(B128_ROWS8) ALL(B128_ROWS4x2) ALL(B64_CONTIG) ALL(B64_BCAST8) ALL(B64_BCAST4) ALL(B64_BCAST8_MASK)
  ALL(U16_BCAST8) ALL(U16_BCAST4) ALL(B32_CONTIG) ALL(B64_QUADLANE)
  return 0;
}
