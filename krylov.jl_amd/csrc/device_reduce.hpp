// device_reduce.hpp -- wave64 / workgroup / grid reduction building blocks for gfx950.
//
// All reductions carry a double-double partial (hi, lo).  In compensated mode every
// product is split with TwoProd (one FMA) and added with TwoSum (Knuth), i.e. the Dot2
// algorithm of Ogita, Rump & Oishi: the result is as accurate as if accumulated in twice
// the working precision and then rounded, so it does not depend (beyond one ulp) on the
// reduction tree -- different grids, fused vs. unfused kernels and 1 vs. 8 GPUs agree.
// The kernels are HBM-bound (<= 0.125 flop/B); the ~10 extra fp64 VALU ops per element are free.
//
// Structure (measured on MI355X, profiles/r01*): streaming kernels want LOOP-FREE launches with
// short-lived workgroups (10^5..10^6 of them), and any in-kernel publish/ticket protocol per
// workgroup costs more than the streaming itself (dot 4.3 TB/s with it, 7.1 TB/s without).  So the
// streaming kernel only does a wave64 shuffle reduction and ONE plain 16-byte store per wave
// (`wave_publish`), then exits; a tiny second kernel (`reduce_finish_kernel`, <= 256 workgroups)
// folds the per-wave partials in a FIXED order -- chunk per workgroup, strided within the chunk,
// shuffle tree, then a ticketed last-arriver fold of the <= 256 workgroup partials with agent-scope
// release/acquire (cdna_hip_programming.md Guideline 16).  Deterministic run to run.
//
// Compiled with -ffp-contract=off: every fma() below is explicit, nothing else is fused.
#pragma once

#include <hip/hip_runtime.h>

#include "khip_internal.hpp"

namespace khip {

constexpr int kBlock = 256;            // 4 waves of 64
constexpr int kWavesPerBlock = kBlock / 64;
constexpr int kFinishMaxBlocks = 256;

__device__ __forceinline__ void two_sum(double a, double b, double &s, double &e) {
  s = a + b;
  double z = s - a;
  e = (a - (s - z)) + (b - z);
}

template <bool COMP>
__device__ __forceinline__ void acc_prod(dd &a, double x, double y) {
  if (COMP) {
    double p = x * y;
    double e = fma(x, y, -p);
    double s, err;
    two_sum(a.hi, p, s, err);
    a.hi = s;
    a.lo += err + e;
  } else {
    a.hi = fma(x, y, a.hi);
  }
}

__device__ __forceinline__ dd dd_merge(dd a, dd b) {
  double s, err;
  two_sum(a.hi, b.hi, s, err);
  dd r;
  r.hi = s;
  r.lo = (a.lo + b.lo) + err;
  return r;
}

__device__ __forceinline__ dd wave_reduce(dd v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    dd o;
    o.hi = __shfl_down(v.hi, off, 64);
    o.lo = __shfl_down(v.lo, off, 64);
    v = dd_merge(v, o);
  }
  return v;   // valid in lane 0
}

struct RedArgs {
  dd *wave_partials;   // [NOUT][cap]  one per wave of the streaming kernel
  dd *blk_partials;    // [NOUT][kFinishMaxBlocks]
  unsigned *ticket;    // one word, zero between launches
  double *results;     // ring base
  dd *results_dd;      // ring base
  int64_t cap;
  int64_t wave_offset;  // first partial index of this launch (several launches may feed one finish)
  int slot;
};

// Streaming-kernel side: every wave folds its lanes and stores one partial per output.
template <int NOUT>
__device__ __forceinline__ void wave_publish(dd (&acc)[NOUT], const RedArgs &ra) {
  const int64_t wid = ra.wave_offset + (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
#pragma unroll
  for (int o = 0; o < NOUT; ++o) {
    dd r = wave_reduce(acc[o]);
    if ((threadIdx.x & 63) == 0) ra.wave_partials[(size_t)o * ra.cap + wid] = r;
  }
}

// Workgroup reduction of NOUT partials; result valid in thread 0.
template <int NOUT>
__device__ __forceinline__ void block_reduce(dd (&acc)[NOUT], dd (*s_w)[kWavesPerBlock]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 0; o < NOUT; ++o) {
    dd r = wave_reduce(acc[o]);
    if (lane == 0) s_w[o][wave] = r;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      dd t = s_w[o][0];
#pragma unroll
      for (int w = 1; w < kWavesPerBlock; ++w) t = dd_merge(t, s_w[o][w]);
      acc[o] = t;
    }
  }
}

// Second kernel: P per-wave partials -> results[slot .. slot+NOUT).
template <int NOUT>
__global__ __launch_bounds__(kBlock) void reduce_finish_kernel(RedArgs ra, int64_t P) {
  __shared__ dd s_w[NOUT][kWavesPerBlock];
  __shared__ int s_last;
  const int G = gridDim.x;
  const int64_t chunk = (P + G - 1) / G;
  const int64_t lo = chunk * blockIdx.x;
  const int64_t hi = (lo + chunk < P) ? lo + chunk : P;
  dd acc[NOUT];
#pragma unroll
  for (int o = 0; o < NOUT; ++o) {
    dd a = {0.0, 0.0};
    for (int64_t i = lo + threadIdx.x; i < hi; i += kBlock) a = dd_merge(a, ra.wave_partials[(size_t)o * ra.cap + i]);
    acc[o] = a;
  }
  block_reduce<NOUT>(acc, s_w);
  if (G == 1) {
    if (threadIdx.x == 0) {
#pragma unroll
      for (int o = 0; o < NOUT; ++o) {
        ra.results[ra.slot + o] = acc[o].hi + acc[o].lo;
        ra.results_dd[ra.slot + o] = acc[o];
      }
    }
    return;
  }
  if (threadIdx.x == 0) {
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {   // write-through (sc1) agent-scope stores
      dd *p = ra.blk_partials + (size_t)o * kFinishMaxBlocks + blockIdx.x;
      __hip_atomic_store(&p->hi, acc[o].hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&p->lo, acc[o].lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain before the ticket (Guideline 16 compiler hazard)
    unsigned t = __hip_atomic_fetch_add(ra.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == (unsigned)(G - 1));
  }
  __syncthreads();
  if (!s_last) return;
  dd fin[NOUT];
#pragma unroll
  for (int o = 0; o < NOUT; ++o) {
    fin[o] = dd{0.0, 0.0};
    if ((int)threadIdx.x < G) {                        // L1-bypassing agent-scope loads
      const dd *p = ra.blk_partials + (size_t)o * kFinishMaxBlocks + threadIdx.x;
      fin[o].hi = __hip_atomic_load(&p->hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      fin[o].lo = __hip_atomic_load(&p->lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  block_reduce<NOUT>(fin, s_w);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      ra.results[ra.slot + o] = fin[o].hi + fin[o].lo;
      ra.results_dd[ra.slot + o] = fin[o];
    }
    __hip_atomic_store(ra.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// host side (blas1.hip)
int ensure_reduction_scratch(khip_ctx *ctx, int64_t nwaves, int nout);
int launch_finish(khip_ctx *ctx, int64_t nwaves, int nout, int slot);

inline RedArgs make_red_args(khip_ctx *ctx, int slot) {
  RedArgs ra;
  ra.wave_partials = ctx->partials;
  ra.blk_partials = ctx->partials2;
  ra.ticket = ctx->tickets;
  ra.results = ctx->results;
  ra.results_dd = ctx->results_dd;
  ra.cap = ctx->red_cap1;
  ra.wave_offset = 0;
  ra.slot = slot;
  return ra;
}

}  // namespace khip
