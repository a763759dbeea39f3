// device_reduce.hpp -- wave64 / workgroup / grid reduction building blocks for gfx950.
//
// All reductions carry a double-double partial (hi, lo).  In compensated mode every
// product is split with TwoProd (one FMA) and added with TwoSum (Knuth), i.e. the Dot2
// algorithm of Ogita, Rump & Oishi: the result is as accurate as if accumulated in twice
// the working precision and then rounded, so it does not depend (beyond one ulp) on the
// reduction tree -- different grids, fused vs. unfused kernels and 1 vs. 8 GPUs agree.
// The kernels are HBM-bound (<= 0.125 flop/B); the ~10 extra fp64 VALU ops per element are free.
//
// Compiled with -ffp-contract=off: every fma() below is explicit, nothing else is fused.
#pragma once

#include <hip/hip_runtime.h>

#include "khip_internal.hpp"

namespace khip {

constexpr int kBlock = 256;            // 4 waves of 64
constexpr int kWavesPerBlock = kBlock / 64;

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

struct RedArgs {
  dd *partials;        // [NOUT][kMaxRedBlocks]
  unsigned *ticket;    // one word, zero between launches
  double *results;     // ring base
  dd *results_dd;      // ring base
  int slot;
};

// Grid-level finish: every workgroup publishes its partial with write-through (sc1) agent-scope
// stores, drains them, then takes a ticket; the last arriver re-reads all partials with agent-scope
// loads (L1-bypassing) in a FIXED order and writes the result.  Follows the publish/consume rules of
// cdna_hip_programming.md Guideline 16 ("8-B agent atomics both sides", drain before the flag).
template <int NOUT>
__device__ __forceinline__ void grid_finish(dd (&acc)[NOUT], const RedArgs &ra) {
  __shared__ dd s_w[NOUT][kWavesPerBlock];
  __shared__ int s_last;
  const int nblocks = gridDim.x;
  block_reduce<NOUT>(acc, s_w);
  if (nblocks == 1) {
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
    for (int o = 0; o < NOUT; ++o) {
      dd *p = ra.partials + (size_t)o * kMaxRedBlocks + blockIdx.x;
      __hip_atomic_store(&p->hi, acc[o].hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&p->lo, acc[o].lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned t = __hip_atomic_fetch_add(ra.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == (unsigned)(nblocks - 1));
  }
  __syncthreads();
  if (!s_last) return;
  dd fin[NOUT];
#pragma unroll
  for (int o = 0; o < NOUT; ++o) {
    dd a = {0.0, 0.0};
    for (int b = threadIdx.x; b < nblocks; b += kBlock) {
      const dd *p = ra.partials + (size_t)o * kMaxRedBlocks + b;
      dd v;
      v.hi = __hip_atomic_load(&p->hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v.lo = __hip_atomic_load(&p->lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      a = dd_merge(a, v);
    }
    fin[o] = a;
  }
  __syncthreads();   // s_w reuse
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

inline RedArgs make_red_args(khip_ctx *ctx, int slot) {
  RedArgs ra;
  ra.partials = ctx->partials;
  ra.ticket = ctx->tickets;
  ra.results = ctx->results;
  ra.results_dd = ctx->results_dd;
  ra.slot = slot;
  return ra;
}

}  // namespace khip
