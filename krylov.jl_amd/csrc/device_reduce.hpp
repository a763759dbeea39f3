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
#include "solver_device.hpp"

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

// Cross-lane moves as DPP modifiers (no LDS traffic, unlike ds_bpermute-based __shfl): lanes whose
// source is outside the row / masked off receive 0.0, and dd_merge(v, {0, 0}) is an exact no-op.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ dd dpp_step(dd v) {
  dd o;
  o.hi = dpp_f64<CTRL, ROW_MASK>(v.hi);
  o.lo = dpp_f64<CTRL, ROW_MASK>(v.lo);
  return dd_merge(v, o);
}

constexpr int kResultLane = 63;   // wave_reduce leaves the wave total in the LAST lane

// Fixed-order wave64 fold: inclusive scan inside each row of 16 lanes (row_shr 1, 2, 4, 8), then
// row_bcast:15 into rows 1 and 3, then row_bcast:31 into rows 2 and 3 (the GFX9 DPP reduction).
__device__ __forceinline__ dd wave_reduce(dd v) {
#ifdef KHIP_REDUCE_SHFL           // A/B experiment: ds_bpermute butterfly ending in the last lane
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    dd o;
    o.hi = __shfl_up(v.hi, off, 64);
    o.lo = __shfl_up(v.lo, off, 64);
    if ((int)(threadIdx.x & 63) < off) o = dd{0.0, 0.0};
    v = dd_merge(v, o);
  }
  return v;
#endif
  v = dpp_step<0x111, 0xf>(v);   // row_shr:1
  v = dpp_step<0x112, 0xf>(v);   // row_shr:2
  v = dpp_step<0x114, 0xf>(v);   // row_shr:4
  v = dpp_step<0x118, 0xf>(v);   // row_shr:8
  v = dpp_step<0x142, 0xa>(v);   // row_bcast:15 -> rows 1, 3
  v = dpp_step<0x143, 0xc>(v);   // row_bcast:31 -> rows 2, 3
  return v;   // valid in lane kResultLane
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
  // device-resident loop control (solver_device.hpp); all zero outside "fused = 2" solves
  const long long *stop_seq;
  long long seq;
  int epi;
  void *epi_state;
};

// Streaming-kernel side: every wave folds its lanes and stores one partial per output.
template <int NOUT>
__device__ __forceinline__ void wave_publish(dd (&acc)[NOUT], const RedArgs &ra) {
  const int64_t wid = ra.wave_offset + (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
#pragma unroll
  for (int o = 0; o < NOUT; ++o) {
    dd r = wave_reduce(acc[o]);
    if ((threadIdx.x & 63) == kResultLane) ra.wave_partials[(size_t)o * ra.cap + wid] = r;
  }
}

// The same for kernels whose workgroup is one tile and has LDS to spare at its end (the SpMV windows): the four waves leave
// their lanes' partials in LDS and ONE wave folds all 256 (lane l: threads l, l + 64, l + 128, l + 192 in that order, then the
// wave64 tree).  The double-double tree is ~90 VALU instructions per wave -- as many as a 7-point row block's own arithmetic --
// and three of the four waves now skip it: measured on the fused SpMV + p.Ap at 512^3 (profiles/r04*).  The partial lands
// in the first wave's slot, the other three slots get an exact zero (the finish kernel's input layout is unchanged).
// s_red: kBlock * NOUT dd of LDS that no wave reads any more once every wave has passed the caller's last barrier.
template <int NOUT>
__device__ __forceinline__ void block_publish(dd (&acc)[NOUT], const RedArgs &ra, dd *s_red) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t wid = ra.wave_offset + (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
#pragma unroll
  for (int o = 0; o < NOUT; ++o) s_red[o * kBlock + tid] = acc[o];
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      dd v = s_red[o * kBlock + lane];
      v = dd_merge(v, s_red[o * kBlock + 64 + lane]);
      v = dd_merge(v, s_red[o * kBlock + 128 + lane]);
      v = dd_merge(v, s_red[o * kBlock + 192 + lane]);
      v = wave_reduce(v);
      if (lane == kResultLane) ra.wave_partials[(size_t)o * ra.cap + wid] = v;
    }
  } else if (lane == kResultLane) {
#pragma unroll
    for (int o = 0; o < NOUT; ++o) ra.wave_partials[(size_t)o * ra.cap + wid] = dd{0.0, 0.0};
  }
}

// Workgroup reduction of NOUT partials; result valid in thread 0.
template <int NOUT>
__device__ __forceinline__ void block_reduce(dd (&acc)[NOUT], dd (*s_w)[kWavesPerBlock]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 0; o < NOUT; ++o) {
    dd r = wave_reduce(acc[o]);
    if (lane == kResultLane) s_w[o][wave] = r;
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
  if (seq_skip(ra.stop_seq, ra.seq)) return;
  const int G = gridDim.x;
  const int64_t chunk = (P + G - 1) / G;
  const int64_t lo = chunk * blockIdx.x;
  const int64_t hi = (lo + chunk < P) ? lo + chunk : P;
  dd acc[NOUT];
#pragma unroll
  for (int o = 0; o < NOUT; ++o) {
    // A thread folds its partials i = lo + tid, + 256, ... IN THAT ORDER (the result's bits depend on it).  The loads are issued
    // eight at a time before the first merge (round 6): the loop used to wait for every 16-byte load before issuing the next --
    // eight dependent round trips through L2 per launch (launch_finish gives a thread 8 partials), 11 us per finish kernel where
    // the MGS step it follows takes 70 us at 256^3 (profiles/r06_rocprofv3_kernel_stats.csv).
    dd a = {0.0, 0.0};
    const dd *src = ra.wave_partials + (size_t)o * ra.cap;
    for (int64_t base = lo + threadIdx.x; base < hi; base += (int64_t)kBlock * 8) {
      dd v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int64_t i = base + (int64_t)j * kBlock;
        v[j] = src[i < hi ? i : base];                 // (a clamped index: the value is not merged)
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (base + (int64_t)j * kBlock < hi) a = dd_merge(a, v[j]);
    }
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
      if (ra.epi) solver_epilogue(ra.epi, ra.epi_state, ra.results + ra.slot, ra.seq);
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
    if (ra.epi) solver_epilogue(ra.epi, ra.epi_state, ra.results + ra.slot, ra.seq);
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
  ra.stop_seq = ctx->ctl.stop_seq;
  ra.seq = ctx->ctl.seq;
  ra.epi = ctx->ctl.epi;
  ra.epi_state = ctx->ctl.epi_state;
  return ra;
}

}  // namespace khip
