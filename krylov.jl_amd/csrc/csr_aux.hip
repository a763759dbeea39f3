// csr_aux.hip -- SpMM for row-major panels, CSR set-up helpers (row statistics, index shift, halo
// gather / remap kernels) and the device-side generators of the benchmark operators.
#include <algorithm>
#include <type_traits>
#include <vector>

#include "spmv_common.hpp"

namespace khip {

// ---------------------------------------------------------------- SpMM ----------
// Y(m x p, row-major) = A * X(n x p, row-major); P = lanes per row (>= p, power of two).  The P lanes
// first fetch P consecutive (val, col) entries of the row in one coalesced access each, then every
// entry is broadcast with a shuffle and all P gathers of X rows (one contiguous 8p-byte line each)
// are issued before the first use; accumulation in stored order with a rounded multiply and a rounded
// add => column j of Y is bit-identical to the SpMV of column j.
//
// Tile order (a.sweep_s > 0): PLANE SWEEP.  The gathers of a 3-D stencil row reach the panel rows of the two
// neighbouring grid planes; in the natural order those are S tiles back, long evicted from the XCD's 4 MiB L2
// (216^3, p = 16: three planes of panel rows = 18 MB), so every panel row is fetched ~3x (FETCH_SIZE 16.6 GB for
// 5.85 GB algorithmic) and the kernel runs at the L2<->fabric rate.  With the sweep every XCD takes a column of W
// consecutive tiles and walks it through all planes: the k+-1 reuse distance is W tiles.  S is padded to a
// multiple of 8 W with empty tiles; any (S, W) only permutes the tiles, so Y never depends on it.
template <int P>
__global__ __launch_bounds__(kBlock) void spmm_kernel(SpmvArgs a, int p) {
  constexpr int RPB = kBlock / P;
  const int sub = threadIdx.x / P, c = threadIdx.x % P;
  int64_t first_tile = blockIdx.x, tile_stride = gridDim.x;
  if (a.sweep_s > 0) {
    const int S = a.sweep_s, W = a.sweep_w;          // the host pads S to a multiple of 8 W when it sizes the grid
    const int64_t ntiles = (a.row_hi - a.row_lo + RPB - 1) / RPB;
    const int64_t K = (ntiles + S - 1) / S;
    const int64_t b = blockIdx.x;
    const int pxcd = (int)(b & 7);
    const int64_t l = b >> 3, per = K * W;
    const int64_t tt = l / per, rem = l - tt * per;
    const int64_t k = rem / W;
    const int w = (int)(rem - k * W);
    const int64_t ti = (tt * 8 + pxcd) * W + w;
    if (ti >= S) return;                           // padding tile
    first_tile = k * S + ti;
    if (first_tile >= ntiles) return;
    tile_stride = ntiles;                          // exactly one tile per workgroup
  }
  for (int64_t row = a.row_lo + first_tile * RPB + sub; row < a.row_hi; row += tile_stride * RPB) {
    const int64_t s = a.rowptr[row], e = a.rowptr[row + 1];
    double acc = 0.0;
    for (int64_t base = s; base < e; base += P) {
      const int cnt = (int)((e - base) < P ? (e - base) : P);
      const bool mine = c < cnt;
      const double myv = mine ? a.val[base + c] : 0.0;
      const int32_t myc = mine ? a.col[base + c] : 0;
      double xs[P];
#pragma unroll
      for (int t = 0; t < P; ++t) {
        const int32_t cc = __shfl(myc, t, P);
        const bool own = cc < a.n_owned;                 // else: a ghost panel row of a distributed handle
        const double *src = own ? a.x : a.ghost;
        const int64_t r = own ? (int64_t)cc : (int64_t)cc - a.n_owned;
        xs[t] = (t < cnt && c < p) ? src[r * p + c] : 0.0;
      }
#pragma unroll
      for (int t = 0; t < P; ++t) {
        const double vv = __shfl(myv, t, P);
        if (t < cnt) {
          const double prod = vv * xs[t];
          acc = acc + prod;
        }
      }
    }
    if (c < p) a.y[row * p + c] = acc;
  }
}

// Two panel columns per lane (16-byte gathers): L = p / 2 lanes per row, twice the rows per wave and half the
// gather instructions per row of spmm_kernel -- the kernel is bound by the number of gather instructions (27 per
// row for the 27-point operator), not by bytes.  Same per-column arithmetic and order => Y bit-identical.
template <int L>
__global__ __launch_bounds__(kBlock) void spmm2_kernel(SpmvArgs a, int p) {
  constexpr int RPB = kBlock / L;
  const int sub = threadIdx.x / L, c = threadIdx.x % L;
  const bool col_ok = 2 * c + 1 < p;
  int64_t first_tile = blockIdx.x, tile_stride = gridDim.x;
  if (a.sweep_s > 0) {                               // plane sweep of the tiles, as in spmm_kernel
    const int S = a.sweep_s, W = a.sweep_w;
    const int64_t ntiles = (a.row_hi - a.row_lo + RPB - 1) / RPB;
    const int64_t K = (ntiles + S - 1) / S;
    const int64_t b = blockIdx.x;
    const int pxcd = (int)(b & 7);
    const int64_t l = b >> 3, per = K * W;
    const int64_t tt = l / per, rem = l - tt * per;
    const int64_t k = rem / W;
    const int w = (int)(rem - k * W);
    const int64_t ti = (tt * 8 + pxcd) * W + w;
    if (ti >= S) return;
    first_tile = k * S + ti;
    if (first_tile >= ntiles) return;
    tile_stride = ntiles;
  }
  for (int64_t row = a.row_lo + first_tile * RPB + sub; row < a.row_hi; row += tile_stride * RPB) {
    const int64_t s = a.rowptr[row], e = a.rowptr[row + 1];
    double acc0 = 0.0, acc1 = 0.0;
    for (int64_t base = s; base < e; base += L) {
      const int cnt = (int)((e - base) < L ? (e - base) : L);
      const bool mine = c < cnt;
      const double myv = mine ? a.val[base + c] : 0.0;
      const int32_t myc = mine ? a.col[base + c] : 0;
      dbl2 xs[L];
#pragma unroll
      for (int t = 0; t < L; ++t) {
        const int32_t cc = __shfl(myc, t, L);
        xs[t] = dbl2{0.0, 0.0};
        const bool own = cc < a.n_owned;
        const double *src = own ? a.x : a.ghost;
        const int64_t r = own ? (int64_t)cc : (int64_t)cc - a.n_owned;
        if (t < cnt && col_ok) xs[t] = *reinterpret_cast<const dbl2 *>(src + r * p + 2 * c);
      }
#pragma unroll
      for (int t = 0; t < L; ++t) {
        const double vv = __shfl(myv, t, L);
        if (t < cnt) {
          const double p0 = vv * xs[t].x, p1 = vv * xs[t].y;
          acc0 = acc0 + p0;
          acc1 = acc1 + p1;
        }
      }
    }
    if (col_ok) *reinterpret_cast<dbl2 *>(a.y + row * p + 2 * c) = dbl2{acc0, acc1};
  }
}

// Panel-row window in LDS (default for even p on operators with the locality): spmm2_kernel is bound by the gather
// rate of the vector L1 -- 27 gathers of one 128-byte panel row per matrix row for the 27-point operator,
// ~19 B/clk/CU -- although consecutive matrix rows of a banded operator reach mostly the SAME panel rows (32 rows of
// the 27-point stencil touch 9 x 34 distinct ones, not 27 x 32).  Once per handle and lane count L, spmm_window_build
// (below) finds, for every group of RPB = 256 / L consecutive rows, the list of distinct columns and stores it with a
// 16-bit slot number per nonzero (+2 B/nonzero and 4 B per list entry of HBM).  The kernel then
//   1. copies the distinct panel rows of its group ONCE from global memory into LDS (list order = column order, so the
//      copy runs over contiguous memory) together with the group's (val, slot) entries (fully coalesced),
//   2. runs the row products out of LDS: per nonzero two broadcast reads (val, slot), one 16-byte panel-row read, the
//      rounded multiply and the rounded add of spmm2_kernel, in stored order.
// A group whose rows reach more panel rows or hold more nonzeros than the window takes is flagged at build time and
// goes down the direct-gather path.  Same operations in the same order => Y is bit-identical to spmm2_kernel /
// spmm_kernel / p SpMVs.
//
// What shaped the kernel (216^3, 27 points, p = 16; tools/archive/spmm_window_check.py, rocprofv3 SQ counters):
//  * one-shot workgroups: 2.5 ms (direct gathers 3.05 ms) -- ~16 waves per CU, the LDS limit, cannot hide the three
//    dependent latencies row pointer / list -> panel row -> product;
//  * persistent workgroups with the next groups prefetched into registers: no gain until the prefetch stages were
//    STRAIGHT-LINE, LOAD-ONLY code -- any use of a loaded value (a sign extension, a flag test) or any branch between
//    issue and use makes the compiler emit s_waitcnt vmcnt(0), i.e. wait for everything just issued; __syncthreads()
//    does the same through its fence, hence lds_barrier();
//  * after that the kernel was bound by INSTRUCTION ISSUE (~1000 instructions per wave and group, VALU 76 % busy):
//    64-bit address arithmetic (v_mul_lo_u32, v_mad_u64_u32, owned/ghost selects) and per-entry lane broadcasts.
//    Hence shifts instead of multiplies when p = 2 L, a DIST template, scalar loads for the per-group words, and
//    (val, slot) in LDS where a broadcast read replaces three cross-lane moves.
// LDS for the panel rows and workgroups per CU the kernel is compiled for.  44 KB / 2 is the measured optimum: 40 KB / 3
// (three waves per SIMD, 168 VGPRs with 4 spilled) is 4 % faster at p = 16 but 2-3x slower at p = 8 and 4, whose
// instantiations spill heavily under the tighter register budget (profiles/r02_spmm_experiments.log).
#ifndef KHIP_WIN_SPREAD
#define KHIP_WIN_SPREAD 1
#endif
#ifndef KHIP_WIN_PIPE
#define KHIP_WIN_PIPE 1
#endif
#ifndef KHIP_WIN_KB
#define KHIP_WIN_KB 44
#endif
#ifndef KHIP_WIN_WGS
#define KHIP_WIN_WGS 2
#endif
constexpr int kWinPanelBytes = KHIP_WIN_KB * 1024;

template <int L>
struct WinShape {
  static constexpr int RPB = kBlock / L;                       // matrix rows per workgroup
  static constexpr int CAP = kWinPanelBytes / (16 * L);        // distinct panel rows the window holds
  static constexpr int STRIDE = (CAP + RPB - 1) / RPB * RPB;   // list entries per row group = panel rows of the LDS window
  static constexpr int NU = STRIDE / RPB;                      // list entries per lane
  static constexpr int ENTRIES = RPB * 32;                     // (val, slot) entries of a row group staged in LDS
  static constexpr int NE = ENTRIES / kBlock;                  // ... per lane
  static constexpr int HT = CAP <= 256 ? 512 : (CAP <= 512 ? 1024 : (CAP <= 1024 ? 2048 : 4096));   // >= CAP + kBlock, power of 2
  static constexpr size_t kLds = (size_t)STRIDE * L * 16 + (size_t)(ENTRIES + 8) * 8 + (size_t)(ENTRIES + 8) * 2;
};

struct WinArgs {
  const int32_t *list;       // [groups][STRIDE]: the distinct columns of a row group (ascending-ish), padded with 0
  const uint16_t *slot;      // per nonzero: position of its column in the group's list
  const int32_t *flag;       // per row group: 1 = direct-gather group
  int64_t groups;
  // plane sweep of the row groups (sweep_S > 0): XCD x = blockIdx % 8 walks columns of sweep_W consecutive groups through
  // all sweep_K planes of sweep_S groups, so that the panel rows of plane k are in that XCD's L2 when the groups of the
  // planes k - 1, k, k + 1 ask for them (the 192 workgroups of an XCD span three planes of one column at any time).
  int sweep_S, sweep_W, sweep_K, sweep_cols;   // sweep_cols = columns per XCD (the plane is padded to 8 W sweep_cols groups)
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup fence + s_barrier and the fence
// drains vmcnt as well, i.e. it would wait for the prefetches of the next groups at every barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Value of lane K of every group of L consecutive lanes (L = 2, 4, 8), by DPP moves instead of an LDS broadcast read: an LDS
// instruction costs the CU ~5 cycles whatever its width (tools/ldsbench.hip), and the product loop spent two of them per
// entry (val, slot) next to the one panel-row read; fetched L entries at a time and spread over the row's lanes by one or
// two v_mov_dpp each, (val, slot) cost 2 / L of an LDS instruction per entry.
template <int L, int K>
__device__ __forceinline__ int group_bcast(int v) {
  static_assert(L == 2 || L == 4 || L == 8, "group_bcast: L = 2, 4, 8");
  if (L == 2) {
    constexpr int QP = K | (K << 2) | ((K + 2) << 4) | ((K + 2) << 6);                  // quad_perm [K, K, K+2, K+2]
    return __builtin_amdgcn_update_dpp(0, v, QP, 0xF, 0xF, true);      // old = 0 + bound_ctrl: no register to initialise
  }
  constexpr int Q = K & 3, QP = Q | (Q << 2) | (Q << 4) | (Q << 6);                     // quad_perm [Q, Q, Q, Q]
  int t = __builtin_amdgcn_update_dpp(0, v, QP, 0xF, 0xF, true);
  if (L == 8)       // the other quad of the octet takes the value across row_half_mirror (lane i <- lane 7 - i)
    t = __builtin_amdgcn_update_dpp(t, t, 0x141, 0xF, K < 4 ? 0xA : 0x5, false);
  return t;
}
template <int L, int K>
__device__ __forceinline__ double group_bcast(double v) {
  return __hiloint2double(group_bcast<L, K>(__double2hiint(v)), group_bcast<L, K>(__double2loint(v)));
}
// entries t0 .. t0 + 7 of a row's (val, slot) stream: fetched L at a time (lane c takes entry f L + c), spread by group_bcast
template <int L, int K>
struct BatchSpread {
  __device__ static __forceinline__ void run(const double (&mv)[8 / L], const int (&ms)[8 / L], double (&vv)[8], int (&sl)[8]) {
    vv[K] = group_bcast<L, K % L>(mv[K / L]);
    sl[K] = group_bcast<L, K % L>(ms[K / L]);
    BatchSpread<L, K + 1>::run(mv, ms, vv, sl);
  }
};
template <int L>
struct BatchSpread<L, 8> {
  __device__ static __forceinline__ void run(const double (&)[8 / L], const int (&)[8 / L], double (&)[8], int (&)[8]) {}
};

// Phase stamps of one wave (tools/archive/spmm_trace.py builds a variant of the library with -DKHIP_WIN_TRACE; compiled out otherwise):
// wave 0 of workgroup 5 writes the shader clock at the phase boundaries of its iterations 8..23.  What it showed
// (profiles/r02_spmm_trace.log, 216^3 x 16): of ~8500 cycles per row group, ~1200 go into writing the window to LDS,
// ~2400 into ISSUING the 34 prefetch loads (the CU's texture path takes ~17 cycles per wave instruction and 8 waves issue
// at once), ~3200 into the products -- 27 entry steps x ~115 cycles, which is what 8 waves x (8 B val + 2 B slot + 16 B
// panel piece per lane) cost the one LDS of the CU: the product phase is LDS-bandwidth bound, not latency bound (hiding the
// (val, slot) read latency behind the products changed nothing).
#ifdef KHIP_WIN_TRACE
__device__ unsigned long long g_win_trace[16 * 8];
#define KHIP_STAMP(slot)                                                                                   \
  do {                                                                                                     \
    if (trace_on && trace_it >= 8 && trace_it < 24) g_win_trace[(trace_it - 8) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define KHIP_STAMP(slot) do { } while (0)
#endif

template <int NU, int NE>
struct WinSet {                // what a lane holds of one row group in flight
  int key[NU];                 // stage A: its list entries,
  int32_t s, e;                //          the extent of its row (raw loaded words: no arithmetic before the next iteration),
  int32_t gs, flag;            //          the group's first nonzero and its direct flag (uniform: scalar loads)
  dbl2 v[NU];                  // stage B: its 16-byte pieces of the panel rows
  double ev[NE];               //          and its share of the group's (val, slot) entries
  unsigned short es[NE];
};

template <int L, bool DIST, bool POW2>        // POW2: p == 2 L, a panel row is 16 L bytes and its offset a shift
__global__ __launch_bounds__(kBlock) void spmm_window_kernel(SpmvArgs a, WinArgs w, int p) {
  using W = WinShape<L>;
  constexpr int NU = W::NU, NE = W::NE;
  using Set = WinSet<NU, NE>;
  extern __shared__ dbl2 win_xs[];                                           // [STRIDE][L] panel rows
  double *win_val = reinterpret_cast<double *>(win_xs + W::STRIDE * L);      // [ENTRIES + 8]
  unsigned short *win_slot = reinterpret_cast<unsigned short *>(win_val + W::ENTRIES + 8);
  const int tid = threadIdx.x, sub = tid / L, c = tid % L;
  const bool col_ok = 2 * c + 1 < p;
  const int piece = col_ok ? 2 * c : 0;
  const int nnz_last = (int)(a.nnz_bound - 1);
  if (tid < 8) { win_val[W::ENTRIES + tid] = 0.0; win_slot[W::ENTRIES + tid] = 0; }   // spare entries the tail batch may read
  // virtual index l of this workgroup's t-th group -> group number (w.groups = "no group": a padding slot of the sweep)
  const bool sweep = w.sweep_S > 0;
  const int xcd = (int)(blockIdx.x & 7);
  const int64_t lstep = sweep ? (int64_t)(gridDim.x >> 3) : (int64_t)gridDim.x;
  const int64_t lend = sweep ? (int64_t)w.sweep_cols * w.sweep_K * w.sweep_W : w.groups;
  auto phys = [&](int64_t l) -> int64_t {
    if (!sweep) return l < w.groups ? l : w.groups;
    if (l >= lend) return w.groups;
    const int64_t per = (int64_t)w.sweep_K * w.sweep_W;
    const int64_t tt = l / per, rem = l - tt * per;
    const int64_t k = rem / w.sweep_W, ww = rem - k * w.sweep_W;
    const int64_t pos = (tt * 8 + xcd) * w.sweep_W + ww;              // position inside the plane
    const int64_t g = k * w.sweep_S + pos;
    return (pos < w.sweep_S && g < w.groups) ? g : w.groups;
  };

  // Stages A and B are straight-line code: loads only, no use of a loaded value, no branch.  Indices past the end are
  // clamped and the results ignored.
  auto stage_a = [&](int64_t l, Set &z) {
    const int64_t g = phys(l);
    const int64_t gg = g < w.groups ? g : w.groups - 1;
    const int32_t *lst = w.list + gg * (int64_t)W::STRIDE;
#pragma unroll
    for (int j = 0; j < NU; ++j) z.key[j] = lst[sub + j * W::RPB];
    const int64_t row = a.row_lo + gg * W::RPB + sub;
    const int64_t rr = row < a.row_hi ? row : a.row_hi - 1;
    z.s = a.rowptr[rr];
    z.e = a.rowptr[rr + 1];
    z.gs = a.rowptr[a.row_lo + gg * W::RPB];
    z.flag = w.flag[gg];
  };
  auto stage_b = [&](Set &z) {
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      unsigned q = (unsigned)(z.gs + tid + k * kBlock);
      q = q < (unsigned)nnz_last ? q : (unsigned)nnz_last;
      z.ev[k] = a.val[q];
      z.es[k] = w.slot[q];
    }
#pragma unroll
    for (int j = 0; j < NU; ++j) {
      const unsigned key = (unsigned)z.key[j];
      const double *src = a.x;
      uint64_t r = key;
      if (DIST) {
        const bool own = (int64_t)key < a.n_owned;
        src = own ? a.x : a.ghost;
        r = own ? r : r - (uint64_t)a.n_owned;
      }
      uint64_t off;
      if (POW2) {
        constexpr int SH = L == 2 ? 5 : (L == 4 ? 6 : (L == 8 ? 7 : (L == 16 ? 8 : 9)));      // log2(16 L)
        off = r << SH;
      } else {
        off = r * (uint64_t)(unsigned)(p * 8);
      }
      z.v[j] = *reinterpret_cast<const dbl2 *>(reinterpret_cast<const char *>(src) + off + piece * 8);
    }
  };
  // stage C of group g out of `cur`, while B runs for g + G into `nxt` and A for g + 2 G reuses the key registers of `cur`.
  // The two sets swap roles every iteration (the loop below is unrolled by two): a register copy of a set would wait
  // for the loads that are still filling it.
#ifdef KHIP_WIN_TRACE
  const bool trace_on = blockIdx.x == 5 && tid == 0;
  int trace_it = 0;
#endif
  auto iteration = [&](int64_t l, Set &cur, Set &nxt) {
    KHIP_STAMP(0);
    const int64_t g = phys(l);
    const int64_t row = a.row_lo + g * W::RPB + sub;
    const bool row_ok = row < a.row_hi;
    const int sC = cur.s, eC = row_ok ? cur.e : cur.s;
    const int gsC = __builtin_amdgcn_readfirstlane(cur.gs);
    const bool directC = __builtin_amdgcn_readfirstlane(cur.flag) != 0;
    if (!directC) {
#pragma unroll
      for (int j = 0; j < NU; ++j) win_xs[(sub + j * W::RPB) * L + c] = cur.v[j];
#pragma unroll
      for (int k = 0; k < NE; ++k) {
        win_val[tid + k * kBlock] = cur.ev[k];
        win_slot[tid + k * kBlock] = cur.es[k];
      }
    }
    KHIP_STAMP(1);
    lds_barrier();
    KHIP_STAMP(2);
    stage_b(nxt);
    stage_a(l + 2 * lstep, cur);
    KHIP_STAMP(3);
    if (directC) {
      if (row_ok) {
        double acc0 = 0.0, acc1 = 0.0;
        for (int64_t base = sC; base < eC; base += L) {
          const int cnt = (int)((eC - base) < L ? (eC - base) : L);
          const bool mine = c < cnt;
          const double myv = mine ? a.val[base + c] : 0.0;
          const int32_t myc = mine ? a.col[base + c] : 0;
          dbl2 xg[L];
#pragma unroll
          for (int t = 0; t < L; ++t) {
            const int32_t cc = __shfl(myc, t, L);
            xg[t] = dbl2{0.0, 0.0};
            const bool own = cc < a.n_owned;
            const double *src = own ? a.x : a.ghost;
            const int64_t r = own ? (int64_t)cc : (int64_t)cc - a.n_owned;
            if (t < cnt && col_ok) xg[t] = *reinterpret_cast<const dbl2 *>(src + r * p + 2 * c);
          }
#pragma unroll
          for (int t = 0; t < L; ++t) {
            const double vv = __shfl(myv, t, L);
            if (t < cnt) {
              const double p0 = vv * xg[t].x, p1 = vv * xg[t].y;
              acc0 = acc0 + p0;
              acc1 = acc1 + p1;
            }
          }
        }
        if (col_ok) *reinterpret_cast<dbl2 *>(a.y + row * p + 2 * c) = dbl2{acc0, acc1};
      }
    } else {
      // every lane of a row reads the row's (val, slot) stream at the same LDS address (a broadcast), then its own
      // 16 bytes of the panel row.  Batches of 8 entries: all reads of a batch are issued before its first product.
      const int len = eC - sC;
      const double *lv = win_val + (sC - gsC);
      const unsigned short *lsl = win_slot + (sC - gsC);
      const char *xc = reinterpret_cast<const char *>(win_xs + c);
      double acc0 = 0.0, acc1 = 0.0;
      constexpr bool SPREAD = KHIP_WIN_SPREAD != 0 && L >= 4;   // (val, slot) fetched LS entries at a time and spread by DPP moves (L = 2: 3 % slower, same box)
      constexpr int LS = KHIP_WIN_SPREAD == 8 && L == 8 ? 8 : (L < 4 ? L : 4);   // quads: one v_mov_dpp per word (octets: two)
      const int cs = c & (LS - 1);
      const int len0 = __builtin_amdgcn_readfirstlane(len);
      if (__ballot(len != len0) == 0 && SPREAD && KHIP_WIN_PIPE) {
        // the wave's rows are equally long (interior of a stencil): scalar loop bounds, and two batches of 8 entries in flight --
        // a wave gets an LDS read back every ~45 cycles (tools/ldsbench.hip), so the panel-row reads of batch b + 1 and the
        // (val, slot) fetch of batch b + 2 are issued BEFORE the products of batch b.  Reads run up to 7 entries past the row
        // (into the next rows' entries or the 8 spare ones: valid slots all); the products stop at the uniform bound.
        constexpr int NF = 8 / LS;
        struct Batch { double vv[8]; dbl2 xv[8]; };
        double mv[NF];
        int ms[NF];
        auto fetch = [&](int t0) {
#pragma unroll
          for (int f = 0; f < NF; ++f) { mv[f] = lv[t0 + f * LS + cs]; ms[f] = (int)lsl[t0 + f * LS + cs] * (16 * L); }
        };
        auto spread_issue = [&](Batch &b) {
          int sl[8];
          BatchSpread<LS, 0>::run(mv, ms, b.vv, sl);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            b.xv[k] = *reinterpret_cast<const dbl2 *>(xc + sl[k]);
          }
        };
        auto consume = [&](const Batch &b, int n) {
          if (n >= 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const double p0 = b.vv[k] * b.xv[k].x, p1 = b.vv[k] * b.xv[k].y;
              acc0 = acc0 + p0;
              acc1 = acc1 + p1;
            }
          } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              if (k < n) {
                const double p0 = b.vv[k] * b.xv[k].x, p1 = b.vv[k] * b.xv[k].y;
                acc0 = acc0 + p0;
                acc1 = acc1 + p1;
              }
            }
          }
        };
        const int nb = (len0 + 7) >> 3;
        if (nb > 0) {
          Batch A, B;
          fetch(0);
          spread_issue(A);
          if (nb > 1) fetch(8);
          for (int b = 0;;) {      // A holds batch b, (mv, ms) the fetch of batch b + 1
            if (b + 1 < nb) { spread_issue(B); if (b + 2 < nb) fetch(8 * (b + 2)); }
            consume(A, len0 - 8 * b);
            if (++b >= nb) break;
            if (b + 1 < nb) { spread_issue(A); if (b + 2 < nb) fetch(8 * (b + 2)); }
            consume(B, len0 - 8 * b);
            if (++b >= nb) break;
          }
        }
      } else if (__ballot(len != len0) == 0) {
        // the wave's rows are equally long (interior of a stencil): scalar loop bounds, no masks
        int t0 = 0;
        for (; t0 + 8 <= len0; t0 += 8) {
          double vv[8];
          int sl[8];
          dbl2 xv[8];
          if (SPREAD) {
            double mv[8 / LS];
            int ms[8 / LS];
#pragma unroll
            for (int f = 0; f < 8 / LS; ++f) { mv[f] = lv[t0 + f * LS + cs]; ms[f] = (int)lsl[t0 + f * LS + cs] * (16 * L); }
            BatchSpread<LS, 0>::run(mv, ms, vv, sl);
          } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) { vv[k] = lv[t0 + k]; sl[k] = (int)lsl[t0 + k] * (16 * L); }
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) xv[k] = *reinterpret_cast<const dbl2 *>(xc + sl[k]);     // sl: byte offset of the panel row
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const double p0 = vv[k] * xv[k].x, p1 = vv[k] * xv[k].y;
            acc0 = acc0 + p0;
            acc1 = acc1 + p1;
          }
        }
        if (t0 < len0) {
          // tail: the reads run past the row (into the next rows' entries or the 8 spare ones: valid slots all),
          // the products stop at the uniform bound
          const int n = len0 - t0;
          double vv[8];
          int sl[8];
          dbl2 xv[8];
          if (SPREAD) {
            double mv[8 / LS];
            int ms[8 / LS];
#pragma unroll
            for (int f = 0; f < 8 / LS; ++f) { mv[f] = lv[t0 + f * LS + cs]; ms[f] = (int)lsl[t0 + f * LS + cs] * (16 * L); }
            BatchSpread<LS, 0>::run(mv, ms, vv, sl);
          } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) { vv[k] = lv[t0 + k]; sl[k] = (int)lsl[t0 + k] * (16 * L); }
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) xv[k] = *reinterpret_cast<const dbl2 *>(xc + sl[k]);     // sl: byte offset of the panel row
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            if (k < n) {
              const double p0 = vv[k] * xv[k].x, p1 = vv[k] * xv[k].y;
              acc0 = acc0 + p0;
              acc1 = acc1 + p1;
            }
          }
        }
      } else {
        for (int t0 = 0; __any(t0 < len); t0 += 8) {
          double vv[8];
          int sl[8];
          dbl2 xv[8];
          if (SPREAD) {
            double mv[8 / LS];
            int ms[8 / LS];
#pragma unroll
            for (int f = 0; f < 8 / LS; ++f) {
              const int t = t0 + f * LS + cs < len ? t0 + f * LS + cs : 0;
              mv[f] = lv[t];
              ms[f] = (int)lsl[t] * (16 * L);
            }
            BatchSpread<LS, 0>::run(mv, ms, vv, sl);
          } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const int t = t0 + k < len ? t0 + k : 0;       // past the end of the row: its first entry, not used
              vv[k] = lv[t];
              sl[k] = (int)lsl[t] * (16 * L);
            }
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) xv[k] = *reinterpret_cast<const dbl2 *>(xc + sl[k]);     // sl: byte offset of the panel row
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            if (t0 + k < len) {
              const double p0 = vv[k] * xv[k].x, p1 = vv[k] * xv[k].y;
              acc0 = acc0 + p0;
              acc1 = acc1 + p1;
            }
          }
        }
      }
      KHIP_STAMP(4);
      if (row_ok && col_ok) *reinterpret_cast<dbl2 *>(a.y + row * p + 2 * c) = dbl2{acc0, acc1};
    }
    KHIP_STAMP(5);
    lds_barrier();
    KHIP_STAMP(6);
#ifdef KHIP_WIN_TRACE
    ++trace_it;
#endif
  };

  Set P, Q;
  int64_t l = sweep ? (int64_t)(blockIdx.x >> 3) : (int64_t)blockIdx.x;
  stage_a(l, P);
  stage_b(P);
  stage_a(l + lstep, Q);
  for (;;) {
    if (l >= lend) break;
    iteration(l, P, Q);
    l += lstep;
    if (l >= lend) break;
    iteration(l, Q, P);
    l += lstep;
  }
}

#ifdef KHIP_WIN_TRACE
}  // namespace khip
extern "C" int khip_debug_win_trace(unsigned long long *out128) {
  return hipMemcpyFromSymbol(out128, HIP_SYMBOL(khip::g_win_trace), sizeof(unsigned long long) * 128) == hipSuccess ? 0 : -1;
}
namespace khip {
#endif

int spmm_window_build(khip_ctx *ctx, khip_csr *A, int L);   // below
int launch_spmm_tile(khip_ctx *ctx, const khip_csr *A, const SpmvArgs &a, int p, bool slices);   // spmm_tile.hip

template <int L>
static void launch_window(khip_ctx *ctx, const khip_csr *A, const SpmvArgs &a, int p) {
  using W = WinShape<L>;
  const int64_t groups = (A->m + W::RPB - 1) / W::RPB;
  WinArgs w{A->win_list, A->win_slot, A->win_flag, groups, 0, 0, 0, 0};
  const size_t lds = W::kLds;
  int per_cu = (int)((size_t)(160 * 1024) / lds);                                 // LDS-limited residency ...
  if (per_cu > KHIP_WIN_WGS) per_cu = KHIP_WIN_WGS;                               // ... and the register budget the kernel is compiled for
  if (per_cu < 1) per_cu = 1;
  int64_t grid = (int64_t)ctx->num_cu * per_cu * 3;         // 3 x the resident workgroups: the queue evens out the tail (measured 2 %)
  if (ctx->tune.spmm_window_grid > 0) grid = ctx->tune.spmm_window_grid;
  if (grid > groups) grid = groups;
  if (ctx->tune.spmm_win_sweep != 0 && grid >= 64) {
    // plane sweep: S = groups per grid plane, from the plane distance of the operator (rows); worthwhile when the panel rows
    // a row reaches (2 planes apart) exceed what an XCD's L2 keeps (4 MiB)
    const int64_t plane_rows = A->plane_rows > 0 ? A->plane_rows : A->band;
    const int64_t S = ctx->tune.spmm_sweep_s > 0 ? ctx->tune.spmm_sweep_s : (plane_rows + W::RPB / 2) / W::RPB;
    int Wc = ctx->tune.spmm_sweep_w > 0 ? ctx->tune.spmm_sweep_w : 64;
    const size_t reach = (size_t)plane_rows * 2 * (size_t)p * sizeof(double);
    if (S >= 8 * (int64_t)Wc && S < groups && (ctx->tune.spmm_sweep_s > 0 || reach > ((size_t)2 << 20))) {
      grid &= ~(int64_t)7;                                                        // whole workgroups per XCD
      w.sweep_S = (int)S; w.sweep_W = Wc;
      w.sweep_K = (int)((groups + S - 1) / S);
      w.sweep_cols = (int)((S + 8 * (int64_t)Wc - 1) / (8 * (int64_t)Wc));
    }
  }
  const bool dist = a.ghost != a.x, pow2 = p == 2 * L;
  const dim3 gd((unsigned)grid), bd(kBlock);
  if (lds > 64 * 1024) {                 // beyond the default dynamic-LDS limit: raise it for the instantiation about to run
    const void *fn = dist ? (pow2 ? (const void *)spmm_window_kernel<L, true, true> : (const void *)spmm_window_kernel<L, true, false>)
                          : (pow2 ? (const void *)spmm_window_kernel<L, false, true> : (const void *)spmm_window_kernel<L, false, false>);
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  if (dist && pow2) hipLaunchKernelGGL((spmm_window_kernel<L, true, true>), gd, bd, lds, ctx->stream, a, w, p);
  else if (dist) hipLaunchKernelGGL((spmm_window_kernel<L, true, false>), gd, bd, lds, ctx->stream, a, w, p);
  else if (pow2) hipLaunchKernelGGL((spmm_window_kernel<L, false, true>), gd, bd, lds, ctx->stream, a, w, p);
  else hipLaunchKernelGGL((spmm_window_kernel<L, false, false>), gd, bd, lds, ctx->stream, a, w, p);
}

int launch_spmm(khip_ctx *ctx, const khip_csr *A, const double *X, double *Y, int p) {
  if (p < 1 || p > 64) { set_error("spmm: 1 <= p <= 64 required (got %d)", p); return KHIP_ERR_INVALID; }
  SpmvArgs a;
  a.hole_lo = INT64_MAX; a.hole_len = 0;
  a.rowptr = A->rowptr; a.blockptr = nullptr; a.col = A->col; a.val = A->val; a.x = X; a.ghost = X; a.y = Y;
  a.n_owned = (int64_t)1 << 40;                      // single GPU: every column is owned
  if (A->dist && ctx->comm) {                        // row-partitioned: fetch the remote panel rows first (no overlap yet)
    KHIP_TRY(comm_halo_exchange_begin(ctx, A, X, p));
    KHIP_TRY(comm_halo_exchange_end(ctx, A));
    if (A->n_ghost > 0) { a.ghost = A->ghost_w; a.n_owned = A->m; }
  }
  a.row_lo = 0; a.row_hi = A->m; a.xcd_remap = 0; a.nt_y = 0; a.dot_early = 0; a.fake_gather = 0; a.tiles_per_block = 1;
  a.nnz_bound = A->nnz + kPad;
  ProfScope prof_scope(ctx, kProfSpmm);             // ctx option profile_spmv: HIP events around the product's kernels (after the halo exchange)
  int P = 4;
  while (P < p) P <<= 1;
  const int rpb = kBlock / P;
  int64_t want = (A->m + rpb - 1) / rpb;
  int64_t gcap = 1 << 22;
  int grid = (int)(want < gcap ? want : gcap);     // loop-free: one row group per workgroup slot
  if (grid < 1) grid = 1;
  a.sweep_s = 0; a.sweep_w = 0;
  a.stop_seq = nullptr; a.seq = 0; a.dotw = nullptr; a.dot_sq = 0; a.stage_cap = 0;
  a.tmpl_id = nullptr; a.tmpl_off = nullptr; a.tmpl_val = nullptr; a.tmpl_cnt = nullptr; a.tmpl_T = 0; a.tmpl_K = 0;
  {   // plane sweep when the band is wide enough for the k+-1 panel rows to fall out of L2
    const int W = ctx->tune.spmm_sweep_w > 0 ? ctx->tune.spmm_sweep_w : 64;
    int64_t S = ctx->tune.spmm_sweep_s > 0 ? ctx->tune.spmm_sweep_s : (A->band + rpb / 2) / rpb;
    const size_t window = (size_t)A->band * 2 * (size_t)p * sizeof(double);          // panel rows between the extremes of a row
    if (ctx->tune.spmm_sweep != 0 && S >= 8 * W && (ctx->tune.spmm_sweep_s > 0 || window > ((size_t)2 << 20))) {
      const int64_t Spad = (S + 8 * W - 1) / (8 * W) * (8 * W);
      const int64_t K = (want + S - 1) / S;
      if (K * Spad <= gcap) {
        a.sweep_s = (int)S; a.sweep_w = W;
        grid = (int)(K * Spad);
      }
    }
  }
  if (ctx->tune.spmm_wide && (p & 1) == 0 && p >= 4 && (reinterpret_cast<uintptr_t>(X) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(Y) & 15) == 0) {
    int L = 2;
    while (2 * L < p) L <<= 1;                       // L lanes cover 2 L >= p columns
    const int rpb2 = kBlock / L;
    int64_t want2 = (A->m + rpb2 - 1) / rpb2;
    int grid2 = (int)(want2 < gcap ? (want2 > 0 ? want2 : 1) : gcap);
    a.sweep_s = 0;
    if (ctx->tune.spmm_sweep != 0 && !ctx->tune.spmm_window) {      // direct 16-byte gathers in plane-sweep tile order (experiment)
      const int W2 = ctx->tune.spmm_sweep_w > 0 ? ctx->tune.spmm_sweep_w : 64;
      const int64_t plane_rows = A->plane_rows > 0 ? A->plane_rows : A->band;
      const int64_t S2 = ctx->tune.spmm_sweep_s > 0 ? ctx->tune.spmm_sweep_s : (plane_rows + rpb2 / 2) / rpb2;
      if (S2 >= 8 * (int64_t)W2) {
        const int64_t Spad = (S2 + 8 * W2 - 1) / (8 * W2) * (8 * W2);
        const int64_t K2 = (want2 + S2 - 1) / S2;
        if (K2 * Spad <= gcap) { a.sweep_s = (int)S2; a.sweep_w = W2; grid2 = (int)(K2 * Spad); }
      }
    }
    const bool pow2 = (p & (p - 1)) == 0;
    if (ctx->tune.spmm_tile && pow2 && p >= 8 && A->m > 0 && A->nnz > 0) {
      // wave-private windows, LDS-DMA, grid-tile row groups (spmm_tile.hip).  p = 16 always; p = 8 on grid operators with rows
      // of 16 entries and more (27-point: 1.08 vs 1.41 ms; the 7-point and the banded + random operators are 4 % slower); p = 32
      // and wider either as ONE launch with p / 4 lanes per row (7-point: 1.51 vs 2.33 ms) or as p / 16 launches of the
      // 16-column kernel over column slices (p = 32 with rows of 16 entries and more, 27-point: 2.52 vs 3.23 ms in one launch --
      // the window reads of wide panels load the LDS pipes and halve the residency; p >= 64 always: 5.0 vs 13.0 ms;
      // profiles/r03h_spmm_tile_p.log, r03i_spmm_tile_slices.log).  spmm_tile = 2 takes the tile kernel at every width
      // it has; spmm_tile_slices = 1 / -1 forces / forbids the slices.
      khip_csr *At = const_cast<khip_csr *>(A);
      if (At->tile_state == 0) optional_build(spmm_tile_build(ctx, At));
      const bool long_rows = A->nnz >= 16 * A->m;
      const bool slices = p >= 32 && ctx->tune.spmm_tile_slices >= 0 && (p > 32 || ctx->tune.spmm_tile_slices > 0 || long_rows);
      const bool fits = slices || (size_t)At->tile_cap * 8 * (size_t)p <= (size_t)160 * 1024;
      const bool want = p == 16 || p > 32 || ctx->tune.spmm_tile >= 2 || (p == 8 && At->tile_grid == 1 && long_rows) ||
                        (p == 32 && (slices ? long_rows : At->tile_grid == 1));
      if (At->tile_state == 1 && fits && want && (p <= 32 || slices)) return launch_spmm_tile(ctx, A, a, p, slices);
    }


    if (ctx->tune.spmm_window && want2 <= gcap && A->m > 0) {      // panel-row window in LDS: one row group per workgroup
      khip_csr *Aw = const_cast<khip_csr *>(A);
      if (Aw->win_L != L && Aw->win_L != -L) KHIP_TRY(spmm_window_build(ctx, Aw, L));
      if (Aw->win_L == L) {
        switch (L) {
          case 2: launch_window<2>(ctx, A, a, p); break;
          case 4: launch_window<4>(ctx, A, a, p); break;
          case 8: launch_window<8>(ctx, A, a, p); break;
          case 16: launch_window<16>(ctx, A, a, p); break;
          default: launch_window<32>(ctx, A, a, p); break;
        }
        KHIP_CHECK_HIP(hipGetLastError());
        return KHIP_OK;
      }
    }
    switch (L) {
      case 2: hipLaunchKernelGGL((spmm2_kernel<2>), dim3(grid2), dim3(kBlock), 0, ctx->stream, a, p); break;
      case 4: hipLaunchKernelGGL((spmm2_kernel<4>), dim3(grid2), dim3(kBlock), 0, ctx->stream, a, p); break;
      case 8: hipLaunchKernelGGL((spmm2_kernel<8>), dim3(grid2), dim3(kBlock), 0, ctx->stream, a, p); break;
      case 16: hipLaunchKernelGGL((spmm2_kernel<16>), dim3(grid2), dim3(kBlock), 0, ctx->stream, a, p); break;
      default: hipLaunchKernelGGL((spmm2_kernel<32>), dim3(grid2), dim3(kBlock), 0, ctx->stream, a, p); break;
    }
    KHIP_CHECK_HIP(hipGetLastError());
    return KHIP_OK;
  }
  switch (P) {
    case 4: hipLaunchKernelGGL((spmm_kernel<4>), dim3(grid), dim3(kBlock), 0, ctx->stream, a, p); break;
    case 8: hipLaunchKernelGGL((spmm_kernel<8>), dim3(grid), dim3(kBlock), 0, ctx->stream, a, p); break;
    case 16: hipLaunchKernelGGL((spmm_kernel<16>), dim3(grid), dim3(kBlock), 0, ctx->stream, a, p); break;
    case 32: hipLaunchKernelGGL((spmm_kernel<32>), dim3(grid), dim3(kBlock), 0, ctx->stream, a, p); break;
    default: hipLaunchKernelGGL((spmm_kernel<64>), dim3(grid), dim3(kBlock), 0, ctx->stream, a, p); break;
  }
  KHIP_CHECK_HIP(hipGetLastError());
  return KHIP_OK;
}

// ---------------------------------------------------------------- row statistics -
__global__ __launch_bounds__(kBlock) void row_stats_kernel(const int32_t *rowptr, int64_t m, int *max_out) {
  int mx = 0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock) {
    int len = rowptr[i + 1] - rowptr[i];
    mx = len > mx ? len : mx;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    int o = __shfl_down(mx, off, 64);
    mx = o > mx ? o : mx;
  }
  if ((threadIdx.x & 63) == 0) atomicMax(max_out, mx);
}

// largest |column - row| over the handle (band half-width): the plane distance of a 3-D stencil
__global__ __launch_bounds__(kBlock) void band_kernel(const int32_t *rowptr, const int32_t *col, int64_t m, int *max_out) {
  int mx = 0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock) {
    const int32_t s = rowptr[i], e = rowptr[i + 1];
    if (e > s) {                                   // sorted rows: the extremes are the first and last entry; unsorted: scan
      int d0 = (int)i - col[s], d1 = col[e - 1] - (int)i;
      d0 = d0 < 0 ? -d0 : d0; d1 = d1 < 0 ? -d1 : d1;
      mx = d0 > mx ? d0 : mx; mx = d1 > mx ? d1 : mx;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    int o = __shfl_down(mx, off, 64);
    mx = o > mx ? o : mx;
  }
  if ((threadIdx.x & 63) == 0) atomicMax(max_out, mx);
}

// diag[i] = A[i, i + row0] (0 when the entry is structurally absent)
__global__ __launch_bounds__(kBlock) void diagonal_kernel(const int32_t *rowptr, const int32_t *col, const double *val, int64_t m,
                                                           int64_t col_shift, double *diag) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock) {
    double d = 0.0;
    for (int64_t j = rowptr[i]; j < rowptr[i + 1]; ++j)
      if ((int64_t)col[j] == i + col_shift) d = val[j];
    diag[i] = d;
  }
}

int launch_diagonal(khip_ctx *ctx, const khip_csr *A, double *diag) {
  if (A->m == 0) return KHIP_OK;
  int64_t want = (A->m + kBlock - 1) / kBlock;
  // distributed handles have their columns renumbered to [owned | ghost]: the diagonal is column i again
  hipLaunchKernelGGL(diagonal_kernel, dim3((unsigned)(want < 65536 ? want : 65536)), dim3(kBlock), 0, ctx->stream, A->rowptr, A->col,
                     A->val, A->m, (int64_t)0, diag);
  KHIP_CHECK_HIP(hipGetLastError());
  return KHIP_OK;
}

__global__ __launch_bounds__(kBlock) void blockptr_kernel(const int32_t *rowptr, int64_t m, int64_t nb, int32_t *bp) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i <= nb; i += (int64_t)gridDim.x * kBlock) {
    const int64_t r = i * 256 < m ? i * 256 : m;
    bp[i] = rowptr[r];
  }
}

// Structural validation of user arrays (they are trusted by every kernel afterwards: window loads through buffer
// descriptors, x gathers, the ghost remap): row pointers start at 0, never decrease, end at nnz; columns lie in [0, n).
// bad[0] = smallest offending row (m + 1 = none), bad[1] = kind (1 row pointer, 2 column).
__global__ __launch_bounds__(kBlock) void csr_validate_kernel(const int32_t *rowptr, const int32_t *col, int64_t m, int64_t n,
                                                               int64_t nnz, unsigned long long *bad) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock) {
    const int64_t s = rowptr[i], e = rowptr[i + 1];
    bool ok = s >= 0 && e >= s && e <= nnz && (i > 0 || s == 0) && (i < m - 1 || e == nnz);
    int kind = ok ? 0 : 1;
    if (ok)
      for (int64_t q = s; q < e; ++q) {
        const int32_t c = col[q];
        if (c < 0 || (int64_t)c >= n) { kind = 2; break; }
      }
    if (kind) {
      const unsigned long long old = atomicMin(&bad[0], (unsigned long long)i);
      if ((unsigned long long)i < old) bad[1] = (unsigned long long)kind;
    }
  }
}

// histogram of the row lengths 0..256 (longer rows: bin 256)
__global__ __launch_bounds__(kBlock) void row_len_hist_kernel(const int32_t *rowptr, int64_t m, unsigned long long *hist) {
  __shared__ unsigned int h[257];
  for (int i = threadIdx.x; i < 257; i += kBlock) h[i] = 0;
  __syncthreads();
  for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < m; r += (int64_t)gridDim.x * kBlock) {
    const int len = rowptr[r + 1] - rowptr[r];
    atomicAdd(&h[len < 256 ? len : 256], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 257; i += kBlock) if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}
// first row >= lo with exactly `len` entries
__global__ __launch_bounds__(kBlock) void row_find_kernel(const int32_t *rowptr, int64_t m, int64_t lo, int len, unsigned long long *out) {
  for (int64_t r = lo + (int64_t)blockIdx.x * kBlock + threadIdx.x; r < m; r += (int64_t)gridDim.x * kBlock) {
    if (rowptr[r + 1] - rowptr[r] == len) { atomicMin(out, (unsigned long long)r); break; }
  }
}

int csr_finalize(khip_ctx *ctx, khip_csr *A) {
  A->mean_row_nnz = A->m > 0 ? (double)A->nnz / (double)A->m : 0.0;
  A->max_row_nnz = 0;
  if (A->m == 0) {
    if (A->nnz != 0) { set_error("csr: %lld nonzeros in an operator without rows", (long long)A->nnz); return KHIP_ERR_INVALID; }
    return KHIP_OK;
  }
  {
    unsigned long long *d_bad = nullptr;
    KHIP_CHECK_HIP(hipMalloc(&d_bad, 2 * sizeof(unsigned long long)));
    unsigned long long h_bad[2] = {(unsigned long long)A->m + 1, 0ull};
    hipError_t e = hipMemcpyAsync(d_bad, h_bad, sizeof(h_bad), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
      int64_t wantv = (A->m + kBlock - 1) / kBlock;
      hipLaunchKernelGGL(csr_validate_kernel, dim3((unsigned)(wantv < 65536 ? wantv : 65536)), dim3(kBlock), 0, ctx->stream,
                         A->rowptr, A->col, A->m, A->n, A->nnz, d_bad);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h_bad, d_bad, sizeof(h_bad), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_bad);
    if (e != hipSuccess) { set_error("csr validation: %s", hipGetErrorString(e)); return KHIP_ERR_HIP; }
    if (h_bad[0] <= (unsigned long long)A->m) {
      set_error(h_bad[1] == 2 ? "csr: row %llu has a column index outside [0, %lld) (wrong index_base?)"
                              : "csr: row pointers are not a monotone sequence from 0 to nnz at row %llu (n = %lld; wrong index_base?)",
                h_bad[0], (long long)A->n);
      return KHIP_ERR_INVALID;
    }
  }
  {
    const int64_t nb = (A->m + 255) / 256;
    KHIP_CHECK_HIP(hipMalloc(&A->blockptr, sizeof(int32_t) * (size_t)(nb + 1)));
    int64_t want = (nb + 1 + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(blockptr_kernel, dim3((unsigned)(want < 1024 ? want : 1024)), dim3(kBlock), 0, ctx->stream, A->rowptr,
                       A->m, nb, A->blockptr);
    KHIP_CHECK_HIP(hipGetLastError());
  }
  int *d = ctx->scratch_word;
  KHIP_CHECK_HIP(hipMemsetAsync(d, 0, sizeof(int), ctx->stream));
  int64_t want = (A->m + kBlock - 1) / kBlock;
  int grid = (int)(want < 2048 ? want : 2048);
  hipLaunchKernelGGL(row_stats_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, A->rowptr, A->m, d);
  KHIP_CHECK_HIP(hipGetLastError());
  int h = 0;
  KHIP_CHECK_HIP(hipMemcpyAsync(&h, d, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  A->max_row_nnz = h;
  KHIP_CHECK_HIP(hipMemsetAsync(d, 0, sizeof(int), ctx->stream));
  hipLaunchKernelGGL(band_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, A->rowptr, A->col, A->m, d);
  KHIP_CHECK_HIP(hipGetLastError());
  KHIP_CHECK_HIP(hipMemcpyAsync(&h, d, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  A->band = h;
  // Plane distance of a 3-D operator, for the plane-sweep orders: a longest row near the middle is looked at on the host;
  // its positive offsets fall into clusters (a new one starts where an offset more than doubles): 1 | n1 | n1^2 for the
  // 7-point operator, 1 | n1-1..n1+1 | n1^2-n1-1..n1^2+n1+1 for the 27-point one.  plane_rows = centre of the last cluster.
  A->plane_rows = 0;
  A->line_rows = 0;
  if (A->m >= 128 && A->max_row_nnz >= 2) {
    // an INTERIOR row: the first row of the second half whose length is the most frequent one (<= 256) -- the rows around
    // m / 2 alone may all lie on a face of the grid (216^3: row m / 2 is (0, 0, 108)) and lack whole clusters
    unsigned long long *d_hist = nullptr;
    KHIP_CHECK_HIP(hipMalloc(&d_hist, 258 * sizeof(unsigned long long)));
    struct Free { unsigned long long *&p; ~Free() { (void)hipFree(p); } } fr{d_hist};
    KHIP_CHECK_HIP(hipMemsetAsync(d_hist, 0, 258 * sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(row_len_hist_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, A->rowptr, A->m, d_hist);
    KHIP_CHECK_HIP(hipGetLastError());
    unsigned long long hh[258];
    KHIP_CHECK_HIP(hipMemcpyAsync(hh, d_hist, sizeof(hh), hipMemcpyDeviceToHost, ctx->stream));
    KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    int mode = 2;
    for (int l = 2; l <= 256; ++l) if (hh[l] > hh[mode]) mode = l;
    if (hh[mode] > 0) {
      unsigned long long init = (unsigned long long)A->m;
      KHIP_CHECK_HIP(hipMemcpyAsync(d_hist + 257, &init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
      hipLaunchKernelGGL(row_find_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, A->rowptr, A->m, A->m / 2, mode, d_hist + 257);
      KHIP_CHECK_HIP(hipGetLastError());
      unsigned long long found = init;
      KHIP_CHECK_HIP(hipMemcpyAsync(&found, d_hist + 257, sizeof(found), hipMemcpyDeviceToHost, ctx->stream));
      KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
      if (found < (unsigned long long)A->m) {
        const int64_t row = (int64_t)found;
        int32_t rp[2];
        KHIP_CHECK_HIP(hipMemcpy(rp, A->rowptr + row, sizeof(rp), hipMemcpyDeviceToHost));
        const int len = rp[1] - rp[0];
        std::vector<int32_t> cols((size_t)len);
        KHIP_CHECK_HIP(hipMemcpy(cols.data(), A->col + rp[0], sizeof(int32_t) * (size_t)len, hipMemcpyDeviceToHost));
        std::vector<int64_t> pos;
        for (int32_t cidx : cols) if ((int64_t)cidx > row) pos.push_back((int64_t)cidx - row);
        std::sort(pos.begin(), pos.end());
        if (!pos.empty()) {
          size_t start = 0, second = 0, second_end = 0;      // second cluster = the neighbouring grid line (n1-1..n1+1)
          int clusters = 1;
          for (size_t i = 1; i < pos.size(); ++i)
            if (pos[i] > 2 * pos[i - 1] + 2) {
              ++clusters;
              if (clusters == 2) second = i;
              if (clusters == 3) second_end = i;
              start = i;
            }
          A->plane_rows = (pos[start] + pos.back()) / 2;
          if (clusters >= 2) {
            if (clusters == 2) second_end = pos.size();
            A->line_rows = (pos[second] + pos[second_end - 1]) / 2;
          }
        }
      }
    }
  }
  return KHIP_OK;
}

__global__ __launch_bounds__(kBlock) void index_shift_kernel(int32_t *data, int64_t n, int32_t delta) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) data[i] += delta;
}
int launch_index_shift(khip_ctx *ctx, int32_t *data, int64_t n, int32_t delta) {
  if (n <= 0 || delta == 0) return KHIP_OK;
  int64_t want = (n + kBlock - 1) / kBlock;
  int grid = (int)(want < 4096 ? want : 4096);
  hipLaunchKernelGGL(index_shift_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, data, n, delta);
  KHIP_CHECK_HIP(hipGetLastError());
  return KHIP_OK;
}

// ---------------------------------------------------------------- halo helpers ---
__global__ __launch_bounds__(kBlock) void gather_kernel(int64_t n, const int32_t *idx, const double *x, double *out, int width) {
  const int64_t total = n * width;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int64_t e = i / width;
    const int c = (int)(i - e * width);
    out[i] = x[(int64_t)idx[e] * width + c];
  }
}

int launch_gather(khip_ctx *ctx, int64_t n, const int32_t *idx, const double *x, double *out, int width) {
  if (n <= 0) return KHIP_OK;
  int64_t want = (n * width + kBlock - 1) / kBlock;
  int grid = (int)(want < 4096 ? want : 4096);
  hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, n, idx, x, out, width);
  KHIP_CHECK_HIP(hipGetLastError());
  return KHIP_OK;
}

__global__ __launch_bounds__(kBlock) void collect_offrank_kernel(const int32_t *col, int64_t nnz, int64_t row0,
                                                                  int64_t row1, int32_t *out,
                                                                  unsigned long long *count, int64_t cap) {
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < nnz; j += (int64_t)gridDim.x * kBlock) {
    const int64_t c = col[j];
    if (c < row0 || c >= row1) {
      unsigned long long k = atomicAdd(count, 1ull);
      if ((int64_t)k < cap) out[k] = (int32_t)c;
    }
  }
}

int launch_collect_offrank(khip_ctx *ctx, const khip_csr *A, int64_t row0, int64_t row1, int32_t *out_dev,
                           unsigned long long *count_dev, int64_t cap) {
  if (A->nnz == 0) return KHIP_OK;
  int64_t want = (A->nnz + kBlock - 1) / kBlock;
  int grid = (int)(want < 4096 ? want : 4096);
  hipLaunchKernelGGL(collect_offrank_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, A->col, A->nnz, row0, row1,
                     out_dev, count_dev, cap);
  KHIP_CHECK_HIP(hipGetLastError());
  return KHIP_OK;
}

__global__ __launch_bounds__(kBlock) void col_remap_kernel(int32_t *col, int64_t nnz, int64_t row0, int64_t m,
                                                            const int32_t *ghost_sorted, int64_t n_ghost) {
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < nnz; j += (int64_t)gridDim.x * kBlock) {
    const int64_t c = col[j];
    if (c >= row0 && c < row0 + m) {
      col[j] = (int32_t)(c - row0);
    } else {
      int64_t lo = 0, hi = n_ghost;   // lower_bound
      while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (ghost_sorted[mid] < c) lo = mid + 1; else hi = mid;
      }
      col[j] = (int32_t)(m + lo);
    }
  }
}

int launch_col_remap(khip_ctx *ctx, khip_csr *A, const int32_t *ghost_sorted_dev, int64_t n_ghost) {
  csr_free_codes(A);        // the coded column stream (colcode.hip) is rebuilt from the renumbered columns
  csr_free_delta(A);        // ... and so is the block-delta stream (coldelta.hip)
  if (A->nnz == 0) return KHIP_OK;
  int64_t want = (A->nnz + kBlock - 1) / kBlock;
  int grid = (int)(want < 4096 ? want : 4096);
  hipLaunchKernelGGL(col_remap_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, A->col, A->nnz, A->row0, A->m,
                     ghost_sorted_dev, n_ghost);
  KHIP_CHECK_HIP(hipGetLastError());
  return KHIP_OK;
}

// Gather mode (all-gather of x before the product, comm.cpp): the ghost region is the all-gather's receive buffer, rank r's
// slice at [r * maxm, r * maxm + m_r).  Owned columns -> [0, m), a column owned by rank r at offset o -> m + r * maxm + o.
__global__ __launch_bounds__(kBlock) void col_remap_gather_kernel(int32_t *col, int64_t nnz, int64_t row0, int64_t m,
                                                                   const int64_t *row_starts, int nranks, int64_t maxm) {
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < nnz; j += (int64_t)gridDim.x * kBlock) {
    const int64_t c = col[j];
    if (c >= row0 && c < row0 + m) {
      col[j] = (int32_t)(c - row0);
    } else {
      int lo = 0, hi = nranks;            // owner: last r with row_starts[r] <= c
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (row_starts[mid] <= c) lo = mid; else hi = mid;
      }
      col[j] = (int32_t)(m + (int64_t)lo * maxm + (c - row_starts[lo]));
    }
  }
}

int launch_col_remap_gather(khip_ctx *ctx, khip_csr *A, const int64_t *row_starts_dev, int nranks, int64_t maxm) {
  csr_free_codes(A);
  csr_free_delta(A);
  if (A->nnz == 0) return KHIP_OK;
  int64_t want = (A->nnz + kBlock - 1) / kBlock;
  int grid = (int)(want < 4096 ? want : 4096);
  hipLaunchKernelGGL(col_remap_gather_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, A->col, A->nnz, A->row0, A->m,
                     row_starts_dev, nranks, maxm);
  KHIP_CHECK_HIP(hipGetLastError());
  return KHIP_OK;
}

// rows whose (remapped) columns reach into the ghost region: largest such row in the lower half,
// smallest in the upper half -> [lo, hi) is guaranteed interior.
__global__ __launch_bounds__(kBlock) void ghost_range_kernel(const int32_t *rowptr, const int32_t *col, int64_t m,
                                                              unsigned long long *lo_hi) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock) {
    bool b = false;
    for (int64_t j = rowptr[i]; j < rowptr[i + 1]; ++j) b |= (col[j] >= m);
    if (b) {
      if (i < m / 2) atomicMax(&lo_hi[0], (unsigned long long)(i + 1));
      else atomicMin(&lo_hi[1], (unsigned long long)i);
    }
  }
}

int launch_row_ghost_range(khip_ctx *ctx, const khip_csr *A, int64_t *lo_hi_host) {
  unsigned long long *d = nullptr;
  KHIP_CHECK_HIP(hipMalloc(&d, 2 * sizeof(unsigned long long)));
  unsigned long long init[2] = {0ull, (unsigned long long)A->m};
  KHIP_CHECK_HIP(hipMemcpyAsync(d, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
  if (A->m > 0) {
    int64_t want = (A->m + kBlock - 1) / kBlock;
    int grid = (int)(want < 4096 ? want : 4096);
    hipLaunchKernelGGL(ghost_range_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, A->rowptr, A->col, A->m, d);
    KHIP_CHECK_HIP(hipGetLastError());
  }
  unsigned long long out[2];
  KHIP_CHECK_HIP(hipMemcpyAsync(out, d, sizeof(out), hipMemcpyDeviceToHost, ctx->stream));
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  KHIP_CHECK_HIP(hipFree(d));
  lo_hi_host[0] = (int64_t)out[0];
  lo_hi_host[1] = (int64_t)out[1];
  if (lo_hi_host[1] < lo_hi_host[0]) lo_hi_host[1] = lo_hi_host[0];
  return KHIP_OK;
}

// ---------------------------------------------------------------- generators -----
// Device restatement of the benchmark operators (test/get_div_grad.jl:8-25, test/test_utils.jl:160-169
// and the cfg-5 27-point operator documented in DESIGN.md); parity-tested bit-exact against the oracle.
__device__ __forceinline__ double stencil_coef(int kind, int d1, int d2, int d3) {
  const int ab = abs(d1) + abs(d2) + abs(d3);
  if (kind == 0) return ab == 0 ? 6.0 : (ab == 1 ? -1.0 : 0.0);
  if (kind == 1) {
    if (ab == 0) return 12.0;
    if (ab != 1) return 0.0;
    if (d1 == -1) return -1.0;
    if (d1 == 1) return -2.0;
    if (d2 == -1) return -2.0;
    if (d2 == 1) return -4.0;
    if (d3 == -1) return -1.0;
    return -2.0;
  }
  if (ab == 0) return 16.0;
  const double w = (ab == 1) ? 1.0 : (ab == 2 ? 0.5 : 0.25);
  const int lead = d3 != 0 ? d3 : (d2 != 0 ? d2 : d1);
  return -w * (lead > 0 ? 1.25 : 0.75);
}

template <bool FILL>
__global__ __launch_bounds__(kBlock) void stencil_kernel(int kind, int n1, int n2, int n3, int64_t row0, int64_t m,
                                                          int32_t *rowptr, int32_t *col, double *val) {
  for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < m; r += (int64_t)gridDim.x * kBlock) {
    const int64_t row = row0 + r;
    const int i1 = (int)(row % n1), i2 = (int)((row / n1) % n2), i3 = (int)(row / ((int64_t)n1 * n2));
    int64_t k = FILL ? rowptr[r] : 0;
    int cnt = 0;
    for (int d3 = -1; d3 <= 1; ++d3) {
      const int j3 = i3 + d3;
      if (j3 < 0 || j3 >= n3) continue;
      for (int d2 = -1; d2 <= 1; ++d2) {
        const int j2 = i2 + d2;
        if (j2 < 0 || j2 >= n2) continue;
        for (int d1 = -1; d1 <= 1; ++d1) {
          const int j1 = i1 + d1;
          if (j1 < 0 || j1 >= n1) continue;
          const double v = stencil_coef(kind, d1, d2, d3);
          if (v == 0.0) continue;
          if (FILL) {
            col[k] = (int32_t)((int64_t)j1 + (int64_t)n1 * j2 + (int64_t)n1 * n2 * j3);
            val[k] = v;
            ++k;
          }
          ++cnt;
        }
      }
    }
    if (!FILL) rowptr[r] = cnt;   // counts; scanned afterwards
  }
}

// exclusive scan of int32 counts (3 phases: per-tile sums, scan of tile sums, add back)
constexpr int kScanTile = 2048;   // elements per workgroup (8 per lane)
__global__ __launch_bounds__(kBlock) void scan_tile_sums(const int32_t *in, int64_t n, long long *tile_sums) {
  __shared__ long long s_w[kWavesPerBlock];
  const int64_t base = (int64_t)blockIdx.x * kScanTile;
  long long s = 0;
  for (int k = 0; k < kScanTile / kBlock; ++k) {
    int64_t i = base + (int64_t)threadIdx.x * (kScanTile / kBlock) + k;
    if (i < n) s += in[i];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    long long t = 0;
    for (int w = 0; w < kWavesPerBlock; ++w) t += s_w[w];
    tile_sums[blockIdx.x] = t;
  }
}
__global__ void scan_tiles_serial(long long *tile_sums, int64_t ntiles, long long *total) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    long long run = 0;
    for (int64_t t = 0; t < ntiles; ++t) {
      long long v = tile_sums[t];
      tile_sums[t] = run;
      run += v;
    }
    *total = run;
  }
}
__global__ __launch_bounds__(kBlock) void scan_apply(int32_t *data, int64_t n, const long long *tile_offs) {
  __shared__ long long s_t[kBlock];
  constexpr int PER = kScanTile / kBlock;
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * PER;
  int v[PER];
  long long s = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    v[k] = (base + k < n) ? data[base + k] : 0;
    s += v[k];
  }
  s_t[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {   // serial scan of 256 lane totals: one-time setup kernel
    long long run = tile_offs[blockIdx.x];
    for (int t = 0; t < kBlock; ++t) {
      long long x = s_t[t];
      s_t[t] = run;
      run += x;
    }
  }
  __syncthreads();
  long long run = s_t[threadIdx.x];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    if (base + k < n) data[base + k] = (int32_t)run;
    run += v[k];
  }
}

}  // namespace khip

using namespace khip;

extern "C" int khip_gen_stencil(khip_ctx *ctx, int kind, int n1, int n2, int n3, int64_t row0, int64_t m,
                                int32_t **rowptr_dev, int32_t **col_dev, double **val_dev, int64_t *nnz_out) {
  KHIP_REQUIRE(ctx && rowptr_dev && col_dev && val_dev && nnz_out, "gen_stencil: null argument");
  KHIP_REQUIRE(kind >= 0 && kind <= 2 && n1 > 0 && n2 > 0 && n3 > 0, "gen_stencil: bad kind/dims");
  const int64_t n = (int64_t)n1 * n2 * n3;
  KHIP_REQUIRE(n < (1ll << 31), "gen_stencil: column index would overflow int32");
  KHIP_REQUIRE(row0 >= 0 && m >= 0 && row0 + m <= n, "gen_stencil: bad row range");
  KHIP_CHECK_HIP(hipSetDevice(ctx->device));
  int32_t *rp = nullptr, *cl = nullptr;
  double *vl = nullptr;
  KHIP_CHECK_HIP(hipMalloc(&rp, sizeof(int32_t) * (size_t)(m + 1)));
  const int64_t want = (m + kBlock - 1) / kBlock;
  const int grid = (int)(want < 8192 ? (want > 0 ? want : 1) : 8192);
  hipLaunchKernelGGL((stencil_kernel<false>), dim3(grid), dim3(kBlock), 0, ctx->stream, kind, n1, n2, n3, row0, m, rp,
                     (int32_t *)nullptr, (double *)nullptr);
  KHIP_CHECK_HIP(hipGetLastError());
  KHIP_CHECK_HIP(hipMemsetAsync(rp + m, 0, sizeof(int32_t), ctx->stream));
  // exclusive scan over m+1 entries (last input is 0 -> rowptr[m] = total)
  const int64_t cnt = m + 1;
  const int64_t ntiles = (cnt + kScanTile - 1) / kScanTile;
  long long *tiles = nullptr;
  KHIP_CHECK_HIP(hipMalloc(&tiles, sizeof(long long) * (size_t)(ntiles + 1)));
  hipLaunchKernelGGL(scan_tile_sums, dim3((unsigned)ntiles), dim3(kBlock), 0, ctx->stream, rp, cnt, tiles);
  hipLaunchKernelGGL(scan_tiles_serial, dim3(1), dim3(64), 0, ctx->stream, tiles, ntiles, tiles + ntiles);
  hipLaunchKernelGGL(scan_apply, dim3((unsigned)ntiles), dim3(kBlock), 0, ctx->stream, rp, cnt, tiles);
  KHIP_CHECK_HIP(hipGetLastError());
  long long total = 0;
  KHIP_CHECK_HIP(hipMemcpyAsync(&total, tiles + ntiles, sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  KHIP_CHECK_HIP(hipFree(tiles));
  if (total >= (1ll << 31) - 64) {
    (void)hipFree(rp);
    set_error("gen_stencil: shard nnz %lld does not fit int32 row pointers", total);
    return KHIP_ERR_INVALID;
  }
  KHIP_CHECK_HIP(hipMalloc(&cl, sizeof(int32_t) * (size_t)(total + kPad)));
  KHIP_CHECK_HIP(hipMalloc(&vl, sizeof(double) * (size_t)(total + kPad)));
  KHIP_CHECK_HIP(hipMemsetAsync(cl + total, 0, sizeof(int32_t) * kPad, ctx->stream));
  KHIP_CHECK_HIP(hipMemsetAsync(vl + total, 0, sizeof(double) * kPad, ctx->stream));
  hipLaunchKernelGGL((stencil_kernel<true>), dim3(grid), dim3(kBlock), 0, ctx->stream, kind, n1, n2, n3, row0, m, rp,
                     cl, vl);
  KHIP_CHECK_HIP(hipGetLastError());
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  *rowptr_dev = rp; *col_dev = cl; *val_dev = vl; *nnz_out = total;
  return KHIP_OK;
}


// ---------------------------------------------------------------- transpose (adjoint operator) ----
// A' as its own CSR handle, built on the device: column histogram -> exclusive scan -> scatter with an atomic
// cursor per column -> every row of A' sorted by its column index (= row of A).  The sort makes the result
// deterministic and gives each row of A' the entry order of a column of A, i.e. the summation order of
// `mul!(y, A', x)` on the CSC matrix the reference's users hold (the adjoint product that
// MINRES-QLP / LSQR / LSMR / BiLQ / QMR ... need on the device type).
namespace khip {

__global__ __launch_bounds__(kBlock) void tr_count_kernel(const int32_t *col, int64_t nnz, int32_t *count) {
  for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < nnz; q += (int64_t)gridDim.x * kBlock)
    atomicAdd(&count[col[q]], 1);
}

__global__ __launch_bounds__(kBlock) void tr_scatter_kernel(const int32_t *rowptr, const int32_t *col, const double *val,
                                                            int64_t m, int32_t *cursor, int32_t *colT, double *valT) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock)
    for (int32_t q = rowptr[i]; q < rowptr[i + 1]; ++q) {
      const int32_t pos = atomicAdd(&cursor[col[q]], 1);
      colT[pos] = (int32_t)i;
      valT[pos] = val[q];
    }
}

// insertion sort of each row of A' by column index (rows are short; a row of A' is a column of A)
__global__ __launch_bounds__(kBlock) void tr_sort_rows_kernel(const int32_t *rowptrT, int64_t n, int32_t *colT, double *valT) {
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n; j += (int64_t)gridDim.x * kBlock) {
    const int32_t s = rowptrT[j], e = rowptrT[j + 1];
    for (int32_t a = s + 1; a < e; ++a) {
      const int32_t c = colT[a];
      const double v = valT[a];
      int32_t b = a - 1;
      while (b >= s && colT[b] > c) { colT[b + 1] = colT[b]; valT[b + 1] = valT[b]; --b; }
      colT[b + 1] = c;
      valT[b + 1] = v;
    }
  }
}

// in-place exclusive scan of `cnt` int32 values (the generators' three-phase scan); *total = their sum
static int exclusive_scan_i32(khip_ctx *ctx, int32_t *data, int64_t cnt, long long *total) {
  const int64_t ntiles = (cnt + kScanTile - 1) / kScanTile;
  long long *tiles = nullptr;
  KHIP_CHECK_HIP(hipMalloc(&tiles, sizeof(long long) * (size_t)(ntiles + 1)));
  hipLaunchKernelGGL(scan_tile_sums, dim3((unsigned)ntiles), dim3(kBlock), 0, ctx->stream, data, cnt, tiles);
  hipLaunchKernelGGL(scan_tiles_serial, dim3(1), dim3(64), 0, ctx->stream, tiles, ntiles, tiles + ntiles);
  hipLaunchKernelGGL(scan_apply, dim3((unsigned)ntiles), dim3(kBlock), 0, ctx->stream, data, cnt, tiles);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(total, tiles + ntiles, sizeof(long long), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(tiles);
  if (e != hipSuccess) { set_error("scan: %s", hipGetErrorString(e)); return KHIP_ERR_HIP; }
  return KHIP_OK;
}

int csr_transpose(khip_ctx *ctx, const khip_csr *A, khip_csr *T) {
  const int64_t m = A->m, n = A->n, nnz = A->nnz;
  T->ctx = ctx; T->m = n; T->n = m; T->nnz = nnz;
  KHIP_CHECK_HIP(hipMalloc(&T->rowptr, sizeof(int32_t) * (size_t)(n + 1)));
  KHIP_CHECK_HIP(hipMalloc(&T->col, sizeof(int32_t) * (size_t)(nnz + kPad)));
  KHIP_CHECK_HIP(hipMalloc(&T->val, sizeof(double) * (size_t)(nnz + kPad)));
  KHIP_CHECK_HIP(hipMemsetAsync(T->rowptr, 0, sizeof(int32_t) * (size_t)(n + 1), ctx->stream));
  KHIP_CHECK_HIP(hipMemsetAsync(T->col + nnz, 0, sizeof(int32_t) * kPad, ctx->stream));
  KHIP_CHECK_HIP(hipMemsetAsync(T->val + nnz, 0, sizeof(double) * kPad, ctx->stream));
  if (nnz > 0) {
    int64_t want = (nnz + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(tr_count_kernel, dim3((unsigned)(want < 65536 ? want : 65536)), dim3(kBlock), 0, ctx->stream, A->col, nnz,
                       T->rowptr);
    KHIP_CHECK_HIP(hipGetLastError());
  }
  long long total = 0;
  KHIP_TRY(exclusive_scan_i32(ctx, T->rowptr, n + 1, &total));
  if (total != nnz) { set_error("csr_transpose: a column index lies outside [0, n)"); return KHIP_ERR_INVALID; }
  if (nnz > 0) {
    int32_t *cursor = nullptr;
    KHIP_CHECK_HIP(hipMalloc(&cursor, sizeof(int32_t) * (size_t)(n + 1)));
    KHIP_CHECK_HIP(hipMemcpyAsync(cursor, T->rowptr, sizeof(int32_t) * (size_t)(n + 1), hipMemcpyDeviceToDevice, ctx->stream));
    int64_t want = (m + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(tr_scatter_kernel, dim3((unsigned)(want < 65536 ? want : 65536)), dim3(kBlock), 0, ctx->stream, A->rowptr,
                       A->col, A->val, m, cursor, T->col, T->val);
    want = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(tr_sort_rows_kernel, dim3((unsigned)(want < 65536 ? want : 65536)), dim3(kBlock), 0, ctx->stream,
                       T->rowptr, n, T->col, T->val);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(cursor);
    if (e != hipSuccess) { set_error("csr_transpose: %s", hipGetErrorString(e)); return KHIP_ERR_HIP; }
  }
  return csr_finalize(ctx, T);
}

// ---------------------------------------------------------------- SpMM window metadata ----------
// One workgroup per group of RPB consecutive rows: the distinct columns of the group through an LDS hash set
// (identity hash + linear probing, so neighbouring columns stay neighbours), numbered in slot order by a prefix sum
// => the list of a banded operator comes out in (nearly) ascending column order and the kernel's copy of the panel rows
// runs over contiguous memory.  FILL = false counts (cnt[g] = -1 when the group does not fit), FILL = true writes the
// list at base[g] and the 16-bit slot of every nonzero.
constexpr int kWinEmpty = -1;

template <int L, bool FILL>
__global__ __launch_bounds__(kBlock) void spmm_window_build_kernel(const int32_t *rowptr, const int32_t *col, int64_t m,
                                                                    int32_t *cnt, int stride, int32_t *list, uint16_t *slot,
                                                                    unsigned long long *stat) {
  using W = WinShape<L>;
  constexpr int PER = W::HT / kBlock;            // consecutive hash slots per thread in the numbering pass
  __shared__ int keys[W::HT];
  __shared__ unsigned short ids[W::HT];
  __shared__ int part[kBlock];
  __shared__ int count, overflow;
  const int tid = threadIdx.x, sub = tid / L, c = tid % L;
  const int64_t row = (int64_t)blockIdx.x * W::RPB + sub;
  if (FILL && cnt[blockIdx.x] < 0) return;       // direct group: its list stays zero

  for (int h = tid; h < W::HT; h += kBlock) keys[h] = kWinEmpty;
  if (tid == 0) { count = 0; overflow = 0; }
  __syncthreads();
  int64_t s = 0, e = 0;
  if (row < m) { s = rowptr[row]; e = rowptr[row + 1]; }
  for (int64_t q = s + c; q < e; q += L) {
    if (__atomic_load_n(&overflow, __ATOMIC_RELAXED)) break;
    const int key = col[q];
    unsigned h = (unsigned)key & (W::HT - 1);
    for (;;) {
      const int k = __atomic_load_n(&keys[h], __ATOMIC_RELAXED);
      if (k == key) break;
      if (k == kWinEmpty) {
        const int old = atomicCAS(&keys[h], kWinEmpty, key);
        if (old == kWinEmpty) {
          if (atomicAdd(&count, 1) >= W::CAP) __atomic_store_n(&overflow, 1, __ATOMIC_RELAXED);
          break;
        }
        if (old == key) break;
      }
      h = (h + 1) & (W::HT - 1);
    }
  }
  __syncthreads();
  if (!FILL) {
    if (tid == 0) {
      const int64_t r0 = (int64_t)blockIdx.x * W::RPB, r1 = r0 + W::RPB < m ? r0 + W::RPB : m;
      const bool direct = overflow != 0 || rowptr[r1] - rowptr[r0] > W::ENTRIES;
      cnt[blockIdx.x] = direct ? -1 : count;
      if (direct) atomicAdd(&stat[0], 1ull);
      else { atomicMax(&stat[1], (unsigned long long)count); atomicAdd(&stat[2], (unsigned long long)count); }
    }
    return;
  }
  // numbering in slot order: thread t owns slots [t PER, (t + 1) PER)
  int mine = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) mine += keys[tid * PER + j] != kWinEmpty;
  part[tid] = mine;
  __syncthreads();
  int before = 0;
  for (int t = 0; t < tid; ++t) before += part[t];
  int32_t *lst = list + (int64_t)blockIdx.x * stride;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int k = keys[tid * PER + j];
    if (k != kWinEmpty) {
      ids[tid * PER + j] = (unsigned short)before;
      lst[before] = k;
      ++before;
    }
  }
  __syncthreads();
  for (int64_t q = s + c; q < e; q += L) {
    const int key = col[q];
    unsigned h = (unsigned)key & (W::HT - 1);
    while (keys[h] != key) h = (h + 1) & (W::HT - 1);
    slot[q] = ids[h];
  }
}

__global__ __launch_bounds__(kBlock) void window_flag_kernel(const int32_t *cnt, int64_t groups, int32_t *flag) {
  const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (g < groups) flag[g] = cnt[g] < 0 ? 1 : 0;
}

void csr_free_window(khip_csr *A) {
  (void)hipFree(A->win_list); (void)hipFree(A->win_slot); (void)hipFree(A->win_flag);
  A->win_list = nullptr; A->win_slot = nullptr; A->win_flag = nullptr;
  A->win_L = 0;
}

template <int L>
static int window_build_t(khip_ctx *ctx, khip_csr *A) {
  using W = WinShape<L>;
  const int64_t groups = (A->m + W::RPB - 1) / W::RPB;
  int32_t *cnt = nullptr;
  unsigned long long *stat = nullptr;            // [0] groups on the direct path, [1] largest list, [2] sum of the lists
  struct Scratch { int32_t *&c; unsigned long long *&s; ~Scratch() { (void)hipFree(c); (void)hipFree(s); } } scratch{cnt, stat};
  KHIP_CHECK_HIP(hipMalloc(&cnt, sizeof(int32_t) * (size_t)groups));
  KHIP_CHECK_HIP(hipMalloc(&stat, 3 * sizeof(unsigned long long)));
  KHIP_CHECK_HIP(hipMemsetAsync(stat, 0, 3 * sizeof(unsigned long long), ctx->stream));
  hipLaunchKernelGGL((spmm_window_build_kernel<L, false>), dim3((unsigned)groups), dim3(kBlock), 0, ctx->stream, A->rowptr, A->col,
                     A->m, cnt, 0, nullptr, nullptr, stat);
  KHIP_CHECK_HIP(hipGetLastError());
  unsigned long long st[3] = {0, 0, 0};
  KHIP_CHECK_HIP(hipMemcpyAsync(st, stat, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  // worth it only when most groups fit and the rows really share panel rows (>= 1.5 references per distinct column)
  if (2 * (long long)st[0] > groups || st[2] == 0 || 2 * (unsigned long long)A->nnz < 3 * st[2]) { A->win_L = -L; return KHIP_OK; }
  const int stride = W::STRIDE;
  KHIP_CHECK_HIP(hipMalloc(&A->win_list, sizeof(int32_t) * (size_t)groups * stride));
  KHIP_CHECK_HIP(hipMalloc(&A->win_slot, sizeof(uint16_t) * (size_t)(A->nnz + kPad)));
  KHIP_CHECK_HIP(hipMalloc(&A->win_flag, sizeof(int32_t) * (size_t)groups));
  KHIP_CHECK_HIP(hipMemsetAsync(A->win_list, 0, sizeof(int32_t) * (size_t)groups * stride, ctx->stream));       // padding: column 0
  KHIP_CHECK_HIP(hipMemsetAsync(A->win_slot, 0, sizeof(uint16_t) * (size_t)(A->nnz + kPad), ctx->stream));
  hipLaunchKernelGGL(window_flag_kernel, dim3((unsigned)((groups + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, cnt, groups,
                     A->win_flag);
  hipLaunchKernelGGL((spmm_window_build_kernel<L, true>), dim3((unsigned)groups), dim3(kBlock), 0, ctx->stream, A->rowptr, A->col,
                     A->m, cnt, stride, A->win_list, A->win_slot, nullptr);
  KHIP_CHECK_HIP(hipGetLastError());
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  A->win_L = L;
  return KHIP_OK;
}

// Builds (or rebuilds for another lane count) the window metadata of A.  On return A->win_L == L (usable) or -L (the
// operator lacks the locality, or memory is short: the caller keeps the direct-gather kernel and does not ask again).
int spmm_window_build(khip_ctx *ctx, khip_csr *A, int L) {
  csr_free_window(A);
  int rc;
  switch (L) {
    case 2: rc = window_build_t<2>(ctx, A); break;
    case 4: rc = window_build_t<4>(ctx, A); break;
    case 8: rc = window_build_t<8>(ctx, A); break;
    case 16: rc = window_build_t<16>(ctx, A); break;
    default: rc = window_build_t<32>(ctx, A); break;
  }
  if (rc != KHIP_OK || A->win_L != L) {          // not usable: release whatever was allocated, remember the answer
    (void)hipGetLastError();
    csr_free_window(A);
    A->win_L = -L;
  }
  return KHIP_OK;
}

}  // namespace khip
