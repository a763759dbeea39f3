// spmv.hip -- CSR SpMV for gfx950.
//
// Replaces kmul!(y, A, x) = mul!(y, A, x) (src/krylov_utils.jl:305).  HBM-bound: algorithmic
// bytes = 12 nnz + 4 (m+1) + 8 n + 8 m, AI = 0.135 flop/B on the 7-point
// Poisson operator.
//
// Data layout in HBM: val f64[nnz], col i32[nnz], rowptr i32[m+1] (0-based), x f64[n], y f64[m];
// val/col are padded by 8 zeroed entries so that 16-byte lane loads may overrun the last row.
//
// Launch shape: LOOP-FREE by default -- one tile of rows per workgroup, ~10^5..10^6 workgroups --
// because on MI355X streaming kernels with short-lived workgroups sustain 30-50 % more HBM bandwidth
// than persistent grid-stride loops (profiles/r01_membench*.log).  `spmv_persist` restores the
// persistent form (contiguous row ranges per workgroup, optional XCD-contiguous remap) for A/B runs.
//
// Kernels (all produce y BIT-IDENTICAL to the serial CPU loop except spmv_vector):
//  * spmv_ordered<L>: L = 4..64 lanes cooperate on a row.  Lane k loads entry k of the row (the
//    val/col streams of consecutive rows are contiguous, so a wave reads one contiguous span),
//    multiplies by the gathered x, and the L products are folded IN STORED ORDER with L broadcast
//    shuffles: one rounded multiply and one rounded add per entry, exactly the arithmetic of
//    SparseArrays.mul! / the oracle.  No LDS, no barrier, ~30 VGPRs -> 8 waves per SIMD.  Several
//    rows per lane group are in flight at once (independent loads issued up front).
//  * spmv_stream: the LDS-staged form of the north star: a workgroup reads the val/col streams of
//    256 rows fully coalesced, stages val*x[col] in LDS (14 KB), then one lane per row sums its
//    segment in stored order.  Row pointers live in registers (no LDS, no extra barrier).
//  * spmv_vector<L>: L lanes stride over a long row with FMA accumulation + shuffle tree (rows with
//    hundreds of entries); not bit-identical to the serial loop.
//  * optional fused dot (x . y) for CG's pAp (src/cg.jl:196-197): the lane that owns a row multiplies
//    its y value by x[row]; one partial per wave + the tiny finish kernel (device_reduce.hpp).
#include "spmv_common.hpp"

// cache-policy bits of the staged kernel's val/col window loads when spmv_nt = 1 (gfx940+: 1 = sc0, 2 = nt, 16 = sc1);
// every combination measured slower than the default policy (profiles/r01h_stage_aux.log)
#ifndef KHIP_STAGE_AUX
#define KHIP_STAGE_AUX 2
#endif

namespace khip {

// y[row] = v under the context's store policy: plain, non-temporal (spmv_nty = 1) or write-through (spmv_nty = 2: an sc1 store
// leaves no line in the XCD's L2, MI355X_MICROARCH.md, so y does not compete with the x lines the gathers live on)
__device__ __forceinline__ void store_y(const SpmvArgs &a, int64_t row, double v) {
  if (a.nt_y == 2) {
    const double *p = a.y + row;
    asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
  } else if (a.nt_y) __builtin_nontemporal_store(v, a.y + row);
  else a.y[row] = v;
}

// Logical tile id of workgroup b (of G).  The dispatcher places workgroup b on XCD b % 8 (observed,
// MI355X_MICROARCH.md), each XCD has its own 4 MiB L2.  xcd_run = R > 0 hands every XCD runs of R
// CONSECUTIVE tiles inside each window of 8R workgroups, so the x entries shared by neighbouring rows
// (a stencil's +-n1 couplings) are fetched into ONE L2 instead of three, while the chip-wide access
// front stays compact (window = 8R tiles).  xcd_run = -1: one contiguous eighth per XCD.
//
// xcd_run = -2: PLANE SWEEP.  A 3-D stencil couples row i to i +- S tiles (the neighbouring grid planes); with
// the natural order those x lines are long gone from the 4 MiB L2 when the sweep reaches the next plane
// (S = 1024 tiles = 27 MB of matrix stream at 512^3), so x is fetched ~4x.  Here every XCD takes a column of W
// consecutive tiles and walks it THROUGH all planes before moving on: tile order (t, k, w) -> k*S + t*W + w,
// XCD p owning the columns t = p (mod 8).  The plane-to-plane reuse distance shrinks from S to W tiles.
// Any (S, W) gives a permutation of the tiles, so results never depend on it.
__device__ __forceinline__ int chunk_id(int b, int G, int xcd_run, int S = 0, int W = 0) {
  if (xcd_run == -2) {
    if (S <= 0 || W <= 0 || S % (8 * W) != 0) return b;
    const int K = G / S;                       // full planes
    if (b >= K * S) return b;
    const int p = b & 7, l = b >> 3;
    const int per = K * W;
    const int tt = l / per, rem = l - tt * per;
    const int k = rem / W, w = rem - k * W;
    return k * S + (tt * 8 + p) * W + w;
  }
  if (xcd_run > 0) {
    const int win = 8 * xcd_run;
    const int full = (G / win) * win;
    if (b < full) return (b / win) * win + (b & 7) * xcd_run + ((b % win) >> 3);
    return b;
  }
  if (xcd_run < 0 && (G & 7) == 0) return (b & 7) * (G >> 3) + (b >> 3);
  return b;
}

// ---------------------------------------------------------------- stream (LDS-staged) ----
template <int ROWS, int VEC, bool NT, bool DOT, bool COMP, bool DIST>
__global__ __launch_bounds__(kBlock) void spmv_stream_kernel(SpmvArgs a, RedArgs ra) {
  if (seq_skip(a.stop_seq, a.seq)) return;
  static_assert(ROWS <= kBlock, "one lane per row");
  constexpr int CAP = 2048;   // products staged per pass (16 KB)
  __shared__ double s_prod[CAP + 4];
  const int tid = threadIdx.x;
  const int G = gridDim.x;
  const int64_t nrows = a.row_hi - a.row_lo;
  const int64_t nrb = (nrows + ROWS - 1) / ROWS;
  const int cid = chunk_id(blockIdx.x, G, a.xcd_remap);
  const int64_t rb_begin = nrb * cid / G;
  const int64_t rb_end = nrb * (cid + 1) / G;
  dd dacc[2];                                // [0] = w . y ; [1] = y . y or w . w (only when a.dot_sq)
  dacc[0] = dd{0.0, 0.0};
  dacc[1] = dd{0.0, 0.0};

  for (int64_t rb = rb_begin; rb < rb_end; ++rb) {
    const int64_t r0 = a.row_lo + rb * ROWS;
    const int nr = (int)((a.row_hi - r0) < ROWS ? (a.row_hi - r0) : ROWS);
    // block-uniform range (scalar loads) + this lane's row, all issued before anything waits
    // With a block-pointer table (every 256th row pointer, 2 MB at 512^3 => L2-resident) the first,
    // chain-starting lookup is an L2 hit instead of an HBM miss on a fresh rowptr line.
    int64_t s, e;
    if (ROWS == 256 && a.blockptr != nullptr) {
      const int64_t bi = (r0 >> 8);
      s = a.blockptr[bi];
      e = a.blockptr[bi + 1];
    } else {
      s = a.rowptr[r0];
      e = a.rowptr[r0 + nr];
    }
    const int my_a = (tid < nr) ? a.rowptr[r0 + tid] : 0;
    const int my_b = (tid < nr) ? a.rowptr[r0 + tid + 1] : 0;
    double acc = 0.0;
    for (int64_t c0 = s & ~(int64_t)(VEC - 1); c0 < e; c0 += CAP) {
      const double *vbase = a.val + c0;
      const int32_t *cbase = a.col + c0;
      const int len = (int)((e - c0) < CAP ? (e - c0) : CAP);
      if (VEC == 1) {
        // CAP = 8 * kBlock: all (<= 8) val/col loads of this lane are issued first, then all x
        // gathers, then the LDS writes -- two memory latencies per row block instead of four
        constexpr int IT = CAP / kBlock;
        double v[IT];
        int32_t c[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
          const int o = tid + it * kBlock;
          const bool ok = o < len;
          v[it] = ld<NT>(vbase + (ok ? o : 0));
          c[it] = ld<NT>(cbase + (ok ? o : 0));
        }
        double xg[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
          // a.fake_gather (tuning experiment only): replace the scattered gather by a coalesced read
          const int32_t ci = a.fake_gather ? (int32_t)((tid + it * kBlock) & 2047) + (c[it] & 0) : c[it];
          xg[it] = gather_x<DIST>(a, ci);
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
          const int o = tid + it * kBlock;
          if (o < len) s_prod[o] = v[it] * xg[it];
        }
      } else {
#pragma unroll 4
        for (int o = VEC * tid; o < len; o += VEC * kBlock) {
          dbl2 v = ld<NT>(reinterpret_cast<const dbl2 *>(vbase + o));
          int2v c = ld<NT>(reinterpret_cast<const int2v *>(cbase + o));
          double p0 = v.x * gather_x<DIST>(a, c.x);
          double p1 = v.y * gather_x<DIST>(a, c.y);
          s_prod[o] = p0;
          s_prod[o + 1] = p1;
        }
      }
      __syncthreads();
      if (tid < nr) {
        const int rel_a = (int)(my_a - c0), rel_b = (int)(my_b - c0);
        const int lo = rel_a > 0 ? rel_a : 0;
        const int hi = rel_b < len ? rel_b : len;
        for (int k = lo; k < hi; ++k) acc = acc + s_prod[k];
      }
      __syncthreads();
    }
    if (tid < nr) {
      store_y(a, r0 + tid, acc);
      if (DOT) {
        const double wv = a.dotw[r0 + tid];
        acc_prod<COMP>(dacc[0], wv, acc);
        if (a.dot_sq == 1) acc_prod<COMP>(dacc[1], acc, acc);        // y . y
        else if (a.dot_sq == 2) acc_prod<COMP>(dacc[1], wv, wv);     // w . w
      }
    }
  }
  if (DOT) {
    if (a.dot_sq) wave_publish<2>(dacc, ra);
    else wave_publish<1>(reinterpret_cast<dd (&)[1]>(dacc), ra);
  }
}

// ---------------------------------------------------------------- staged rows ------------
// The scattered x gather is what limits the kernels above: with val*x[col] formed in nnz order a
// wave's 64 gather addresses fall in ~14 different cache lines, and the texture-address path, not
// HBM, sets the pace (measured: replacing the gather by a coalesced L1-resident read takes the
// stream kernel from 2.79 ms to 1.96 ms at 512^3, profiles/r01_sweep2d.log).  This kernel keeps the
// val/col streams fully coalesced (HBM -> registers -> LDS, 24 KB per 256 rows) and then lets ONE
// LANE PER ROW walk its row out of LDS: at step k the 64 lanes of a wave gather x[col(row, k)] for 64
// CONSECUTIVE rows -- for banded / stencil operators those are consecutive addresses, i.e. the
// gather becomes a coalesced load.  Each lane accumulates its row in stored order with a rounded
// multiply and a rounded add per entry: bit-identical to the serial CPU loop for any matrix.
// LDS reads are conflict-free for odd row lengths (stride 7 doubles -> distinct bank pairs).
template <int ROWS, bool NT, bool DOT, bool COMP, bool DIST>
__global__ __launch_bounds__(kBlock) void spmv_stage_kernel(SpmvArgs a, RedArgs ra) {
  if (seq_skip(a.stop_seq, a.seq)) return;
  static_assert(ROWS <= kBlock, "one lane per row");
  // window of staged entries: up to 8 per lane, moved as 16-byte vectors.  Its size is a launch parameter
  // (multiple of 4, <= 2048) sized to the widest row block of the matrix: LDS per workgroup is 12 * CAP
  // bytes, so a 7-entries-per-row operator keeps 7 instead of 6 workgroups resident per CU.
  const int CAP = a.stage_cap;
  constexpr int UK = 8;                      // row entries whose gathers are in flight together
  extern __shared__ __attribute__((aligned(16))) unsigned char s_stage[];
  double *s_val = reinterpret_cast<double *>(s_stage);
  int32_t *s_col = reinterpret_cast<int32_t *>(s_stage + (size_t)CAP * sizeof(double));
  const int tid = threadIdx.x;
  const int64_t nrows = a.row_hi - a.row_lo - a.hole_len;
  const int64_t nrb = (nrows + ROWS - 1) / ROWS;
  const int tpb = a.tiles_per_block > 0 ? a.tiles_per_block : 1;
  const int cid = chunk_id(blockIdx.x, gridDim.x, a.xcd_remap, a.sweep_s, a.sweep_w);
  const int64_t rb_begin = (int64_t)cid * tpb;
  const int64_t rb_end = (rb_begin + tpb < nrb) ? rb_begin + tpb : nrb;
  dd dacc[2];                                // [0] = w . y ; [1] = y . y (only when a.dot_sq)
  dacc[0] = dd{0.0, 0.0};
  dacc[1] = dd{0.0, 0.0};

  for (int64_t rb = rb_begin; rb < rb_end; ++rb) {
    int64_t r0 = a.row_lo + rb * ROWS;
    if (r0 >= a.hole_lo) r0 += a.hole_len;   // the second range of a two-range launch
    const int nr = (int)((a.row_hi - r0) < ROWS ? (a.row_hi - r0) : ROWS);
    const int64_t s = a.rowptr[r0], e = a.rowptr[r0 + nr];
    const int my_a = (tid < nr) ? a.rowptr[r0 + tid] : 0;
    const int my_b = (tid < nr) ? a.rowptr[r0 + tid + 1] : 0;
    double acc = 0.0, wv = 0.0;
    if (DOT && a.dot_early && tid < nr) wv = a.dotw[r0 + tid];       // in flight beside the window, not behind the row walk
    // windows start on a multiple of 4 entries so that every lane moves aligned 16-byte vectors:
    // 4 val loads (2 entries each) + 2 col loads (4 entries each) per lane and window -- the
    // vector-memory instruction count, not HBM, is what bounds this kernel
    for (int64_t c0 = s & ~(int64_t)3; c0 < e; c0 += CAP) {
      const int lim = (int)((e - c0) < (int64_t)CAP ? (e - c0) : (int64_t)CAP);   // entries of this window
      // The window is read through buffer descriptors whose extent is the row block's own entries
      // (rounded up to the 16-byte vector): lanes past it get zeros WITHOUT a memory request.  An
      // over-read into the neighbouring block's entries would be fetched from HBM twice, since that block
      // usually runs on another XCD (other L2); predicating with branches instead costs ~25 VGPRs.
      const int lim4 = (lim + 3) & ~3;
      const __amdgpu_buffer_rsrc_t rv =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(a.val + c0), 0, lim4 * 8, kBufRsrcWord3);
      const __amdgpu_buffer_rsrc_t rc =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(a.col + c0), 0, lim4 * 4, kBufRsrcWord3);
      u32x4 v[4], c[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int o = q * (4 * kBlock) + 4 * tid;
        v[2 * q] = __builtin_amdgcn_raw_buffer_load_b128(rv, o * 8, 0, NT ? KHIP_STAGE_AUX : 0);
        v[2 * q + 1] = __builtin_amdgcn_raw_buffer_load_b128(rv, o * 8 + 16, 0, NT ? KHIP_STAGE_AUX : 0);
        c[q] = __builtin_amdgcn_raw_buffer_load_b128(rc, o * 4, 0, NT ? KHIP_STAGE_AUX : 0);
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int o = q * (4 * kBlock) + 4 * tid;
        if (o < CAP) {
          *reinterpret_cast<u32x4 *>(s_val + o) = v[2 * q];
          *reinterpret_cast<u32x4 *>(s_val + o + 2) = v[2 * q + 1];
          *reinterpret_cast<u32x4 *>(s_col + o) = c[q];
        }
      }
      __syncthreads();
      if (tid < nr) {
        const int rel_a = (int)(my_a - c0), rel_b = (int)(my_b - c0);
        const int lo = rel_a > 0 ? rel_a : 0;
        const int hi = rel_b < lim ? rel_b : lim;
        for (int k0 = lo; k0 < hi; k0 += UK) {
          int32_t cc[UK];
          double vv[UK], xx[UK];
#pragma unroll
          for (int u = 0; u < UK; ++u) {
            const int j = (k0 + u < hi) ? k0 + u : k0;
            cc[u] = s_col[j];
            vv[u] = s_val[j];
          }
#pragma unroll
          for (int u = 0; u < UK; ++u) {
            xx[u] = 0.0;
            if (k0 + u < hi) xx[u] = gather_x<DIST>(a, cc[u]);      // skipped when no lane of the wave needs it
          }
#pragma unroll
          for (int u = 0; u < UK; ++u) {
            if (k0 + u < hi) {
              const double prod = vv[u] * xx[u];
              acc = acc + prod;
            }
          }
        }
      }
      __syncthreads();
    }
    if (tid < nr) {
      store_y(a, r0 + tid, acc);
      if (DOT) {
        if (!a.dot_early) wv = a.dotw[r0 + tid];
        acc_prod<COMP>(dacc[0], wv, acc);
        if (a.dot_sq == 1) acc_prod<COMP>(dacc[1], acc, acc);        // y . y
        else if (a.dot_sq == 2) acc_prod<COMP>(dacc[1], wv, wv);     // w . w (single-reduction CG: r.r beside r.(A r))
      }
    }
  }
  if (DOT) {
    // the window is free: every wave has passed the loop's last barrier (a tile has at least one window when nnz > 0;
    // a.blk_pub is off otherwise)
    if (a.blk_pub) {
      dd *s_red = reinterpret_cast<dd *>(s_stage);
      if (a.dot_sq) block_publish<2>(dacc, ra, s_red);
      else block_publish<1>(reinterpret_cast<dd (&)[1]>(dacc), ra, s_red);
    } else if (a.dot_sq) wave_publish<2>(dacc, ra);
    else wave_publish<1>(reinterpret_cast<dd (&)[1]>(dacc), ra);
  }
}

// ---------------------------------------------------------------- staged rows, coded columns ----
// Same kernel with the column stream re-encoded (colcode.hip): banded / stencil operators use few distinct
// DIAGONALS d = column - row (7 for get_div_grad, whatever the grid size; two more on a rank's [owned | ghost]
// slab), so a handle whose entries lie on at most 256 (2048) diagonals keeps ONE BYTE (two) per entry -- the index
// of d in a sorted table that every workgroup holds in LDS -- next to the CSR arrays, and this kernel streams
// 9 (10) bytes per entry instead of 12.  col = row + tab[code] is exact integer arithmetic and the row is walked in
// stored order with the same rounded multiply and rounded add: y is BIT-IDENTICAL to spmv_stage_kernel and to the
// serial CPU loop.  The boundary still takes (and the handle still owns) plain CSR; this is how the staged kernel
// reads it.  Reported bandwidths keep the CSR byte formula of SURVEY 8(d) as "algorithmic" and state the bytes
// actually moved beside it (khip_spmv_bytes_stored).
template <typename CODE> struct code_load;
template <> struct code_load<uint8_t> {    // 4 codes = one dword
  typedef unsigned int vec;
  template <int AUX = 0>
  static __device__ __forceinline__ vec ld(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, AUX);
  }
};
template <> struct code_load<uint16_t> {   // 4 codes = two dwords
  typedef unsigned int vec __attribute__((ext_vector_type(2)));
  template <int AUX = 0>
  static __device__ __forceinline__ vec ld(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, AUX);
  }
};

template <typename CODE, bool DOT, bool COMP, bool DIST>
__global__ __launch_bounds__(kBlock) void spmv_code_kernel(SpmvArgs a, RedArgs ra) {
  if (seq_skip(a.stop_seq, a.seq)) return;
  typedef typename code_load<CODE>::vec cvec;
  const int ROWS = a.stage_rows;             // rows per block (<= kBlock: one lane per row)
  const int CAP = a.stage_cap;               // window in entries (multiple of 4, <= 2048)
  constexpr int UK = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char s_stage[];
  double *s_val = reinterpret_cast<double *>(s_stage);
  CODE *s_code = reinterpret_cast<CODE *>(s_stage + (size_t)CAP * sizeof(double));
  int32_t *s_tab = reinterpret_cast<int32_t *>(s_stage + (size_t)CAP * (sizeof(double) + sizeof(CODE)));
  const int tid = threadIdx.x;
  const CODE *code = reinterpret_cast<const CODE *>(a.code);
  const int64_t nrows = a.row_hi - a.row_lo - a.hole_len;
  const int64_t nrb = (nrows + ROWS - 1) / ROWS;
  const int tpb = a.tiles_per_block > 0 ? a.tiles_per_block : 1;
  const int cid = chunk_id(blockIdx.x, gridDim.x, a.xcd_remap, a.sweep_s, a.sweep_w);
  const int64_t rb_begin = (int64_t)cid * tpb;
  const int64_t rb_end = (rb_begin + tpb < nrb) ? rb_begin + tpb : nrb;
  dd dacc[2];
  dacc[0] = dd{0.0, 0.0};
  dacc[1] = dd{0.0, 0.0};
  // the diagonal table: L2-resident, issued beside the first window's loads, visible after the first barrier
  for (int i = tid; i < a.code_T; i += kBlock) s_tab[i] = a.code_tab[i];

  for (int64_t rb = rb_begin; rb < rb_end; ++rb) {
    int64_t r0 = a.row_lo + rb * ROWS;
    if (r0 >= a.hole_lo) r0 += a.hole_len;   // the second range of a two-range launch
    const int nr = (int)((a.row_hi - r0) < ROWS ? (a.row_hi - r0) : ROWS);
    const int64_t s = a.rowptr[r0], e = a.rowptr[r0 + nr];
    const int my_a = (tid < nr) ? a.rowptr[r0 + tid] : 0;
    const int my_b = (tid < nr) ? a.rowptr[r0 + tid + 1] : 0;
    const int32_t row = (int32_t)(r0 + tid);
    double acc = 0.0, wv = 0.0;
    if (DOT && a.dot_early && tid < nr) wv = a.dotw[r0 + tid];       // in flight beside the window, not behind the row walk
    for (int64_t c0 = s & ~(int64_t)3; c0 < e; c0 += CAP) {
      const int lim = (int)((e - c0) < (int64_t)CAP ? (e - c0) : (int64_t)CAP);
      const int lim4 = (lim + 3) & ~3;
      // descriptors whose extent is the row block's own entries: lanes past it cost no memory request
      const __amdgpu_buffer_rsrc_t rv =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(a.val + c0), 0, lim4 * 8, kBufRsrcWord3);
      const __amdgpu_buffer_rsrc_t rc =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<CODE *>(code + c0), 0, lim4 * (int)sizeof(CODE), kBufRsrcWord3);
      u32x4 v[4];
      cvec c[2];
      // cache policy of the matrix stream (a.stream_nt; experiment of round 4): 0 default, 1 nt, 2 sc0, 3 sc0 nt, 4 sc1, 5 sc1 nt
#define KHIP_WIN_LOADS(AUX)                                                                   \
      _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                        \
        const int o = q * (4 * kBlock) + 4 * tid;                                            \
        v[2 * q] = __builtin_amdgcn_raw_buffer_load_b128(rv, o * 8, 0, AUX);                 \
        v[2 * q + 1] = __builtin_amdgcn_raw_buffer_load_b128(rv, o * 8 + 16, 0, AUX);        \
        c[q] = code_load<CODE>::template ld<AUX>(rc, o * (int)sizeof(CODE));                 \
      }
      switch (a.stream_nt) {
        case 1: KHIP_WIN_LOADS(2) break;
        case 2: KHIP_WIN_LOADS(1) break;
        case 3: KHIP_WIN_LOADS(3) break;
        case 4: KHIP_WIN_LOADS(16) break;
        case 5: KHIP_WIN_LOADS(18) break;
        default: KHIP_WIN_LOADS(0) break;
      }
#undef KHIP_WIN_LOADS
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int o = q * (4 * kBlock) + 4 * tid;
        if (o < CAP) {
          *reinterpret_cast<u32x4 *>(s_val + o) = v[2 * q];
          *reinterpret_cast<u32x4 *>(s_val + o + 2) = v[2 * q + 1];
          *reinterpret_cast<cvec *>(s_code + o) = c[q];
        }
      }
      __syncthreads();
      if (tid < nr) {
        const int rel_a = (int)(my_a - c0), rel_b = (int)(my_b - c0);
        const int lo = rel_a > 0 ? rel_a : 0;
        const int hi = rel_b < lim ? rel_b : lim;
        for (int k0 = lo; k0 < hi; k0 += UK) {
          int32_t cc[UK];
          double vv[UK], xx[UK];
#pragma unroll
          for (int u = 0; u < UK; ++u) {
            const int j = (k0 + u < hi) ? k0 + u : k0;
            cc[u] = (int32_t)s_code[j];
            vv[u] = s_val[j];
          }
#pragma unroll
          for (int u = 0; u < UK; ++u) cc[u] = row + s_tab[cc[u]];
#pragma unroll
          for (int u = 0; u < UK; ++u) {
            xx[u] = 0.0;
            if (k0 + u < hi) xx[u] = gather_x<DIST>(a, cc[u]);
          }
#pragma unroll
          for (int u = 0; u < UK; ++u) {
            if (k0 + u < hi) {
              const double prod = vv[u] * xx[u];
              acc = acc + prod;
            }
          }
        }
      }
      __syncthreads();
    }
    if (tid < nr) {
      store_y(a, r0 + tid, acc);
      if (DOT) {
        if (!a.dot_early) wv = a.dotw[r0 + tid];
        acc_prod<COMP>(dacc[0], wv, acc);
        if (a.dot_sq == 1) acc_prod<COMP>(dacc[1], acc, acc);
        else if (a.dot_sq == 2) acc_prod<COMP>(dacc[1], wv, wv);
      }
    }
  }
  if (DOT) {
    // the window is free: every wave has passed the loop's last barrier (a tile has at least one window when nnz > 0;
    // a.blk_pub is off otherwise)
    if (a.blk_pub) {
      dd *s_red = reinterpret_cast<dd *>(s_stage);
      if (a.dot_sq) block_publish<2>(dacc, ra, s_red);
      else block_publish<1>(reinterpret_cast<dd (&)[1]>(dacc), ra, s_red);
    } else if (a.dot_sq) wave_publish<2>(dacc, ra);
    else wave_publish<1>(reinterpret_cast<dd (&)[1]>(dacc), ra);
  }
}

// ---------------------------------------------------------------- coded columns, sliced rows ----
// The coded operator in its sliced form (colcode.hip csr_build_sell): lane i owns row r0 + i as in spmv_code_kernel -- same row
// blocks, same workgroups, same partial slots, so the fused dot's partials are the SAME numbers -- but it loads the entries of its
// row itself: one 64-bit word of eight codes and up to eight values per step, each a coalesced 8-byte load (entry k of the 64 rows
// of a slice is 512 contiguous bytes).  No LDS window, no barrier and no row pointer in the row walk: the only dependent hop is
// code word -> table lookup (LDS) -> x gather.  Products in stored order, one rounded multiply and one rounded add per entry:
// y is bit-identical to every other kernel.
// C4: narrow codes -- eight 4-bit codes per row in one 32-bit word indexed by the row, the slices hold values only
// PAIR: rows of at most 8 entries, the code word and the values of a row in 16-byte pairs (element e of lane l = words 2e, 2e + 1)
// PAIRG: the general pair layout -- an even number of head words (codes or columns), then an even number of value words; any row length
template <bool DOT, bool COMP, bool DIST, bool NTM, bool COLS32, bool C4, bool PAIR, bool PAIRG>
__global__ __launch_bounds__(kBlock) void spmv_sell_kernel(SpmvArgs a, RedArgs ra) {
  if (seq_skip(a.stop_seq, a.seq)) return;
  const int ROWS = a.stage_rows;
  extern __shared__ __attribute__((aligned(16))) unsigned char s_stage[];
  // [block_publish scratch][diagonal table, 256 entries]: the scratch must not overlay the table, no barrier separates the row walks
  // of different waves from the publish
  const size_t pub = (DOT && a.blk_pub) ? sizeof(dd) * (size_t)kBlock * (a.dot_sq ? 2u : 1u) : 0u;
  int32_t *s_tab = reinterpret_cast<int32_t *>(s_stage + pub);
  const int tid = threadIdx.x;
  const int64_t nrows = a.row_hi - a.row_lo - a.hole_len;
  const int64_t nrb = (nrows + ROWS - 1) / ROWS;
  const int tpb = a.tiles_per_block > 0 ? a.tiles_per_block : 1;
  const int cid = chunk_id(blockIdx.x, gridDim.x, a.xcd_remap, a.sweep_s, a.sweep_w);
  const int64_t rb_begin = (int64_t)cid * tpb;
  const int64_t rb_end = (rb_begin + tpb < nrb) ? rb_begin + tpb : nrb;
  dd dacc[2];
  dacc[0] = dd{0.0, 0.0};
  dacc[1] = dd{0.0, 0.0};
  if (!COLS32) {
    for (int i = tid; i < 256; i += kBlock) s_tab[i] = i < a.code_T ? a.code_tab[i] : 0;
    __syncthreads();
  }
  for (int64_t rb = rb_begin; rb < rb_end; ++rb) {
    int64_t r0 = a.row_lo + rb * ROWS;
    if (r0 >= a.hole_lo) r0 += a.hole_len;   // the second range of a two-range launch
    const int nr = (int)((a.row_hi - r0) < ROWS ? (a.row_hi - r0) : ROWS);
    if (tid < nr) {
      const int64_t rowl = r0 + tid;
      const int32_t row = (int32_t)rowl;
      const int64_t sl = rowl >> 6;
      int64_t o0;
      int T;
      if (a.sell_units) { o0 = sl * a.sell_units; T = a.sell_units; }
      else { const uint32_t b0 = a.sell_off[sl], b1 = a.sell_off[sl + 1]; o0 = (int64_t)b0; T = (int)(b1 - b0); }
      const int W = T <= 0 ? 0 : (PAIRG ? (COLS32 ? 2 * ((T - 4) / 6) + 2 : (T <= 18 ? 2 : (T <= 36 ? 4 : (T <= 54 ? 6 : 8))))
                                        : (PAIR ? 1 : (C4 ? 0 : (COLS32 ? (T + 2) / 3 : (T + 8) / 9))));
      const int L = T - W;
      const unsigned long long *base = a.sell + (size_t)o0 * 64 + (rowl & 63);
      double acc = 0.0, wv = 0.0;
      if (DOT) wv = a.dotw[rowl];                 // in flight beside the row's entries (2.06 -> 1.97 ms fused at 512^3, profiles/r06ap/aq)
      if (PAIR) {                                // T = 8 or 10 words: four or five 16-byte loads bring the whole row
        const dbl2 *pb = reinterpret_cast<const dbl2 *>(a.sell + (size_t)o0 * 64) + (rowl & 63);
        dbl2 el[5];
#pragma unroll
        for (int e = 0; e < 5; ++e) el[e] = (2 * e < T) ? ld<NTM>(pb + (size_t)e * 64) : dbl2{0.0, 0.0};
        const unsigned long long cw = (unsigned long long)__double_as_longlong(el[0].x);
        double vv[8], xx[8];
        int32_t cc[8];
        bool on[8];
        vv[0] = el[0].y; vv[1] = el[1].x; vv[2] = el[1].y; vv[3] = el[2].x; vv[4] = el[2].y; vv[5] = el[3].x; vv[6] = el[3].y; vv[7] = el[4].x;
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int c = (int)((cw >> (8 * u)) & 0xFFull); on[u] = T > 0 && c != 0xFF; cc[u] = row + s_tab[c]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          xx[u] = 0.0;
          if (on[u]) xx[u] = gather_x<DIST>(a, cc[u]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (on[u]) {
            const double prod = vv[u] * xx[u];
            acc = acc + prod;
          }
        }
      } else if (PAIRG) {                        // eight entries per step: four 16-byte value loads + half a code element, or two column elements
        const dbl2 *pb = reinterpret_cast<const dbl2 *>(a.sell + (size_t)o0 * 64) + (rowl & 63);
        dbl2 ce = dbl2{0.0, 0.0};
        for (int k0 = 0; k0 < L; k0 += 8) {
          const int left = L - k0;               // >= 2, even
          double vv[8], xx[8];
          int32_t cc[8];
          bool on[8];
          if (COLS32) {
            const dbl2 c0 = ld<NTM>(pb + (size_t)(k0 / 4) * 64);
            const dbl2 c1 = (4 < left) ? ld<NTM>(pb + (size_t)(k0 / 4 + 1) * 64) : dbl2{0.0, 0.0};
            const unsigned long long w0 = (unsigned long long)__double_as_longlong(c0.x), w1 = (unsigned long long)__double_as_longlong(c0.y);
            const unsigned long long w2 = (4 < left) ? (unsigned long long)__double_as_longlong(c1.x) : ~0ull, w3 = (4 < left) ? (unsigned long long)__double_as_longlong(c1.y) : ~0ull;
            cc[0] = (int32_t)(uint32_t)w0; cc[1] = (int32_t)(uint32_t)(w0 >> 32); cc[2] = (int32_t)(uint32_t)w1; cc[3] = (int32_t)(uint32_t)(w1 >> 32);
            cc[4] = (int32_t)(uint32_t)w2; cc[5] = (int32_t)(uint32_t)(w2 >> 32); cc[6] = (int32_t)(uint32_t)w3; cc[7] = (int32_t)(uint32_t)(w3 >> 32);
#pragma unroll
            for (int u = 0; u < 8; ++u) on[u] = u < left && cc[u] != -1;
          } else {
            const int w = k0 >> 3;
            if ((w & 1) == 0) ce = ld<NTM>(pb + (size_t)(w >> 1) * 64);
            const unsigned long long cw = (unsigned long long)__double_as_longlong((w & 1) ? ce.y : ce.x);
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int c = (int)((cw >> (8 * u)) & 0xFFull); on[u] = c != 0xFF; cc[u] = row + s_tab[c]; }
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const dbl2 v = (2 * j < left) ? ld<NTM>(pb + (size_t)((W + k0) / 2 + j) * 64) : dbl2{0.0, 0.0};
            vv[2 * j] = v.x; vv[2 * j + 1] = v.y;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            xx[u] = 0.0;
            if (on[u]) xx[u] = gather_x<DIST>(a, cc[u]);
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            if (on[u]) {
              const double prod = vv[u] * xx[u];
              acc = acc + prod;
            }
          }
        }
      } else
      for (int k0 = 0; k0 < L; k0 += 8) {      // eight entries per step: one code word, or four words of two int32 columns
        const unsigned long long *vb = base + (size_t)(W + k0) * 64;
        const int left = L - k0;               // >= 1
        double vv[8], xx[8];
        int32_t cc[8];
        bool on[8];
        if (COLS32) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const unsigned long long cw = (2 * j < left) ? ld<NTM>(base + (size_t)(k0 / 2 + j) * 64) : ~0ull;
            cc[2 * j] = (int32_t)(uint32_t)(cw & 0xFFFFFFFFull);
            cc[2 * j + 1] = (int32_t)(uint32_t)(cw >> 32);
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) on[u] = cc[u] != -1;
        } else if (C4) {
          const uint32_t cw = ld<NTM>(a.sell_c4 + rowl);          // L <= 8: one step
#pragma unroll
          for (int u = 0; u < 8; ++u) { const int c = (int)((cw >> (4 * u)) & 0xFu); on[u] = c != 0xF; cc[u] = row + s_tab[c]; }
        } else {
          const unsigned long long cw = ld<NTM>(base + (size_t)(k0 / 8) * 64);
#pragma unroll
          for (int u = 0; u < 8; ++u) { const int c = (int)((cw >> (8 * u)) & 0xFFull); on[u] = c != 0xFF; cc[u] = row + s_tab[c]; }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) vv[u] = (u < left) ? __longlong_as_double((long long)ld<NTM>(vb + (size_t)u * 64)) : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          xx[u] = 0.0;
          if (on[u]) xx[u] = gather_x<DIST>(a, cc[u]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (on[u]) {
            const double prod = vv[u] * xx[u];
            acc = acc + prod;
          }
        }
      }
      store_y(a, rowl, acc);
      if (DOT) {
        acc_prod<COMP>(dacc[0], wv, acc);
        if (a.dot_sq == 1) acc_prod<COMP>(dacc[1], acc, acc);
        else if (a.dot_sq == 2) acc_prod<COMP>(dacc[1], wv, wv);
      }
    }
  }
  if (DOT) {
    if (a.blk_pub) {
      dd *s_red = reinterpret_cast<dd *>(s_stage);
      if (a.dot_sq) block_publish<2>(dacc, ra, s_red);
      else block_publish<1>(reinterpret_cast<dd (&)[1]>(dacc), ra, s_red);
    } else if (a.dot_sq) wave_publish<2>(dacc, ra);
    else wave_publish<1>(reinterpret_cast<dd (&)[1]>(dacc), ra);
  }
}

// ---------------------------------------------------------------- staged rows, software-pipelined ----
// The staged kernels above run one dependent chain per workgroup -- row pointers -> window (val + columns) -> x gathers
// -> y -- and only the middle link carries the bulk of the bytes: with 8 workgroups per CU about 40 % of them have their
// window in flight at any time, ~50 KB per CU, and by Little's law that, not HBM, sets the rate (5.2-5.5 TB/s of real
// traffic while the streaming BLAS-1 kernels reach 6.2-6.7).  This kernel gives every workgroup a run of `tiles_per_block`
// CONSECUTIVE row blocks and keeps three of them in different stages at once:
//     iteration t:   walk block t out of LDS (LDS reads, x gathers issued)
//                    -> issue the window loads of block t+1 into registers   (its row pointers arrived an iteration ago)
//                    -> issue the row-pointer loads of block t+2
//                    -> wait for the gathers only (counted vmcnt: vector loads return in order and the gathers are the
//                       OLDEST outstanding ones), finish the rows, store y
// so a workgroup has a window in flight practically all the time.  Everything is vector memory (the block-uniform row
// pointers too: scalar loads return out of order and would turn every LDS wait into a wait for them), the loop body is
// straight-line, lanes without a k-th entry are predicated by selects (they gather x[0]), and barriers order LDS only
// (s_waitcnt lgkmcnt(0) + s_barrier: __syncthreads() would drain the prefetches).  Arithmetic per row is unchanged --
// stored order, one rounded multiply and one rounded add per entry -- so y is bit-identical to the other kernels.
// CODE = uint8_t / uint16_t: dictionary-coded columns (colcode.hip); CODE = int32_t: the plain CSR column stream.
// Requires every row block to fit one window (rows * max_row_nnz + 3 <= 2048) and nnz > 0; launch_spmv checks.
template <> struct code_load<int32_t> {    // 4 columns = four dwords
  typedef u32x4 vec;
  template <int AUX = 0>
  static __device__ __forceinline__ vec ld(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, AUX);
  }
};

__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct RowPtrSet { int s, e, a, b; };      // block start / end (uniform), this lane's row start / end

template <typename CODE, bool DOT, bool COMP, bool DIST>
__global__ __launch_bounds__(kBlock) void spmv_pipe_kernel(SpmvArgs a, RedArgs ra) {
  if (seq_skip(a.stop_seq, a.seq)) return;
  typedef typename code_load<CODE>::vec cvec;
  constexpr bool CODED = sizeof(CODE) < 4;
  constexpr int UK = 8;
  const int ROWS = a.stage_rows;
  const int CAP = a.stage_cap;
  extern __shared__ __attribute__((aligned(16))) unsigned char s_stage[];
  double *s_val = reinterpret_cast<double *>(s_stage);
  CODE *s_code = reinterpret_cast<CODE *>(s_stage + (size_t)CAP * sizeof(double));
  int32_t *s_tab = reinterpret_cast<int32_t *>(s_stage + (size_t)CAP * (sizeof(double) + sizeof(CODE)));
  const int tid = threadIdx.x;
  const CODE *code = CODED ? reinterpret_cast<const CODE *>(a.code) : reinterpret_cast<const CODE *>(a.col);
  const int64_t nrows = a.row_hi - a.row_lo;
  const int64_t nrb = (nrows + ROWS - 1) / ROWS;
  const int tpb = a.tiles_per_block > 0 ? a.tiles_per_block : 1;
  const int cid = chunk_id(blockIdx.x, gridDim.x, a.xcd_remap, a.sweep_s, a.sweep_w);
  const int64_t rb_begin = (int64_t)cid * tpb;
  const int64_t rb_end = (rb_begin + tpb < nrb) ? rb_begin + tpb : nrb;
  const int kmax = a.max_row;
  dd dacc[2];
  dacc[0] = dd{0.0, 0.0};
  dacc[1] = dd{0.0, 0.0};
  if (CODED)
    for (int i = tid; i < a.code_T; i += kBlock) s_tab[i] = a.code_tab[i];

  // row pointers of block rb; blocks past the run are empty (all four words equal) and cost one cached line
  auto load_rp = [&](int64_t rb) -> RowPtrSet {
    int64_t r0 = a.row_lo + rb * ROWS;
    if (rb >= rb_end || r0 > a.row_hi) r0 = a.row_hi;
    const int64_t rend = (r0 + ROWS < a.row_hi) ? r0 + ROWS : a.row_hi;
    const int64_t ia = (r0 + tid < rend) ? r0 + tid : rend;
    const int64_t ib = (r0 + tid + 1 < rend) ? r0 + tid + 1 : rend;
    RowPtrSet p;
    p.s = a.rowptr[r0];
    p.e = a.rowptr[rend];
    p.a = a.rowptr[ia];
    p.b = a.rowptr[ib];
    return p;
  };
  u32x4 v[4];
  cvec c[2];
  auto issue_window = [&](const RowPtrSet &p) {
    const int s = __builtin_amdgcn_readfirstlane(p.s), e = __builtin_amdgcn_readfirstlane(p.e);
    const int c0 = s & ~3;
    const int lim4 = (e - c0 + 3) & ~3;
    const __amdgpu_buffer_rsrc_t rv =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(a.val + c0), 0, lim4 * 8, kBufRsrcWord3);
    const __amdgpu_buffer_rsrc_t rc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<CODE *>(code + c0), 0, lim4 * (int)sizeof(CODE), kBufRsrcWord3);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int o = q * (4 * kBlock) + 4 * tid;
      v[2 * q] = __builtin_amdgcn_raw_buffer_load_b128(rv, o * 8, 0, 0);
      v[2 * q + 1] = __builtin_amdgcn_raw_buffer_load_b128(rv, o * 8 + 16, 0, 0);
      c[q] = code_load<CODE>::ld(rc, o * (int)sizeof(CODE));
    }
  };

  RowPtrSet p0 = load_rp(rb_begin);
  issue_window(p0);
  RowPtrSet p1 = load_rp(rb_begin + 1);

  for (int64_t rb = rb_begin; rb < rb_end; ++rb) {
    // window of block rb: registers -> LDS (waits for its loads; the row pointers issued behind them stay in flight)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int o = q * (4 * kBlock) + 4 * tid;
      if (o < CAP) {
        *reinterpret_cast<u32x4 *>(s_val + o) = v[2 * q];
        *reinterpret_cast<u32x4 *>(s_val + o + 2) = v[2 * q + 1];
        *reinterpret_cast<cvec *>(s_code + o) = c[q];
      }
    }
    lds_only_barrier();
    const int64_t r0 = a.row_lo + rb * ROWS;
    const int64_t rend = (r0 + ROWS < a.row_hi) ? r0 + ROWS : a.row_hi;
    const bool live = r0 + tid < rend;
    const int32_t row = (int32_t)(r0 + tid);
    const int base = p0.a - (p0.s & ~3);
    const int cnt = p0.b - p0.a;
    double acc = 0.0;
    int k0 = 0;
    for (; k0 + UK < kmax; k0 += UK) {          // rows longer than UK entries: all batches but the last, plain
      int32_t cc[UK];
      double vv[UK], xx[UK];
#pragma unroll
      for (int u = 0; u < UK; ++u) {
        const int j = (k0 + u < cnt) ? base + k0 + u : 0;
        cc[u] = (int32_t)s_code[j];
        vv[u] = s_val[j];
      }
#pragma unroll
      for (int u = 0; u < UK; ++u) {
        const int32_t col = CODED ? row + s_tab[cc[u]] : cc[u];
        xx[u] = gather_x<DIST>(a, (k0 + u < cnt) ? col : 0);
      }
#pragma unroll
      for (int u = 0; u < UK; ++u) {
        const double prod = vv[u] * xx[u];
        const double next = acc + prod;
        acc = (k0 + u < cnt) ? next : acc;
      }
    }
    {   // last batch, with the prefetches between the issue of its gathers and their use
      int32_t cc[UK];
      double vv[UK], xx[UK];
#pragma unroll
      for (int u = 0; u < UK; ++u) {
        const int j = (k0 + u < cnt) ? base + k0 + u : 0;
        cc[u] = (int32_t)s_code[j];
        vv[u] = s_val[j];
      }
#pragma unroll
      for (int u = 0; u < UK; ++u) {
        const int32_t col = CODED ? row + s_tab[cc[u]] : cc[u];
        xx[u] = gather_x<DIST>(a, (k0 + u < cnt) ? col : 0);
      }
      double wv = 0.0;
      if (DOT) wv = a.dotw[live ? r0 + tid : rend - 1];
      __builtin_amdgcn_sched_barrier(0);
      issue_window(p1);                          // block rb + 1 (v, c are free: they went to LDS before the barrier)
      const RowPtrSet p2 = load_rp(rb + 2);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < UK; ++u) {
        const double prod = vv[u] * xx[u];
        const double next = acc + prod;
        acc = (k0 + u < cnt) ? next : acc;
      }
      if (live) {
        store_y(a, r0 + tid, acc);
        if (DOT) {
          acc_prod<COMP>(dacc[0], wv, acc);
          if (a.dot_sq == 1) acc_prod<COMP>(dacc[1], acc, acc);
          else if (a.dot_sq == 2) acc_prod<COMP>(dacc[1], wv, wv);
        }
      }
      p0 = p1;
      p1 = p2;
    }
    lds_only_barrier();                          // every lane is done with the LDS window
  }
  if (DOT) {
    if (a.dot_sq) wave_publish<2>(dacc, ra);
    else wave_publish<1>(reinterpret_cast<dd (&)[1]>(dacc), ra);
  }
}

// ---------------------------------------------------------------- stream, wide loads, block-delta columns ----
// The stream kernel (products in nnz order into LDS, then one lane per row sums its segment in stored order) for
// mid-length rows, rebuilt around two measurements: (1) a wave's vector-memory instruction costs the CU about the same
// whatever its width (profiles/r02_l2bench.log), so the val / column window is moved as 16-byte buffer loads with the
// block's extent in the descriptor (4 + 2 instructions per lane and 2048 entries instead of 16 of 8 and 4 bytes);
// (2) the bytes are what is left to cut: with CODE = uint8_t / uint16_t the columns come as block-delta codes
// (coldelta.hip: col = base[block] + code, 1 or 2 bytes per entry), the entries that do not fit (code = all ones: long-range
// links, dense rows) from the block's escape list -- (position, int32 column) pairs patched into the product window by
// the first lanes after the main pass.  CODE = int32_t reads the plain CSR columns (no escapes).  A lane holds four
// consecutive entries; its four gathers are issued together.  Arithmetic per row is unchanged -- one rounded multiply per
// entry, rounded adds in stored order -- so y is bit-identical to every other kernel and to the serial loop.
// Requires row_lo to be a multiple of the row block (launch_spmv checks).
template <typename CODE, bool DOT, bool COMP, bool DIST>
__global__ __launch_bounds__(kBlock) void spmv_delta_kernel(SpmvArgs a, RedArgs ra) {
  if (seq_skip(a.stop_seq, a.seq)) return;
  typedef typename code_load<CODE>::vec cvec;
  constexpr bool PLAIN = sizeof(CODE) == 4;
  constexpr unsigned ESC = PLAIN ? 0xffffffffu : ((1u << (8 * (sizeof(CODE) & 3))) - 1u);
  constexpr int CAP = 2048;                  // products per window (16 KB)
  __shared__ __attribute__((aligned(16))) double s_prod[CAP + 8];
  const int ROWS = a.stage_rows;
  const int tid = threadIdx.x;
  const CODE *code = reinterpret_cast<const CODE *>(PLAIN ? (const void *)a.col : a.dcode);
  const int64_t nrows = a.row_hi - a.row_lo;
  const int64_t nrb = (nrows + ROWS - 1) / ROWS;
  const int G = gridDim.x;
  const int cid = chunk_id(blockIdx.x, G, a.xcd_remap);
  const int64_t rb_begin = nrb * cid / G;
  const int64_t rb_end = nrb * (cid + 1) / G;
  dd dacc[2];
  dacc[0] = dd{0.0, 0.0};
  dacc[1] = dd{0.0, 0.0};

  for (int64_t rb = rb_begin; rb < rb_end; ++rb) {
    const int64_t r0 = a.row_lo + rb * ROWS;
    const int nr = (int)((a.row_hi - r0) < ROWS ? (a.row_hi - r0) : ROWS);
    const int64_t s = a.rowptr[r0], e = a.rowptr[r0 + nr];
    const int my_a = (tid < nr) ? a.rowptr[r0 + tid] : 0;
    const int my_b = (tid < nr) ? a.rowptr[r0 + tid + 1] : 0;
    int32_t base = 0, E0 = 0, E1 = 0;
    if (!PLAIN) {
      const int64_t bi = r0 / ROWS;
      base = a.dbase[bi];
      E0 = a.desc_ptr[bi];
      E1 = a.desc_ptr[bi + 1];
    }
    double acc = 0.0, wv = 0.0;
    if (DOT && a.dot_early && tid < nr) wv = a.dotw[r0 + tid];
    for (int64_t c0 = s & ~(int64_t)3; c0 < e; c0 += CAP) {
      const int lim = (int)((e - c0) < (int64_t)CAP ? (e - c0) : (int64_t)CAP);
      const int lim4 = (lim + 3) & ~3;
      const int first = (int)(s - c0) > 0 ? (int)(s - c0) : 0;      // entries below it belong to the previous block (other base)
      const __amdgpu_buffer_rsrc_t rv =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(a.val + c0), 0, lim4 * 8, kBufRsrcWord3);
      const __amdgpu_buffer_rsrc_t rc =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<CODE *>(code + c0), 0, lim4 * (int)sizeof(CODE), kBufRsrcWord3);
      u32x4 v[4];
      cvec c[2];
      if (a.stream_nt) {       // matrix stream with the non-temporal policy (read once, by this CU): x lines stay longer in the L2
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int o = q * (4 * kBlock) + 4 * tid;
          v[2 * q] = __builtin_amdgcn_raw_buffer_load_b128(rv, o * 8, 0, 2);
          v[2 * q + 1] = __builtin_amdgcn_raw_buffer_load_b128(rv, o * 8 + 16, 0, 2);
          c[q] = code_load<CODE>::template ld<2>(rc, o * (int)sizeof(CODE));
        }
      } else {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int o = q * (4 * kBlock) + 4 * tid;
          v[2 * q] = __builtin_amdgcn_raw_buffer_load_b128(rv, o * 8, 0, 0);
          v[2 * q + 1] = __builtin_amdgcn_raw_buffer_load_b128(rv, o * 8 + 16, 0, 0);
          c[q] = code_load<CODE>::ld(rc, o * (int)sizeof(CODE));
        }
      }
      // this lane's escape of the block (blocks with more than 256 escapes: the loop behind the barrier takes the rest)
      int epos = 0;
      int32_t ecol = 0;
      const int eb = E0 + tid;
      const bool has_esc = !PLAIN && eb < E1;
      if (has_esc) { epos = a.desc_pos[eb]; ecol = a.desc_col[eb]; }
      // columns of this lane's 8 entries
      int32_t cc[8];
      bool ok[8];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int o = q * (4 * kBlock) + 4 * tid;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          unsigned cd;
          if (PLAIN) cd = ((const unsigned *)&c[q])[j];
          else if (sizeof(CODE) == 1) cd = (((const unsigned *)&c[q])[0] >> (8 * j)) & 0xffu;
          else cd = (((const unsigned *)&c[q])[j >> 1] >> (16 * (j & 1))) & 0xffffu;
          ok[4 * q + j] = (o + j >= first) && (o + j < lim) && (PLAIN || cd != ESC);
          cc[4 * q + j] = base + (int32_t)cd;
        }
      }
      double xx[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        xx[u] = 0.0;
        if (ok[u]) xx[u] = gather_x<DIST>(a, cc[u]);
      }
      // second level of the escape: its value (an L2 hit: this workgroup has just streamed it) and its x entry
      double ev = 0.0, ex = 0.0;
      int eidx = -1;
      if (has_esc) {
        eidx = (int)(s - c0) + epos;
        // an escape of an earlier / later window of a multi-window row block is skipped BEFORE its loads (a negative
        // index used to fetch val and gather x only to be discarded)
        if (eidx >= 0 && eidx < lim && eidx < CAP) { ev = a.val[s + epos]; ex = gather_x<DIST>(a, ecol); } else eidx = -1;
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int o = q * (4 * kBlock) + 4 * tid;
        const double *vv = reinterpret_cast<const double *>(&v[2 * q]);
        dbl2 p01, p23;
        p01.x = vv[0] * xx[4 * q];
        p01.y = vv[1] * xx[4 * q + 1];
        p23.x = vv[2] * xx[4 * q + 2];
        p23.y = vv[3] * xx[4 * q + 3];
        if (o < CAP) {
          *reinterpret_cast<dbl2 *>(s_prod + o) = p01;
          *reinterpret_cast<dbl2 *>(s_prod + o + 2) = p23;
        }
      }
      if (!PLAIN) {
        __syncthreads();
        if (eidx >= 0) s_prod[eidx] = ev * ex;
        for (int eb2 = eb + kBlock; eb2 < E1; eb2 += kBlock) {
          const int ep = a.desc_pos[eb2];
          const int ei = (int)(s - c0) + ep;
          if (ei >= 0 && ei < lim) s_prod[ei] = a.val[s + ep] * gather_x<DIST>(a, a.desc_col[eb2]);
        }
      }
      __syncthreads();
      if (tid < nr) {
        const int rel_a = (int)(my_a - c0), rel_b = (int)(my_b - c0);
        const int lo = rel_a > 0 ? rel_a : 0;
        const int hi = rel_b < lim ? rel_b : lim;
        int k = lo;
        for (; k + 4 <= hi; k += 4) {
          const double p0 = s_prod[k], p1 = s_prod[k + 1], p2 = s_prod[k + 2], p3 = s_prod[k + 3];
          acc = acc + p0; acc = acc + p1; acc = acc + p2; acc = acc + p3;
        }
        for (; k < hi; ++k) acc = acc + s_prod[k];
      }
      if (c0 + CAP < e) __syncthreads();
    }
    if (rb + 1 < rb_end) __syncthreads();
    if (tid < nr) {
      store_y(a, r0 + tid, acc);
      if (DOT) {
        if (!a.dot_early) wv = a.dotw[r0 + tid];
        acc_prod<COMP>(dacc[0], wv, acc);
        if (a.dot_sq == 1) acc_prod<COMP>(dacc[1], acc, acc);
        else if (a.dot_sq == 2) acc_prod<COMP>(dacc[1], wv, wv);
      }
    }
  }
  if (DOT) {
    if (a.blk_pub) {
      __syncthreads();               // the last window's row sums are done in every wave: s_prod is free
      dd *s_red = reinterpret_cast<dd *>(s_prod);
      if (a.dot_sq) block_publish<2>(dacc, ra, s_red);
      else block_publish<1>(reinterpret_cast<dd (&)[1]>(dacc), ra, s_red);
    } else if (a.dot_sq) wave_publish<2>(dacc, ra);
    else wave_publish<1>(reinterpret_cast<dd (&)[1]>(dacc), ra);
  }
}

// ---------------------------------------------------------------- wave-private windows (LDS-DMA) ----
// One WAVE owns 64 consecutive rows, one lane per row, and the wave is the whole workgroup: no barrier anywhere.  The val / col
// entries of its rows land in the wave's LDS window by global_load_lds_dwordx4 (1 KiB per instruction, no staging registers,
// no ds_write -- gfx950), then every lane walks its row out of LDS in stored order, eight gathers in flight: at step k the
// wave gathers x for 64 CONSECUTIVE rows, which for the near-diagonal part of any banded operator is a coalesced access
// (the texture path handles it in ~16 cycles; the same entries taken in nnz order by the stream kernels cost ~100 because
// a wave then touches ~10 scattered lines -- TA_BUSY 80 % of the kernel on the banded + random operator,
// profiles/r04_spmv_irregular_pmc.json).  Unlike the staged kernel (256 threads, rows x entries <= 2048 per block, so 27
// entries per row leave 3 of 4 waves idle during the walk) every resident wave walks all the time, and 6 ... 13 independent
// waves per CU overlap each other's copies and walks.  Rows longer than the window are walked window by window (lanes
// whose row does not reach into a window sit it out).  Arithmetic per row as everywhere: rounded multiply, rounded add, stored
// order => y bit-identical to the serial loop.
typedef __attribute__((address_space(3))) char lds_char_t;
template <bool DOT, bool COMP, bool DIST>
__global__ __launch_bounds__(64) void spmv_wave_kernel(SpmvArgs a, RedArgs ra) {
  if (seq_skip(a.stop_seq, a.seq)) return;
  constexpr int UK = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char s_stage[];
  const int CAP = a.stage_cap;               // entries per window: a multiple of 256 (whole 1-KiB copies of both streams)
  double *s_val = reinterpret_cast<double *>(s_stage);
  int32_t *s_col = reinterpret_cast<int32_t *>(s_stage + (size_t)CAP * sizeof(double));
  const unsigned lds_val = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(lds_char_t *)s_stage);
  const unsigned lds_col = lds_val + (unsigned)CAP * 8u;
  const int lane = threadIdx.x;
  const int64_t nrows = a.row_hi - a.row_lo;
  const int64_t ngrp = (nrows + 63) / 64;
  dd dacc[2];
  dacc[0] = dd{0.0, 0.0};
  dacc[1] = dd{0.0, 0.0};
  for (int64_t g = blockIdx.x; g < ngrp; g += gridDim.x) {
    const int64_t r0 = a.row_lo + g * 64;
    const int nr = (int)((a.row_hi - r0) < 64 ? (a.row_hi - r0) : 64);
    const int my_a = a.rowptr[r0 + (lane < nr ? lane : nr)];          // lanes past the last row: an empty row at the group's end
    const int my_b = a.rowptr[r0 + (lane < nr ? lane + 1 : nr)];
    const int64_t s = __builtin_amdgcn_readfirstlane(my_a);
    const int64_t e = __builtin_amdgcn_readlane(my_b, 63);
    double acc = 0.0, wv = 0.0;
    if (DOT && lane < nr) wv = a.dotw[r0 + lane];
    for (int64_t c0 = s & ~(int64_t)3; c0 < e; c0 += CAP) {
      const int lim = (int)((e - c0) < (int64_t)CAP ? (e - c0) : (int64_t)CAP);
      // copies: val 128 entries per instruction, col 256; lanes past the window's end are masked off (no request)
      const int nv = (lim + 127) >> 7, nc = (lim + 255) >> 8;
      for (int i = 0; i < nv; ++i) {
        const int o = i * 128 + lane * 2;
        if (o < lim) {
          const double *gsrc = a.val + c0 + o;
          const unsigned dst = lds_val + 1024u * (unsigned)i;
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
        }
      }
      for (int i = 0; i < nc; ++i) {
        const int o = i * 256 + lane * 4;
        if (o < lim) {
          const int32_t *gsrc = a.col + c0 + o;
          const unsigned dst = lds_col + 1024u * (unsigned)i;
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      const int rel_a = (int)(my_a - c0), rel_b = (int)(my_b - c0);
      const int lo = rel_a > 0 ? rel_a : 0;
      const int hi = rel_b < lim ? rel_b : lim;
      for (int k0 = lo; k0 < hi; k0 += UK) {
        int32_t cc[UK];
        double vv[UK], xx[UK];
#pragma unroll
        for (int u = 0; u < UK; ++u) {
          const int j = (k0 + u < hi) ? k0 + u : k0;
          cc[u] = s_col[j];
          vv[u] = s_val[j];
        }
#pragma unroll
        for (int u = 0; u < UK; ++u) {
          xx[u] = 0.0;
          if (k0 + u < hi) xx[u] = gather_x<DIST>(a, cc[u]);
        }
#pragma unroll
        for (int u = 0; u < UK; ++u) {
          if (k0 + u < hi) {
            const double prod = vv[u] * xx[u];
            acc = acc + prod;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();      // every lane is done with the window before the next copies overwrite it
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (lane < nr) {
      if (a.nt_y) __builtin_nontemporal_store(acc, a.y + r0 + lane); else a.y[r0 + lane] = acc;
      if (DOT) {
        acc_prod<COMP>(dacc[0], wv, acc);
        if (a.dot_sq == 1) acc_prod<COMP>(dacc[1], acc, acc);
        else if (a.dot_sq == 2) acc_prod<COMP>(dacc[1], wv, wv);
      }
    }
  }
  if (DOT) {
    if (a.dot_sq) wave_publish<2>(dacc, ra);
    else wave_publish<1>(reinterpret_cast<dd (&)[1]>(dacc), ra);
  }
}

// ---------------------------------------------------------------- row templates ----------
// Compressed handles (template.hip): a row is a 16-bit id into a table of (column - row, value) sequences
// held in LDS.  One lane per row; lanes of a wave mostly share the template (LDS broadcast), and at step k
// they gather x at row + off_k for 64 consecutive rows: coalesced.  Same rounded multiply / rounded add per
// entry in stored order as every other kernel here => bit-identical y.  Matrix traffic: 2 bytes per row.
template <bool DOT, bool COMP, bool DIST>
__global__ __launch_bounds__(kBlock) void spmv_template_kernel(SpmvArgs a, RedArgs ra) {
  if (seq_skip(a.stop_seq, a.seq)) return;
  constexpr int UK = 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char s_tab[];
  const int T = a.tmpl_T, K = a.tmpl_K;
  double *s_val = reinterpret_cast<double *>(s_tab);
  int32_t *s_off = reinterpret_cast<int32_t *>(s_val + (size_t)T * K);
  int32_t *s_cnt = s_off + (size_t)T * K;
  for (int i = threadIdx.x; i < T * K; i += kBlock) { s_val[i] = a.tmpl_val[i]; s_off[i] = a.tmpl_off[i]; }
  for (int i = threadIdx.x; i < T; i += kBlock) s_cnt[i] = a.tmpl_cnt[i];
  __syncthreads();
  dd dacc[2];
  dacc[0] = dd{0.0, 0.0};
  dacc[1] = dd{0.0, 0.0};
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t row = a.row_lo + (int64_t)blockIdx.x * kBlock + threadIdx.x; row < a.row_hi; row += stride) {
    const int t = a.tmpl_id[row];
    const int cnt = s_cnt[t];
    const double *tv = s_val + (size_t)t * K;
    const int32_t *to = s_off + (size_t)t * K;
    double acc = 0.0;
    for (int k0 = 0; k0 < cnt; k0 += UK) {
      double vv[UK], xx[UK];
#pragma unroll
      for (int u = 0; u < UK; ++u) {
        xx[u] = 0.0;
        if (k0 + u < cnt) {
          vv[u] = tv[k0 + u];
          xx[u] = gather_x<DIST>(a, (int32_t)row + to[k0 + u]);
        }
      }
#pragma unroll
      for (int u = 0; u < UK; ++u) {
        if (k0 + u < cnt) {
          const double prod = vv[u] * xx[u];
          acc = acc + prod;
        }
      }
    }
    if (a.nt_y) __builtin_nontemporal_store(acc, a.y + row); else a.y[row] = acc;
    if (DOT) {
      const double wv = a.dotw[row];
      acc_prod<COMP>(dacc[0], wv, acc);
      if (a.dot_sq == 1) acc_prod<COMP>(dacc[1], acc, acc);
      else if (a.dot_sq == 2) acc_prod<COMP>(dacc[1], wv, wv);
    }
  }
  if (DOT) {
    if (a.dot_sq) wave_publish<2>(dacc, ra);
    else wave_publish<1>(reinterpret_cast<dd (&)[1]>(dacc), ra);
  }
}

// ---------------------------------------------------------------- ordered sub-wave -------
// L lanes per row, RPG rows per lane group in flight.  Wave layout: NG = 64 / L groups; step u of a
// wave covers rows wave_row0 + u * NG + g (g = group index) so that each load instruction of the
// wave touches one contiguous span of the val / col streams.
template <int L, int RPG, bool NT, bool DOT, bool COMP, bool DIST>
__global__ __launch_bounds__(kBlock) void spmv_ordered_kernel(SpmvArgs a, RedArgs ra) {
  if (seq_skip(a.stop_seq, a.seq)) return;
  constexpr int NG = 64 / L;                       // lane groups per wave
  constexpr int ROWS_PER_WAVE = NG * RPG;
  constexpr int ROWS_PER_BLOCK = ROWS_PER_WAVE * kWavesPerBlock;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane / L, k = lane % L;
  dd dacc[1];
  dacc[0] = dd{0.0, 0.0};
  const int64_t nrows = a.row_hi - a.row_lo;
  const int64_t ntiles = (nrows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t wrow0 = a.row_lo + tile * ROWS_PER_BLOCK + (int64_t)wave * ROWS_PER_WAVE;
    int32_t rs[RPG], re[RPG];
    double prod[RPG];
#pragma unroll
    for (int u = 0; u < RPG; ++u) {
      const int64_t row = wrow0 + u * NG + g;
      const bool ok = row < a.row_hi;
      rs[u] = ok ? a.rowptr[row] : 0;
      re[u] = ok ? a.rowptr[row + 1] : 0;
    }
    double v[RPG];
    int32_t c[RPG];
#pragma unroll
    for (int u = 0; u < RPG; ++u) {
      const bool valid = rs[u] + k < re[u];
      const int64_t j = valid ? (int64_t)rs[u] + k : 0;
      v[u] = ld<NT>(a.val + j);
      c[u] = ld<NT>(a.col + j);
      if (!valid) { v[u] = 0.0; c[u] = 0; }
    }
#pragma unroll
    for (int u = 0; u < RPG; ++u) {
      const double xv = gather_x<DIST>(a, c[u]);
      prod[u] = (rs[u] + k < re[u]) ? v[u] * xv : 0.0;
    }
    double mine = 0.0;        // result of the row this lane will store (row u == k for k < RPG)
#pragma unroll
    for (int u = 0; u < RPG; ++u) {
      // in-order fold of the first L entries: acc = (((0 + p0) + p1) + ...) ; padding adds +0.0
      double acc = 0.0;
#pragma unroll
      for (int t = 0; t < L; ++t) acc = acc + __shfl(prod[u], t, L);
      // rows longer than L (group-uniform loop): further chunks of L entries, still in order
      for (int32_t base = rs[u] + L; base < re[u]; base += L) {
        const bool valid = base + k < re[u];
        const int64_t j = valid ? (int64_t)base + k : 0;
        const double vv = a.val[j];
        const int32_t cc = a.col[j];
        const double p = valid ? vv * gather_x<DIST>(a, cc) : 0.0;
#pragma unroll
        for (int t = 0; t < L; ++t) acc = acc + __shfl(p, t, L);
      }
      if (k == u) mine = acc;
    }
    if (k < RPG) {
      const int64_t row = wrow0 + k * NG + g;
      if (row < a.row_hi) {
        a.y[row] = mine;
        if (DOT) acc_prod<COMP>(dacc[0], a.dotw[row], mine);
      }
    }
  }
  if (DOT) wave_publish<1>(dacc, ra);
}

// ---------------------------------------------------------------- vector (long rows) -----
template <int LPR, bool DOT, bool COMP, bool DIST>
__global__ __launch_bounds__(kBlock) void spmv_vector_kernel(SpmvArgs a, RedArgs ra) {
  if (seq_skip(a.stop_seq, a.seq)) return;
  constexpr int RPB = kBlock / LPR;   // rows per workgroup per sweep
  const int tid = threadIdx.x;
  const int sub = tid / LPR, sl = tid % LPR;
  dd dacc[1];
  dacc[0] = dd{0.0, 0.0};
  for (int64_t row = a.row_lo + (int64_t)blockIdx.x * RPB + sub; row < a.row_hi; row += (int64_t)gridDim.x * RPB) {
    const int64_t s = a.rowptr[row], e = a.rowptr[row + 1];
    double acc = 0.0;
    for (int64_t j = s + sl; j < e; j += LPR) acc = fma(a.val[j], gather_x<DIST>(a, a.col[j]), acc);
#pragma unroll
    for (int off = LPR / 2; off > 0; off >>= 1) acc += __shfl_down(acc, off, LPR);
    if (sl == 0) {
      a.y[row] = acc;
      if (DOT) acc_prod<COMP>(dacc[0], a.dotw[row], acc);
    }
  }
  if (DOT) wave_publish<1>(dacc, ra);
}

// ---------------------------------------------------------------- dispatch ------
#define KHIP_DISPATCH_DCD(LAUNCH)                                                        \
  do {                                                                                   \
    if (dot) {                                                                           \
      if (comp) { if (dist) LAUNCH(true, true, true); else LAUNCH(true, true, false); }  \
      else      { if (dist) LAUNCH(true, false, true); else LAUNCH(true, false, false); } \
    } else {                                                                             \
      if (dist) LAUNCH(false, false, true); else LAUNCH(false, false, false);            \
    }                                                                                    \
  } while (0)

template <int ROWS, int VEC, bool NT>
static void launch_stream_cfg(khip_ctx *ctx, const SpmvArgs &a, const RedArgs &ra, unsigned grid, bool dot, bool comp,
                              bool dist) {
#define KHIP_L(DOT, COMP, DIST) \
  hipLaunchKernelGGL((spmv_stream_kernel<ROWS, VEC, NT, DOT, COMP, DIST>), dim3(grid), dim3(kBlock), 0, ctx->stream, a, ra)
  KHIP_DISPATCH_DCD(KHIP_L);
#undef KHIP_L
}

template <int ROWS, bool NT>
static void launch_stage_cfg(khip_ctx *ctx, const SpmvArgs &a, const RedArgs &ra, unsigned grid, bool dot, bool comp,
                             bool dist) {
  size_t lds = (size_t)ctx->tune.spmv_lds_pad + 12u * (size_t)a.stage_cap;
  const size_t pub = dot && a.blk_pub ? sizeof(dd) * (size_t)kBlock * (a.dot_sq ? 2u : 1u) : 0u;    // block_publish reuses the window
  if (lds < pub) lds = pub;
#define KHIP_L(DOT, COMP, DIST) \
  hipLaunchKernelGGL((spmv_stage_kernel<ROWS, NT, DOT, COMP, DIST>), dim3(grid), dim3(kBlock), lds, ctx->stream, a, ra)
  KHIP_DISPATCH_DCD(KHIP_L);
#undef KHIP_L
}

template <typename CODE>
static void launch_code_cfg(khip_ctx *ctx, const SpmvArgs &a, const RedArgs &ra, unsigned grid, bool dot, bool comp,
                            bool dist) {
  size_t lds = (size_t)ctx->tune.spmv_lds_pad + (8u + sizeof(CODE)) * (size_t)a.stage_cap + 4u * (size_t)a.code_T;
  const size_t pub = dot && a.blk_pub ? sizeof(dd) * (size_t)kBlock * (a.dot_sq ? 2u : 1u) : 0u;    // block_publish reuses the window
  if (lds < pub) lds = pub;
#define KHIP_L(DOT, COMP, DIST) \
  hipLaunchKernelGGL((spmv_code_kernel<CODE, DOT, COMP, DIST>), dim3(grid), dim3(kBlock), lds, ctx->stream, a, ra)
  KHIP_DISPATCH_DCD(KHIP_L);
#undef KHIP_L
}

static void launch_sell_cfg(khip_ctx *ctx, const SpmvArgs &a, const RedArgs &ra, unsigned grid, bool dot, bool comp, bool dist) {
  const size_t pub = dot && a.blk_pub ? sizeof(dd) * (size_t)kBlock * (a.dot_sq ? 2u : 1u) : 0u;
  const size_t lds = pub + 4u * 256u;
#define KHIP_SELL(DOT, COMP, DIST, NTM, COLS32, C4) \
  do { if (a.sell_pair == 1) hipLaunchKernelGGL((spmv_sell_kernel<DOT, COMP, DIST, NTM, false, false, true, false>), dim3(grid), dim3(kBlock), lds, ctx->stream, a, ra); \
       else if (a.sell_pair == 2) hipLaunchKernelGGL((spmv_sell_kernel<DOT, COMP, DIST, NTM, COLS32, false, false, true>), dim3(grid), dim3(kBlock), lds, ctx->stream, a, ra); \
       else hipLaunchKernelGGL((spmv_sell_kernel<DOT, COMP, DIST, NTM, COLS32, C4, false, false>), dim3(grid), dim3(kBlock), lds, ctx->stream, a, ra); } while (0)
#define KHIP_L(DOT, COMP, DIST) \
  do { const bool ntm = ctx->tune.spmv_sell == 2; \
       if (a.sell_cols) { if (ntm) KHIP_SELL(DOT, COMP, DIST, true, true, false); else KHIP_SELL(DOT, COMP, DIST, false, true, false); } \
       else if (a.sell_c4) { if (ntm) KHIP_SELL(DOT, COMP, DIST, true, false, true); else KHIP_SELL(DOT, COMP, DIST, false, false, true); } \
       else { if (ntm) KHIP_SELL(DOT, COMP, DIST, true, false, false); else KHIP_SELL(DOT, COMP, DIST, false, false, false); } } while (0)
  KHIP_DISPATCH_DCD(KHIP_L);
#undef KHIP_L
#undef KHIP_SELL
}

template <typename CODE>
static void launch_pipe_cfg(khip_ctx *ctx, const SpmvArgs &a, const RedArgs &ra, unsigned grid, bool dot, bool comp,
                            bool dist) {
  const size_t lds = (size_t)ctx->tune.spmv_lds_pad + (8u + sizeof(CODE)) * (size_t)a.stage_cap +
                     (sizeof(CODE) < 4 ? 4u * (size_t)a.code_T : 0u);
#define KHIP_L(DOT, COMP, DIST) \
  hipLaunchKernelGGL((spmv_pipe_kernel<CODE, DOT, COMP, DIST>), dim3(grid), dim3(kBlock), lds, ctx->stream, a, ra)
  KHIP_DISPATCH_DCD(KHIP_L);
#undef KHIP_L
}

template <typename CODE>
static void launch_delta_cfg(khip_ctx *ctx, const SpmvArgs &a, const RedArgs &ra, unsigned grid, bool dot, bool comp,
                             bool dist) {
#define KHIP_L(DOT, COMP, DIST) \
  hipLaunchKernelGGL((spmv_delta_kernel<CODE, DOT, COMP, DIST>), dim3(grid), dim3(kBlock), 0, ctx->stream, a, ra)
  KHIP_DISPATCH_DCD(KHIP_L);
#undef KHIP_L
}

static void launch_wave_cfg(khip_ctx *ctx, const SpmvArgs &a, const RedArgs &ra, unsigned grid, bool dot, bool comp, bool dist) {
  const size_t lds = 12u * (size_t)a.stage_cap;
#define KHIP_L(DOT, COMP, DIST) \
  hipLaunchKernelGGL((spmv_wave_kernel<DOT, COMP, DIST>), dim3(grid), dim3(64), lds, ctx->stream, a, ra)
  KHIP_DISPATCH_DCD(KHIP_L);
#undef KHIP_L
}

template <int L, int RPG, bool NT>
static void launch_ordered_cfg(khip_ctx *ctx, const SpmvArgs &a, const RedArgs &ra, unsigned grid, bool dot, bool comp,
                               bool dist) {
#define KHIP_L(DOT, COMP, DIST) \
  hipLaunchKernelGGL((spmv_ordered_kernel<L, RPG, NT, DOT, COMP, DIST>), dim3(grid), dim3(kBlock), 0, ctx->stream, a, ra)
  KHIP_DISPATCH_DCD(KHIP_L);
#undef KHIP_L
}

template <int LPR>
static void launch_vector_cfg(khip_ctx *ctx, const SpmvArgs &a, const RedArgs &ra, unsigned grid, bool dot, bool comp,
                              bool dist) {
#define KHIP_L(DOT, COMP, DIST) \
  hipLaunchKernelGGL((spmv_vector_kernel<LPR, DOT, COMP, DIST>), dim3(grid), dim3(kBlock), 0, ctx->stream, a, ra)
  KHIP_DISPATCH_DCD(KHIP_L);
#undef KHIP_L
}

static void launch_template_cfg(khip_ctx *ctx, const SpmvArgs &a, const RedArgs &ra, unsigned grid, bool dot, bool comp,
                                bool dist) {
  const size_t lds = (size_t)a.tmpl_T * a.tmpl_K * 12 + (size_t)a.tmpl_T * 4;
#define KHIP_L(DOT, COMP, DIST) \
  hipLaunchKernelGGL((spmv_template_kernel<DOT, COMP, DIST>), dim3(grid), dim3(kBlock), lds, ctx->stream, a, ra)
  KHIP_DISPATCH_DCD(KHIP_L);
#undef KHIP_L
}

static inline unsigned pick_grid(khip_ctx *ctx, int64_t tiles, bool persist) {
  if (tiles < 1) tiles = 1;
  if (persist) {
    int64_t cap = (int64_t)ctx->num_cu * 8;
    int64_t g = tiles < cap ? tiles : cap;
    if (g >= 8) g &= ~(int64_t)7;      // XCD remap needs a multiple of 8
    return (unsigned)g;
  }
  const int64_t cap = 1 << 22;         // 4M workgroups: beyond that tiles are looped over
  return (unsigned)(tiles < cap ? tiles : cap);
}

// dot_slot >= 0 fuses x . y.  Several launches (interior + boundary ranges of a distributed operator) can
// feed ONE reduction: each passes the running partial count in *wave_cursor (updated here) and only the
// last one sets `finish`, which folds all partials written so far into results[dot_slot].
// which kernel launch_spmv will pick for this operator (the staged one is the only one with the y.y output)
int spmv_kernel_choice(const khip_ctx *ctx, const khip_csr *A) {
  if (A->tmpl_id && ctx->tune.spmv_template) return 5;      // compressed handle (khip_csr_compress)
  int kernel = ctx->tune.spmv_kernel;
  // Measured on seven operators (profiles/r03_sweep_spmv_choice.log, int32 columns): the staged-rows kernel wins for short
  // rows (7 per row: 0.69-0.71 of the HBM peak on the algorithmic bytes, stream kernel 0.61, ordered 0.46-0.51), the stream
  // kernel for everything longer (27 per row: 0.51-0.59, staged 0.41-0.59, ordered 0.13; 75 per row: 0.49 / 0.28 / 0.05);
  // the ordered sub-wave kernel (one row per lane group, serial fold by shuffles) never -- it stays selectable (3).  An
  // operator whose columns are dictionary coded (colcode.hip: stencils of up to 64 entries per row) keeps the staged kernel,
  // which is the one that streams the codes.  Very long rows: strided vector kernel (not bit-identical).
  if (kernel == 0) {
    const bool short_rows = A->mean_row_nnz <= 12.0 && A->max_row_nnz <= 64;
    const bool coded = A->code_state == 1 && ctx->tune.spmv_codes != 0 && A->max_row_nnz <= 64;
    kernel = (short_rows || coded) ? 4 : (A->mean_row_nnz <= 96.0 ? 1 : 2);
  }
  return kernel;
}

int launch_spmv(khip_ctx *ctx, const khip_csr *A, const double *x, double *y, int dot_slot, int64_t row_lo,
                int64_t row_hi, int64_t *wave_cursor, bool finish, const double *dotw, int dot_sq, int64_t hole_lo, int64_t hole_hi) {
  int64_t local_cursor = 0;
  if (!wave_cursor) wave_cursor = &local_cursor;
  // mid-length rows on few diagonals (the 27-point stencil): try the coded column stream once, it decides between the staged
  // and the stream kernel (spmv_kernel_choice).  BEFORE the two-range decision below, which asks spmv_kernel_choice: a build
  // that changed the choice after a hole had been accepted would hand a row range with a hole to a kernel that does not
  // skip it (tests/test_gpu_dist.py: first product on a handle = the boundary launch).
  if (ctx->tune.spmv_kernel == 0 && A->code_state == 0 && A->mean_row_nnz > 12.0 && A->mean_row_nnz <= 64.0 && A->max_row_nnz <= 64 &&
      ctx->tune.spmv_codes && (ctx->tune.spmv_codes != 1 || A->nnz >= ((int64_t)1 << 22)) && ctx->tune.spmv_nt == 0 && !ctx->tune.spmv_fake_gather &&
      !(A->tmpl_id && ctx->tune.spmv_template))
    optional_build(csr_build_codes(ctx, const_cast<khip_csr *>(A)));
  // Two ranges [row_lo, hole_lo) and [hole_hi, row_hi) in one launch (the boundary rows of a row-partitioned product, api.cpp
  // spmv_any): taken by the staged / coded kernels when the first range is whole row blocks; anything else runs them as two launches.
  bool hole = hole_hi > hole_lo && hole_lo >= row_lo && hole_hi <= row_hi;
  if (hole && (hole_lo == row_lo || hole_hi == row_hi)) {        // one of the ranges is empty: an ordinary launch
    if (hole_lo == row_lo) row_lo = hole_hi; else row_hi = hole_lo;
    hole = false;
  }
  if (hole) {
    const bool one_launch = spmv_kernel_choice(ctx, A) == 4 && ((hole_lo - row_lo) & 255) == 0 && ctx->tune.spmv_pipe <= 0 &&
                            ctx->tune.spmv_persist == 0 && ctx->tune.spmv_nt == 0 && ctx->tune.spmv_delta == 0 && ctx->tune.spmv_wide == 0;
    if (!one_launch) {
      KHIP_TRY(launch_spmv(ctx, A, x, y, dot_slot, row_lo, hole_lo, wave_cursor, false, dotw, dot_sq));
      return launch_spmv(ctx, A, x, y, dot_slot, hole_hi, row_hi, wave_cursor, finish, dotw, dot_sq);
    }
  }
  const int nout = dot_sq ? 2 : 1;
  if (dot_sq && spmv_kernel_choice(ctx, A) != 4 && spmv_kernel_choice(ctx, A) != 5 && spmv_kernel_choice(ctx, A) != 1 && spmv_kernel_choice(ctx, A) != 6) { set_error("spmv: the second reduction output needs the staged, stream, wave or template kernel"); return KHIP_ERR_UNSUPPORTED; }
  if (row_hi <= row_lo) {
    if (dot_slot >= 0 && finish) {
      if (*wave_cursor == 0) {                                              // nothing at all: writes 0
        KHIP_TRY(launch_nrm2sq(ctx, 0, x, dot_slot));
        return dot_sq ? launch_nrm2sq(ctx, 0, x, dot_slot + 1) : KHIP_OK;
      }
      return launch_finish(ctx, *wave_cursor, nout, dot_slot);
    }
    return KHIP_OK;
  }
  SpmvArgs a;
  a.rowptr = A->rowptr; a.col = A->col; a.val = A->val;
  a.x = x; a.ghost = A->ghost; a.y = y;
  a.n_owned = A->dist ? A->m : A->n;
  a.row_lo = row_lo; a.row_hi = row_hi;
  a.hole_lo = hole ? hole_lo : INT64_MAX; a.hole_len = hole ? hole_hi - hole_lo : 0;
  a.xcd_remap = ctx->tune.spmv_xcd;
  a.sweep_s = ctx->tune.spmv_sweep_s;
  a.sweep_w = ctx->tune.spmv_sweep_w;
  // y store: non-temporal once y is far beyond the 256 MiB Infinity Cache (512^3: 1 GiB) -- the coded product 2.04 -> 1.96 ms,
  // fused 2.18 -> 2.14, CG +1.3 % (profiles/r04l_sweep_nty.log, r04m); below that the next kernel finds y in the cache and the
  // cacheable store wins (-3 % at 256^3, r01).  Write-through (2) measures like the plain store.
  a.nt_y = ctx->tune.spmv_nty >= 0 ? ctx->tune.spmv_nty : ((size_t)A->m * sizeof(double) >= ((size_t)512 << 20) ? 1 : 0);
  a.dot_early = ctx->tune.spmv_dot_early;
  a.tiles_per_block = 1;
  a.stage_cap = 2048;
  a.stop_seq = ctx->ctl.stop_seq;
  a.seq = ctx->ctl.seq;
  a.dotw = dotw ? dotw : x;
  a.tmpl_id = A->tmpl_id; a.tmpl_off = A->tmpl_off; a.tmpl_val = A->tmpl_val; a.tmpl_cnt = A->tmpl_cnt;
  a.tmpl_T = A->tmpl_T; a.tmpl_K = A->tmpl_K;
  a.dot_sq = dot_sq;
  a.nnz_bound = A->nnz + kPad;
  a.fake_gather = ctx->tune.spmv_fake_gather;
  a.code = nullptr; a.code_tab = nullptr; a.code_T = 0; a.stage_rows = 256; a.max_row = 0;
  a.sell = nullptr; a.sell_off = nullptr; a.sell_units = 0; a.sell_cols = 0; a.sell_c4 = nullptr; a.sell_pair = 0;
  a.blk_pub = ctx->tune.spmv_blk_pub;
  a.stream_nt = ctx->tune.spmv_stream_nt;
  a.dcode = nullptr; a.dbase = nullptr; a.desc_ptr = nullptr; a.desc_pos = nullptr; a.desc_col = nullptr;
  a.blockptr = (ctx->tune.spmv_blockptr && A->blockptr && (row_lo & 255) == 0) ? A->blockptr : nullptr;
  const bool dot = dot_slot >= 0, comp = ctx->tune.compensated != 0, dist = A->dist;
  const bool persist = ctx->tune.spmv_persist != 0;
  const bool nt = ctx->tune.spmv_nt != 0;
  const int64_t nrows = row_hi - row_lo - a.hole_len;

  hipEvent_t ev_stop = nullptr;
  if (ctx->tune.profile_spmv) {
    if (ctx->prof_used + 2 > ctx->prof_events.size()) {
      for (int i = 0; i < 64; ++i) {
        hipEvent_t e;
        KHIP_CHECK_HIP(hipEventCreate(&e));
        ctx->prof_events.push_back(e);
      }
    }
    KHIP_CHECK_HIP(hipEventRecord(ctx->prof_events[ctx->prof_used], ctx->stream));
    ev_stop = ctx->prof_events[ctx->prof_used + 1];
    if (ctx->prof_tags.size() < ctx->prof_events.size() / 2) ctx->prof_tags.resize(ctx->prof_events.size() / 2, 0);
    ctx->prof_tags[ctx->prof_used / 2] = ctx->prof_spmv_tag;
    ctx->prof_used += 2;
  }

  const int kernel = spmv_kernel_choice(ctx, A);               // (the lazy code build that can change it ran at the top of this call)
  if (hole && kernel != 4) { set_error("spmv: a two-range launch reached kernel %d, which takes one range", kernel); return KHIP_ERR_INVALID; }
  unsigned grid = 1;
  RedArgs ra;
  int wpb = kWavesPerBlock;                   // reduction partials one workgroup of the chosen kernel writes
  if (kernel == 6) {
    // wave-private windows: a window that takes a typical 64-row group whole (+ 20 %), in 256-entry steps, at most 2048 entries
    int64_t cap = ctx->tune.spmv_cap > 0 ? ctx->tune.spmv_cap : (int64_t)(64.0 * A->mean_row_nnz * 1.2) + 3;
    cap = (cap + 255) & ~(int64_t)255;
    a.stage_cap = (int)(cap < 256 ? 256 : (cap > 2048 ? 2048 : cap));
    wpb = 1;
    grid = pick_grid(ctx, (nrows + 63) / 64, false);
    if (dot) KHIP_TRY(ensure_reduction_scratch(ctx, *wave_cursor + (int64_t)grid * wpb, nout));
    ra = make_red_args(ctx, dot ? dot_slot : 0); ra.wave_offset = *wave_cursor;
    launch_wave_cfg(ctx, a, ra, grid, dot, comp, dist);
  } else if (kernel == 5) {
    const int rpt = ctx->tune.spmv_tmpl_rows > 0 ? ctx->tune.spmv_tmpl_rows : 1;      // rows per lane: amortises the table load
    grid = pick_grid(ctx, (nrows + (int64_t)kBlock * rpt - 1) / ((int64_t)kBlock * rpt), false);
    if (dot) KHIP_TRY(ensure_reduction_scratch(ctx, *wave_cursor + (int64_t)grid * kWavesPerBlock, nout));
    ra = make_red_args(ctx, dot ? dot_slot : 0); ra.wave_offset = *wave_cursor;
    launch_template_cfg(ctx, a, ra, grid, dot, comp, dist);
  } else if (kernel == 1) {
    int rows = ctx->tune.spmv_rows;
    if (rows * A->mean_row_nnz > 2048.0) {           // keep one LDS pass per row block
      rows = 256;
      while (rows > 32 && rows * A->mean_row_nnz > 2048.0) rows >>= 1;
    }
    if (rows != 256 && rows != 128 && rows != 64 && rows != 32) rows = 256;
    grid = pick_grid(ctx, (nrows + rows - 1) / rows, persist);
    if (dot) KHIP_TRY(ensure_reduction_scratch(ctx, *wave_cursor + (int64_t)grid * kWavesPerBlock, nout));
    ra = make_red_args(ctx, dot ? dot_slot : 0); ra.wave_offset = *wave_cursor;
    const int vec = ctx->tune.spmv_vec == 2 ? 2 : 1;
    // 16-byte-load form (spmv_delta_kernel): with the block-delta column stream (coldelta.hip, built once per handle at the
    // first product that gets here) where that saves bytes, else on the plain int32 columns
    khip_csr *Am = const_cast<khip_csr *>(A);
    const int dl = ctx->tune.spmv_delta;
    const bool wide_ok = !nt && !a.fake_gather && vec != 2 && !persist && A->nnz > 0;
    if (wide_ok && dl && (dl != 1 || A->nnz >= ((int64_t)1 << 22)) && Am->delta_state == 0)
      optional_build(csr_build_delta(ctx, Am, rows));
    const bool delta = wide_ok && dl && Am->delta_state == 1 && row_lo % Am->delta_rows == 0;
    if (delta || (wide_ok && ctx->tune.spmv_wide)) {
      a.stage_rows = delta ? Am->delta_rows : rows;
      if (delta) { a.dcode = Am->dcode; a.dbase = Am->dbase; a.desc_ptr = Am->desc_ptr; a.desc_pos = Am->desc_pos; a.desc_col = Am->desc_col; }
      grid = pick_grid(ctx, (nrows + a.stage_rows - 1) / a.stage_rows, false);
      if (dot) KHIP_TRY(ensure_reduction_scratch(ctx, *wave_cursor + (int64_t)grid * kWavesPerBlock, nout));
      ra = make_red_args(ctx, dot ? dot_slot : 0); ra.wave_offset = *wave_cursor;
      if (!delta) launch_delta_cfg<int32_t>(ctx, a, ra, grid, dot, comp, dist);
      else if (Am->delta_bits == 8) launch_delta_cfg<uint8_t>(ctx, a, ra, grid, dot, comp, dist);
      else launch_delta_cfg<uint16_t>(ctx, a, ra, grid, dot, comp, dist);
      rows = 0;        // launched
    }
#define KHIP_ROWS(R)                                                                              \
  do {                                                                                            \
    if (vec == 2) { if (nt) launch_stream_cfg<R, 2, true>(ctx, a, ra, grid, dot, comp, dist);     \
                    else    launch_stream_cfg<R, 2, false>(ctx, a, ra, grid, dot, comp, dist); }  \
    else          { if (nt) launch_stream_cfg<R, 1, true>(ctx, a, ra, grid, dot, comp, dist);     \
                    else    launch_stream_cfg<R, 1, false>(ctx, a, ra, grid, dot, comp, dist); }  \
  } while (0)
    switch (rows) {
      case 0: break;
      case 256: KHIP_ROWS(256); break;
      case 128: KHIP_ROWS(128); break;
      case 64: KHIP_ROWS(64); break;
      default: KHIP_ROWS(32); break;
    }
#undef KHIP_ROWS
  } else if (kernel == 4) {
    khip_csr *Am = const_cast<khip_csr *>(A);
    // spmv_codes: 1 = coded stream for operators large enough to be bandwidth bound (>= 4 M entries: below that an iteration is
    // latency bound and the table lookup costs ~5 %, profiles/r02_bench_sizes.jsonl), 2 = whenever the operator qualifies,
    // 16 = two-byte codes, 0 = never
    const bool try_codes = ctx->tune.spmv_codes && (ctx->tune.spmv_codes != 1 || A->nnz >= ((int64_t)1 << 22)) && !nt && !a.fake_gather;
    // coded column stream (colcode.hip): built once per handle, at the first product that gets here
    if (try_codes && Am->code_state == 0) optional_build(csr_build_codes(ctx, Am));
    const bool coded = try_codes && Am->code_state == 1;
    // sliced form of the coded operator (csr_build_sell): built once per handle, at the first product that gets here with spmv_sell on
    if (coded && ctx->tune.spmv_sell && Am->code_bits == 8 && Am->sell_state == 0 && ctx->tune.spmv_pipe <= 0) optional_build(csr_build_sell(ctx, Am));
    const bool sliced = coded && ctx->tune.spmv_sell && Am->sell_state == 1 && ctx->tune.spmv_pipe <= 0;
    int rows = ctx->tune.spmv_rows;
    if (rows != 256 && rows != 128 && rows != 64 && rows != 32) rows = 256;
    // the LDS window of the staged / coded kernels holds a whole row block: fewer rows per block for longer rows (three quarters of the
    // lanes idle in the row walk at 27 entries per row).  The sliced form has no window: one row per lane whatever the row length.
    if (!sliced) while (rows > 32 && rows * A->mean_row_nnz > 2048.0) rows >>= 1;
    // row blocks per workgroup: with a fused dot, two -- the double-double tree at a workgroup's end is ~0.1 us of dependent
    // fp64 latency on a 2.5 us lifetime, and it is paid once per workgroup whatever it covered: 2.19 -> 2.10 ms fused at 512^3,
    // the plain product is unchanged and four blocks per workgroup cost more than they save (profiles/r04b_sweep_headline.log)
    const int64_t nrb_all = (nrows + rows - 1) / rows;
    a.tiles_per_block = ctx->tune.spmv_tiles > 0 ? ctx->tune.spmv_tiles : ((dot && nrb_all >= 8192) ? 2 : 1);
    {   // LDS window: the widest row block (+3 for the 4-entry alignment of its start), at most 2048 entries
      int64_t cap = ctx->tune.spmv_cap > 0 ? ctx->tune.spmv_cap : (int64_t)rows * A->max_row_nnz + 3;
      cap = (cap + 3) & ~(int64_t)3;
      a.stage_cap = (int)(cap < 64 ? 64 : (cap > 2048 ? 2048 : cap));
    }
    const int64_t nrb4 = (nrows + rows - 1) / rows;
    grid = pick_grid(ctx, (nrb4 + a.tiles_per_block - 1) / a.tiles_per_block, false);
    if ((int64_t)grid * a.tiles_per_block < nrb4) a.tiles_per_block = (int)((nrb4 + grid - 1) / grid);
    if (dot) KHIP_TRY(ensure_reduction_scratch(ctx, *wave_cursor + (int64_t)grid * kWavesPerBlock, nout));
    ra = make_red_args(ctx, dot ? dot_slot : 0); ra.wave_offset = *wave_cursor;
#define KHIP_STG(R) do { if (nt) launch_stage_cfg<R, true>(ctx, a, ra, grid, dot, comp, dist); \
                         else launch_stage_cfg<R, false>(ctx, a, ra, grid, dot, comp, dist); } while (0)
    if (coded) { a.code = Am->code; a.code_tab = Am->code_tab; a.code_T = Am->code_T; }
    a.stage_rows = rows;
    a.max_row = (int)A->max_row_nnz;
    // ... and of the int32 column stream, for operators that are not coded (or with spmv_codes = 0); the same size rule as the codes:
    // below 4 M entries an iteration is latency bound and the handle keeps one copy of its entries
    const bool try32 = !coded && ctx->tune.spmv_sell && !nt && !a.fake_gather && ctx->tune.spmv_pipe <= 0 && rows == 256 &&
                       (ctx->tune.spmv_codes == 2 || ctx->tune.spmv_sell >= 3 || A->nnz >= ((int64_t)1 << 22));
    if (try32 && Am->sell32_state == 0) optional_build(csr_build_sell32(ctx, Am));
    const bool sliced32 = try32 && Am->sell32_state == 1;
    // software-pipelined form: needs one window per row block
    const bool pipe = ctx->tune.spmv_pipe > 0 && !nt && !a.fake_gather && A->nnz > 0 && A->max_row_nnz >= 1 &&
                      (int64_t)rows * A->max_row_nnz + 3 <= 2048 && ctx->tune.spmv_cap == 0;
    if (pipe) {
      a.tiles_per_block = ctx->tune.spmv_pipe;
      grid = pick_grid(ctx, (nrb4 + a.tiles_per_block - 1) / a.tiles_per_block, false);
      if ((int64_t)grid * a.tiles_per_block < nrb4) a.tiles_per_block = (int)((nrb4 + grid - 1) / grid);
      if (dot) KHIP_TRY(ensure_reduction_scratch(ctx, *wave_cursor + (int64_t)grid * kWavesPerBlock, nout));
      ra = make_red_args(ctx, dot ? dot_slot : 0); ra.wave_offset = *wave_cursor;
      if (!coded) launch_pipe_cfg<int32_t>(ctx, a, ra, grid, dot, comp, dist);
      else if (Am->code_bits == 8) launch_pipe_cfg<uint8_t>(ctx, a, ra, grid, dot, comp, dist);
      else launch_pipe_cfg<uint16_t>(ctx, a, ra, grid, dot, comp, dist);
      rows = 0;      // launched
    } else if (coded && sliced) {
      a.sell = Am->sell; a.sell_off = Am->sell_off; a.sell_units = Am->sell_units; a.sell_cols = 0; a.sell_c4 = Am->sell_c4; a.sell_pair = Am->sell_pair;
      launch_sell_cfg(ctx, a, ra, grid, dot, comp, dist);
      rows = 0;      // launched
    } else if (sliced32) {
      a.sell = Am->sell32; a.sell_off = Am->sell32_off; a.sell_units = Am->sell32_units; a.sell_cols = 1; a.sell_pair = Am->sell32_pair; a.sell_c4 = nullptr;
      launch_sell_cfg(ctx, a, ra, grid, dot, comp, dist);
      rows = 0;      // launched
    } else if (coded) {
      if (Am->code_bits == 8) launch_code_cfg<uint8_t>(ctx, a, ra, grid, dot, comp, dist);
      else launch_code_cfg<uint16_t>(ctx, a, ra, grid, dot, comp, dist);
      rows = 0;      // launched
    }
    switch (rows) {
      case 0: break;
      case 256: KHIP_STG(256); break;
      case 128: KHIP_STG(128); break;
      case 64: KHIP_STG(64); break;
      default: KHIP_STG(32); break;
    }
#undef KHIP_STG
  } else if (kernel == 3) {
    int L = ctx->tune.spmv_lanes;
    if (L == 0) {                                    // smallest power of two covering a typical row
      L = 4;
      const double target = A->max_row_nnz <= 64 ? (double)A->max_row_nnz : A->mean_row_nnz;
      while (L < 64 && L < target) L <<= 1;
    }
#define KHIP_ORD(LL, RPG)                                                                     \
  do {                                                                                        \
    constexpr int rpb = (64 / LL) * RPG * kWavesPerBlock;                                     \
    grid = pick_grid(ctx, (nrows + rpb - 1) / rpb, persist);                                  \
    if (dot) KHIP_TRY(ensure_reduction_scratch(ctx, *wave_cursor + (int64_t)grid * kWavesPerBlock, nout));                                \
    ra = make_red_args(ctx, dot ? dot_slot : 0); ra.wave_offset = *wave_cursor;                                              \
    if (nt) launch_ordered_cfg<LL, RPG, true>(ctx, a, ra, grid, dot, comp, dist);             \
    else launch_ordered_cfg<LL, RPG, false>(ctx, a, ra, grid, dot, comp, dist);               \
  } while (0)
    switch (L) {
      case 4: KHIP_ORD(4, 4); break;
      case 8: KHIP_ORD(8, 4); break;
      case 16: KHIP_ORD(16, 4); break;
      case 32: KHIP_ORD(32, 2); break;
      default: KHIP_ORD(64, 1); break;
    }
#undef KHIP_ORD
  } else {
    int lpr = ctx->tune.spmv_lanes;
    if (lpr == 0) {
      lpr = 4;
      while (lpr < 64 && lpr * 2 <= A->mean_row_nnz) lpr <<= 1;
    }
    const int rpb = kBlock / lpr;
    grid = pick_grid(ctx, (nrows + rpb - 1) / rpb, persist);
    if (dot) KHIP_TRY(ensure_reduction_scratch(ctx, *wave_cursor + (int64_t)grid * kWavesPerBlock, nout));
    ra = make_red_args(ctx, dot ? dot_slot : 0); ra.wave_offset = *wave_cursor;
    switch (lpr) {
      case 4: launch_vector_cfg<4>(ctx, a, ra, grid, dot, comp, dist); break;
      case 8: launch_vector_cfg<8>(ctx, a, ra, grid, dot, comp, dist); break;
      case 16: launch_vector_cfg<16>(ctx, a, ra, grid, dot, comp, dist); break;
      case 32: launch_vector_cfg<32>(ctx, a, ra, grid, dot, comp, dist); break;
      default: launch_vector_cfg<64>(ctx, a, ra, grid, dot, comp, dist); break;
    }
  }
  KHIP_CHECK_HIP(hipGetLastError());
  // the profiling bracket closes right behind the SpMV kernel: the tiny finish kernel of a fused dot is another kernel
  if (ev_stop) KHIP_CHECK_HIP(hipEventRecord(ev_stop, ctx->stream));
  if (dot) {
    *wave_cursor += (int64_t)grid * wpb;
    if (finish) KHIP_TRY(launch_finish(ctx, *wave_cursor, nout, dot_slot));
  }
  return KHIP_OK;
}

}  // namespace khip
