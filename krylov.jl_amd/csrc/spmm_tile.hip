// spmm_tile.hip -- SpMM for p = 8, 16 or 32 right-hand sides: wave-private panel-row windows filled by LDS-DMA.
//
//   Y(m x p, row-major) = A * X(n x p, row-major), p = 4 L, L = 2, 4, 8   (mul!(W, A, P), src/block_gmres.jl:242)
// (numbers in this header: p = 16)
//
// Why another kernel.  spmm_window_kernel (csr_aux.hip) shares one 44 KB window between the four waves of a
// persistent workgroup: two workgroups per CU, every row group one memory round trip behind a workgroup barrier, and a
// window of 32 CONSECUTIVE rows of the 27-point operator holds 9 x 34 panel rows -- each panel row crosses L2 -> LDS 9.6
// times (16.4 GB of L2 misses for 5.9 GB algorithmic, 2.2 ms).  Here
//   * a row group (32 rows) belongs to ONE wave: no workgroup barrier anywhere, 8+ independent waves per CU, each either
//     waiting for its loads or computing -- the hardware interleaves them;
//   * the group's distinct panel rows land in the wave's LDS window by global_load_lds_dwordx4 (1 KiB per instruction, no
//     staging registers, no ds_write);
//   * the 32 rows of a group need not be consecutive: on operators whose pattern is a structured grid the groups are
//     4 x 4 x 2 grid tiles (6 x 6 x 4 = 144 panel rows instead of 306: 18 KB windows, 4.5 instead of 9.6 L2 -> LDS passes
//     per panel row), found from the row order alone (spmm_tile_build tries the identity order and the tile order and keeps
//     the one with fewer distinct columns per group); the groups are dealt to the XCDs in eight contiguous runs so that
//     neighbouring tiles share an L2;
//   * a lane owns FOUR panel columns of a row (4 lanes per row, 16 rows per wave pass): per entry one broadcast of
//     (val, slot) inside a quad (v_mov_dpp quad_perm) and two ds_read_b128 -- half the per-entry VALU work of two columns per lane;
//   * everything a group needs besides val sits in one contiguous record: {row, first entry, length} per row, the list of
//     distinct columns, one byte per nonzero (position of its column in the list).
// Arithmetic: per row and column a rounded multiply and a rounded add in stored order, as in spmm_kernel / spmm2_kernel /
// spmm_window_kernel and p SpMVs => Y is bit-identical to all of them (tests/test_gpu_block.py::test_spmm_tile_*).
// Groups with a row longer than 32 entries or more distinct columns than the window takes are flagged at build time and
// go down a direct-gather path inside the same kernel.
#include <algorithm>
#include <type_traits>
#include <vector>

#include "spmv_common.hpp"

namespace khip {

#ifndef KHIP_TILE_ROWS
#define KHIP_TILE_ROWS 32           // rows per group.  64 (an experiment build, -DKHIP_TILE_ROWS=64): 4 x 4 x 4 grid tiles, 216 instead of 2 x 144 panel rows
#endif                              // per 64 rows of the 27-point operator -- the synthetic twin's floor is 13 % lower for them (profiles/r05b_spmm_floor.log)
constexpr int kTileR = KHIP_TILE_ROWS;   // rows per group (p = 16: kTileR / 16 passes of 16 rows x 4 lanes)
static_assert(kTileR == 32 || kTileR == 64, "spmm_tile: groups of 32 or 64 rows");
constexpr int kTileLen = 32;        // entries per row the register path takes
constexpr int kTileCapMax = 256;    // distinct panel rows per group (one-byte slots); the window is cap x 128 B of LDS
constexpr int kTileDescBytes = kTileR * 16;
constexpr int kTileSlotBytes = kTileR * kTileLen;

struct TileArgs {
  const char *meta;       // [groups] records: int4 {row, start, len, aux}[32] | int32 list[cap] | uint8 slot[32][32]
  int64_t groups, per_xcd;
  int64_t runs, runs_per_xcd;   // sliding windows (run_len > 0): the groups form runs of run_len consecutive records, a wave walks whole runs
  int run_len;
  int ahead;              // 1 (two-wave kernel, sliding windows): the copies of group g + 1 are issued before the products of g where its record allows it
  int dbuf;               // 1: two windows per wave, the copies of the next group land while this group's products run (LDS: 2 x cap x 32 L bytes)
  int cap;                // list entries per group (multiple of 8)
  int stride;             // bytes per record
  int exp;                // tuning experiments, WRONG results: 1 no panel-row copies, 2 no products, 4 no (val, slot) loads, 8 groups dealt round-robin to the XCDs
  int gshift;             // log2 of the bytes of a panel row in HBM: 5 + log2 L, or more when the launch covers a column slice of wider panels
  int coff;               // byte offset of that slice in a panel row
};

typedef double dbl2u __attribute__((ext_vector_type(2), aligned(8)));
typedef __attribute__((address_space(3))) void lds_void;

// Value of lane K of every group of L consecutive lanes (L = 2, 4, 8) by DPP moves; old = 0 + bound_ctrl: no register to initialise.
template <int L, int K>
__device__ __forceinline__ int tile_bcast(int v) {
  static_assert(L == 2 || L == 4 || L == 8, "tile_bcast: L = 2, 4, 8");
  if (L == 2) {
    constexpr int QP = K | (K << 2) | ((K + 2) << 4) | ((K + 2) << 6);                  // quad_perm [K, K, K+2, K+2]
    return __builtin_amdgcn_update_dpp(0, v, QP, 0xF, 0xF, true);
  }
  constexpr int Q = K & 3, QP = Q | (Q << 2) | (Q << 4) | (Q << 6);                     // quad_perm [Q, Q, Q, Q]
  int t = __builtin_amdgcn_update_dpp(0, v, QP, 0xF, 0xF, true);
  if (L == 8)       // the other quad of the octet takes the value across row_half_mirror (lane i <- lane 7 - i)
    t = __builtin_amdgcn_update_dpp(t, t, 0x141, 0xF, K < 4 ? 0xA : 0x5, false);
  return t;
}
template <int L, int K>
__device__ __forceinline__ double tile_bcast(double v) {
  return __hiloint2double(tile_bcast<L, K>(__double2hiint(v)), tile_bcast<L, K>(__double2loint(v)));
}

// Shapes for p = 4 L right-hand sides: L lanes per matrix row, FOUR panel columns per lane (pieces c and L + c of the 2 L
// 16-byte pieces of a panel row), 64 / L rows per wave pass, 32 rows per group.
template <int L>
struct TileShape {
  static constexpr int RP = 64 / L;            // rows per pass
  static constexpr int NPASS = kTileR / RP;    // L = 2: 1, 4: 2, 8: 4
  static constexpr int F = 16 / L;             // 16-byte val loads per lane and pass: entries 2 c, 2 c + 1 (+ 2 L f)
  static constexpr int SW = 8 / L;             // slot words per lane and pass: bytes 4 SW c .. 4 SW (c + 1) - 1 of the row's 32
  static constexpr int ROWB = 32 * L;          // bytes of a panel row
  static constexpr int SHIFT = L == 2 ? 6 : (L == 4 ? 7 : 8);
  static constexpr int EPI = 32 / L;           // list entries one LDS-DMA instruction copies (1 KiB)
};

// One entry step of a pass: (val, slot) of entry T spread over the row's L lanes, the two 16-byte pieces of the panel row out
// of the window, four rounded products and four rounded adds.  xa / xb_: the lane's window base for its first and second
// piece.  Which half of the panel row a lane group reads FIRST alternates with bit 1 of the group number: a ds_read_b128 is
// served in four groups of 16 lanes, a row's lanes touch a quarter (L = 4) of the 64 banks, and which one is decided by
// (parity of the slot, half of the row) -- with every group reading the same half first, the rows of a service group shared
// two bank ranges (45 % of the LDS cycles were conflicts, profiles/r03_spmm_tile_pmc.log); alternating halves gives them the
// four ranges whenever their slot parities alternate too (consecutive rows of a grid tile).
template <int L, int T, bool MASK>
__device__ __forceinline__ void tile_entry(const dbl2 (&v)[TileShape<L>::F], const int (&sw)[TileShape<L>::SW], const char *xa,
                                           const char *xb_, int len, double (&acc)[4]) {
  using S = TileShape<L>;
  constexpr int f = T / (2 * L), e = T % (2 * L);
  const double vv = tile_bcast<L, e / 2>((e & 1) ? v[f].y : v[f].x);
  constexpr int bpl = 4 * S::SW;                                     // slot bytes per lane
  const int word = tile_bcast<L, T / bpl>(sw[(T % bpl) / 4]);
  const int off = (int)(((unsigned)word >> (8 * (T & 3))) & 0xffu) << S::SHIFT;
  const dbl2 x0 = *reinterpret_cast<const dbl2 *>(xa + off);
  const dbl2 x1 = *reinterpret_cast<const dbl2 *>(xb_ + off);
  const double p0 = vv * x0.x, p1 = vv * x0.y, p2 = vv * x1.x, p3 = vv * x1.y;
  if (!MASK || T < len) {
    acc[0] = acc[0] + p0;
    acc[1] = acc[1] + p1;
    acc[2] = acc[2] + p2;
    acc[3] = acc[3] + p3;
  }
}
template <int L, int B, bool MASK>       // entries 8 B .. 8 B + 7; n: wave-uniform number of them that any row still has (1..8)
__device__ __forceinline__ void tile_batch(const dbl2 (&v)[TileShape<L>::F], const int (&sw)[TileShape<L>::SW], const char *xa,
                                           const char *xb_, int len, int n, double (&acc)[4]) {
  tile_entry<L, 8 * B + 0, MASK>(v, sw, xa, xb_, len, acc);
  if (n > 1) tile_entry<L, 8 * B + 1, MASK>(v, sw, xa, xb_, len, acc);
  if (n > 2) tile_entry<L, 8 * B + 2, MASK>(v, sw, xa, xb_, len, acc);
  if (n > 3) tile_entry<L, 8 * B + 3, MASK>(v, sw, xa, xb_, len, acc);
  if (n > 4) tile_entry<L, 8 * B + 4, MASK>(v, sw, xa, xb_, len, acc);
  if (n > 5) tile_entry<L, 8 * B + 5, MASK>(v, sw, xa, xb_, len, acc);
  if (n > 6) tile_entry<L, 8 * B + 6, MASK>(v, sw, xa, xb_, len, acc);
  if (n > 7) tile_entry<L, 8 * B + 7, MASK>(v, sw, xa, xb_, len, acc);
}
// CHUNKS (round 6).  The entry-by-entry form above exposes one LDS round trip per entry: two ds_read_b128, a wait, four products,
// four adds, a scalar branch (profiles/r06c: ~170 cycles per entry and wave, ~2 us per group -- a quarter of the 8 us a
// workgroup spends per group).  Where all rows of a pass have the same length (wave-uniform n, no masks) the entries are taken
// C at a time: the C (val, slot) broadcasts, then all 2 C window reads, then the 4 C products and adds IN STORED ORDER -- the same
// rounded operations per row and column, so Y stays bit-identical; one LDS round trip per C entries.
// Measured (profiles/r06e_spmm_chunk_ab.log, r06t_spmm_grid_attribution.log): the chunks cost registers (120 -> 141 VGPRs at NL = 3: 6
// instead of 7 workgroups per CU) and the products they speed up are hidden anyway (the kernel is 1.6 % faster with NO products at all,
// r06m_spmm_phases.log): chunks of 4 run 2-4 % SLOWER than the entry-by-entry loop on the same grid.  The library is built with 1 (off);
// -DKHIP_TILE_CHUNK=2 / 4 builds the variants (tools/spmm_chunk_ab.sh).
#ifndef KHIP_TILE_CHUNK
#define KHIP_TILE_CHUNK 1
#endif
template <int L, int T0, int C>
__device__ __forceinline__ void tile_chunk(const dbl2 (&v)[TileShape<L>::F], const int (&sw)[TileShape<L>::SW], const char *xa,
                                           const char *xb_, double (&acc)[4]) {
  using S = TileShape<L>;
  constexpr int bpl = 4 * S::SW;
  double vv[C];
  dbl2 x0[C], x1[C];
  int off[C];
  auto one = [&](auto kc) {
    constexpr int k = decltype(kc)::value, T = T0 + k, f = T / (2 * L), e = T % (2 * L);
    vv[k] = tile_bcast<L, e / 2>((e & 1) ? v[f].y : v[f].x);
    const int word = tile_bcast<L, T / bpl>(sw[(T % bpl) / 4]);
    off[k] = (int)(((unsigned)word >> (8 * (T & 3))) & 0xffu) << S::SHIFT;
  };
  one(std::integral_constant<int, 0>{});
  if constexpr (C > 1) one(std::integral_constant<int, 1>{});
  if constexpr (C > 2) one(std::integral_constant<int, 2>{});
  if constexpr (C > 3) one(std::integral_constant<int, 3>{});
#pragma unroll
  for (int k = 0; k < C; ++k) {
    x0[k] = *reinterpret_cast<const dbl2 *>(xa + off[k]);
    x1[k] = *reinterpret_cast<const dbl2 *>(xb_ + off[k]);
  }
#pragma unroll
  for (int k = 0; k < C; ++k) {
    const double p0 = vv[k] * x0[k].x, p1 = vv[k] * x0[k].y, p2 = vv[k] * x1[k].x, p3 = vv[k] * x1[k].y;
    acc[0] = acc[0] + p0;
    acc[1] = acc[1] + p1;
    acc[2] = acc[2] + p2;
    acc[3] = acc[3] + p3;
  }
}
template <int L, int Q>        // entries C Q .. of a pass whose rows all have n entries (wave-uniform): whole chunks, then the rest of the last one
__device__ __forceinline__ void tile_chunks_from(const dbl2 (&v)[TileShape<L>::F], const int (&sw)[TileShape<L>::SW], const char *xa,
                                                 const char *xb_, int n, double (&acc)[4]) {
  constexpr int C = KHIP_TILE_CHUNK, T0 = C * Q;
  if constexpr (T0 < kTileLen) {
    if (n >= T0 + C) {
      tile_chunk<L, T0, C>(v, sw, xa, xb_, acc);
      tile_chunks_from<L, Q + 1>(v, sw, xa, xb_, n, acc);
    } else {
      const int r = n - T0;
      if (r == 1) tile_chunk<L, T0, 1>(v, sw, xa, xb_, acc);
      if constexpr (C > 2) { if (r == 2) tile_chunk<L, T0, 2>(v, sw, xa, xb_, acc); }
      if constexpr (C > 3) { if (r == 3) tile_chunk<L, T0, 3>(v, sw, xa, xb_, acc); }
    }
  }
}
template <int L, bool MASK, bool CHUNKED = false>        // CHUNKED: the two-wave kernel (the one-wave kernel keeps its 4 waves per SIMD with the entry-by-entry loop)
__device__ __forceinline__ void tile_rows(const dbl2 (&v)[TileShape<L>::F], const int (&sw)[TileShape<L>::SW], const char *xa,
                                          const char *xb_, int len, int nmax, double (&acc)[4]) {
  if constexpr (CHUNKED && !MASK && KHIP_TILE_CHUNK > 1 && L == 4) {       // p = 16 (and its column slices) only: at p = 8 the chunks cost 17 % (0.96 -> 1.12 ms, profiles/r06e_spmm_chunk_ab.log)
    tile_chunks_from<L, 0>(v, sw, xa, xb_, nmax, acc);
    return;
  }
  if (nmax > 0) tile_batch<L, 0, MASK>(v, sw, xa, xb_, len, nmax < 8 ? nmax : 8, acc);
  if (nmax > 8) tile_batch<L, 1, MASK>(v, sw, xa, xb_, len, nmax < 16 ? nmax - 8 : 8, acc);
  if (nmax > 16) tile_batch<L, 2, MASK>(v, sw, xa, xb_, len, nmax < 24 ? nmax - 16 : 8, acc);
  if (nmax > 24) tile_batch<L, 3, MASK>(v, sw, xa, xb_, len, nmax - 24, acc);
}

template <int L, int NL>
struct TileRec {            // what a lane holds of a group's record: its row descriptors (one per pass) and its share of the list
  int4v d[TileShape<L>::NPASS];
  int lw[NL];
};
template <int L>
struct TileEnt {            // ... and of the (val, slot) stream of its rows
  dbl2 v[TileShape<L>::NPASS][TileShape<L>::F];
  int sw[TileShape<L>::NPASS][TileShape<L>::SW];
};

// Persistent waves, one per workgroup, wave i takes the groups i, i + G, i + 2 G, ...  Three stages in flight per wave:
// the record of group g + 2 G, the (val, slot) entries of g + G (they need its record), the panel rows of g (LDS-DMA; they need
// its list).  After issuing all three the wave waits for the DMA alone (vmcnt = the loads issued after it), so that per
// group ONE latency -- that of panel rows, mostly L2 / Infinity-Cache hits -- is exposed, and the HBM streams (val, records)
// run a whole group ahead.  (The one-shot form, one group per wave launch, spent 8.8 us per wave, 63 % of it waiting on
// two dependent round trips: 1.35-1.5 ms; profiles/r03_spmm_tile_sweep.log.)
template <int L, bool DIST, int NL, bool NT>      // p = 4 L; NL = ceil(cap / 64): list words per lane; NT: non-temporal hints on the streams
__global__ __launch_bounds__(64) void spmm_tile_kernel(SpmvArgs a, TileArgs w) {
  using S = TileShape<L>;
  extern __shared__ dbl2 tile_win[];                 // [cap][2 L]: the group's distinct panel rows
  typedef __attribute__((address_space(3))) char lds_char;
  const int lane = threadIdx.x, sub = lane / L, c = lane % L;
  // Wave b runs on XCD b % 8 (round-robin dispatch of the workgroups).  XCD x takes the x-th of eight contiguous runs of
  // groups and its waves walk that run side by side, so that the groups in flight on an XCD are neighbours and share
  // panel rows through its L2 (exp & 8: plain round-robin over all waves instead).
  const int64_t last = w.groups - 1;
  int64_t G = gridDim.x, g = blockIdx.x, gend = w.groups;
  if (!(w.exp & 8) && (gridDim.x & 7) == 0) {
    const int64_t x = blockIdx.x & 7;
    G = gridDim.x >> 3;
    g = x * w.per_xcd + (blockIdx.x >> 3);
    gend = (x + 1) * w.per_xcd < w.groups ? (x + 1) * w.per_xcd : w.groups;
  }
  // SLIDING WINDOWS (w.run_len > 0): the records form runs of run_len groups that follow each other along the slowest grid
  // direction (or are consecutive row blocks); a wave walks a whole run and KEEPS its window from one group to the next -- the
  // record of a group assigns the panel rows it shares with its predecessor the slots they already sit in and lists in a bit
  // mask the octets of slots that hold new rows: only those are copied.  4 x 4 x 2 tiles of a 27-point grid share two of
  // their four planes with the next tile: 72 instead of 144 panel rows per group cross L2 -> LDS.  XCD x takes the x-th
  // eighth of the runs; its waves walk neighbouring runs side by side.
  const bool slide = w.run_len > 0;
  int64_t run = 0, run_end = 0;
  int tpos = 0;
  if (slide) {
    const int64_t x = blockIdx.x & 7;
    G = (gridDim.x & 7) == 0 ? (gridDim.x >> 3) : gridDim.x;
    run = (gridDim.x & 7) == 0 ? x * w.runs_per_xcd + (blockIdx.x >> 3) : blockIdx.x;
    run_end = (gridDim.x & 7) == 0 ? ((x + 1) * w.runs_per_xcd < w.runs ? (x + 1) * w.runs_per_xcd : w.runs) : w.runs;
    g = run * w.run_len;
    gend = run_end * w.run_len;
  }
  // the group after (rr, tt) in this wave's sequence
  auto next_of = [&](int64_t &rr, int &tt) -> int64_t {
    if (!slide) return 0;
    if (++tt == w.run_len) { tt = 0; rr += G; }
    return rr * w.run_len + tt;
  };
  if (g >= gend) return;
  const int hq = (sub & 2) ? 16 * L : 0;             // lane groups with bit 1 set read the second half of a panel row first
  const char *xa0 = reinterpret_cast<const char *>(tile_win) + 16 * c + hq;
  const char *xb0 = reinterpret_cast<const char *>(tile_win) + 16 * c + (16 * L - hq);
  const int64_t vlast = a.nnz_bound - 2;
  const int slot_off = kTileDescBytes + 4 * w.cap;

  auto load_rec = [&](int64_t gg, TileRec<L, NL> &r) {
    const char *rec = w.meta + (gg < last ? gg : last) * (int64_t)w.stride;
    const int4v *desc = reinterpret_cast<const int4v *>(rec);
    const int32_t *lst = reinterpret_cast<const int32_t *>(rec + kTileDescBytes);
#pragma unroll
    for (int q = 0; q < S::NPASS; ++q) r.d[q] = ld<NT>(desc + q * S::RP + sub);   // records, entries and Y are streams
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int q = lane + 64 * j;
      r.lw[j] = ld<NT>(lst + (q < w.cap ? q : w.cap - 1));
    }
  };
  // the (val, slot) stream of the lane's rows: entries 2 c, 2 c + 1 (+ 2 L f) and the slot bytes 4 SW c .. 4 SW (c + 1) - 1
  auto load_ent = [&](int64_t gg, const TileRec<L, NL> &r, TileEnt<L> &e) {
    const char *slots = w.meta + (gg < last ? gg : last) * (int64_t)w.stride + slot_off;
#pragma unroll
    for (int q = 0; q < S::NPASS; ++q) {
#pragma unroll
      for (int f = 0; f < S::F; ++f) {
        int64_t i0 = (int64_t)r.d[q].y + 2 * c + 2 * L * f;
        i0 = i0 < vlast ? i0 : vlast;
        if (w.exp & 4) i0 = 2 * c + 2 * L * f;
        e.v[q][f] = NT ? __builtin_nontemporal_load(reinterpret_cast<const dbl2u *>(a.val + i0)) : *reinterpret_cast<const dbl2u *>(a.val + i0);
      }
      const char *sp = slots + (q * S::RP + sub) * kTileLen + 4 * S::SW * c;
      if (S::SW == 1) {
        e.sw[q][0] = ld<NT>(reinterpret_cast<const int *>(sp));
      } else if (S::SW == 2) {
        const int2v t = ld<NT>(reinterpret_cast<const int2v *>(sp));
        e.sw[q][0] = t.x; e.sw[q][S::SW - 1] = t.y;
      } else {
        const int4v t = ld<NT>(reinterpret_cast<const int4v *>(sp));
        e.sw[q][0] = t.x; e.sw[q][1 % S::SW] = t.y; e.sw[q][2 % S::SW] = t.z; e.sw[q][3 % S::SW] = t.w;
      }
    }
  };
  // panel rows -> LDS: instruction wq copies list entries EPI wq .. EPI (wq + 1) - 1, lane i the 16-byte piece i % (2 L) of
  // entry i / (2 L).  The copy is inline asm on purpose: with the builtin, hipcc tracks the LDS-DMA as a pending LDS write and
  // puts a vmcnt wait that also drains the prefetches of the next groups behind every ds_read of the product loop (seen in
  // the ISA); an asm VMEM instruction is outside its bookkeeping, which is safe here: the copies are OLDER than every load the
  // compiler counts in this iteration, so its counted waits stay sufficient, and the wait for the copies themselves is the
  // explicit vmcnt below.  M0 (the LDS destination base) is compiler-reserved: saved and restored per statement.
  const unsigned win_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(lds_char *)tile_win);
  auto issue_dma = [&](const TileRec<L, NL> &r, unsigned mask, unsigned wbase) {     // mask: bit o set = the slots 8 o .. 8 o + 7 hold rows that are not in the window yet; wbase: byte offset of the window (double buffering)
    constexpr int PER = 64 / S::EPI;                 // instructions per list word
    int col[NL * PER];
#pragma unroll
    for (int j = 0; j < NL; ++j) {                   // all the lane exchanges first: one lgkmcnt wait, then the copies back to back
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        col[j * PER + u] = __builtin_amdgcn_ds_bpermute(4 * (u * S::EPI + lane / (2 * L)), r.lw[j]);
        if (w.exp & 16) col[j * PER + u] = S::EPI * (j * PER + u) + lane / (2 * L);
      }
    }
#pragma unroll
    for (int wq = 0; wq < NL * PER; ++wq) {
      constexpr unsigned OCT = S::EPI >= 8 ? (1u << (S::EPI / 8)) - 1u : 1u;      // octets one instruction covers (a part of one when EPI = 4)
      if (S::EPI * wq < w.cap && !(w.exp & 1) && ((mask >> ((S::EPI * wq) >> 3)) & OCT) != 0) {
        const char *src = reinterpret_cast<const char *>(a.x);
        uint64_t rr = (unsigned)col[wq];
        if (DIST) {
          const bool own = (int64_t)col[wq] < a.n_owned;
          src = own ? src : reinterpret_cast<const char *>(a.ghost);
          rr = own ? rr : rr - (uint64_t)a.n_owned;
        }
        const char *gsrc = src + (rr << w.gshift) + w.coff + 16 * (lane % (2 * L));
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(win_lds + wbase + 1024u * (unsigned)wq));   // wave-uniform by construction; the "s" operand needs the compiler to know it
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
      }
    }
  };
  auto products = [&](const TileRec<L, NL> &r, const TileEnt<L> &e, unsigned wbase) {
    const char *xa = xa0 + wbase, *xb_ = xb0 + wbase;
#pragma unroll
    for (int q = 0; q < S::NPASS; ++q) {
      const int4v &d = r.d[q];
      const int len = d.z;
      const int len0 = __builtin_amdgcn_readfirstlane(len);
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      if (w.exp & 2) {
        acc[0] = e.v[q][0].x + e.v[q][S::F - 1].y + (double)e.sw[q][0];
      } else if (__ballot(len != len0) == 0) {
        tile_rows<L, false>(e.v[q], e.sw[q], xa, xb_, len, len0, acc);
      } else {
        int nmax = len;
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) { const int o = __shfl_xor(nmax, sft); nmax = o > nmax ? o : nmax; }
        nmax = __builtin_amdgcn_readfirstlane(nmax);
        tile_rows<L, true>(e.v[q], e.sw[q], xa, xb_, len, nmax, acc);
      }
      if (d.x >= 0) {
        double *yr = a.y + ((int64_t)d.x << (w.gshift - 3)) + (w.coff >> 3) + 2 * c;
        if (w.exp & 32) {
          // experiment (results unchanged): write-through stores (sc1) -- they leave no line in the XCD's L2 (MI355X_MICROARCH.md:
          // "sc1 stores DROP it"), so Y does not push the panel rows of the next layer of tiles out of the 4 MB
          const dbl2 v0{acc[0], acc[1]}, v1{acc[2], acc[3]};
          const double *p0 = yr + (hq >> 3), *p1 = yr + 2 * L - (hq >> 3);
          asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %2, %3, off sc1\n\ts_nop 1" :: "v"(p0), "v"(v0), "v"(p1), "v"(v1) : "memory");
        } else if (NT || (w.exp & 64)) {          // exp & 64: non-temporal stores of Y alone (results unchanged)
          __builtin_nontemporal_store(dbl2{acc[0], acc[1]}, reinterpret_cast<dbl2 *>(yr + (hq >> 3)));          // the half this lane group read first
          __builtin_nontemporal_store(dbl2{acc[2], acc[3]}, reinterpret_cast<dbl2 *>(yr + 2 * L - (hq >> 3)));
        } else {
          *reinterpret_cast<dbl2 *>(yr + (hq >> 3)) = dbl2{acc[0], acc[1]};
          *reinterpret_cast<dbl2 *>(yr + 2 * L - (hq >> 3)) = dbl2{acc[2], acc[3]};
        }
      }
    }
  };

  TileRec<L, NL> r0, r1, r2;
  TileEnt<L> e0, e1;
  int64_t run1 = run, run2;
  int t1 = tpos, t2;
  int64_t g1 = slide ? next_of(run1, t1) : g + G;
  run2 = run1; t2 = t1;
  int64_t g2 = slide ? next_of(run2, t2) : g + 2 * G;
  load_rec(g, r0);
  load_rec(g1, r1);
  load_ent(g, r0, e0);
  __builtin_amdgcn_s_waitcnt(0x0F70);                          // vmcnt(0): the loop starts with nothing in flight
  if (w.dbuf) {
    // DOUBLE-BUFFERED WINDOWS (round 4).  With one window a wave cannot start the copies of group g + 1 before the products of
    // g have read the window empty: copy latency and ~2 us of products alternate inside every wave, and 7 waves per CU do not
    // cover for each other (TCP misses in flight ~80 lines per CU, profiles/r03c_spmm_tile_pmc_persistent.log).  With two
    // windows the iteration is: everything outstanding has landed (it is all needed now) -> issue the copies of g + 1 into the
    // other window, its (val, slot) loads and the record of g + 2 -> products of g while those fly.
    const unsigned wbytes = (unsigned)w.cap * 32u * (unsigned)L;
    unsigned cur = 0;
    {
      const bool d0 = __builtin_amdgcn_readfirstlane(__shfl(r0.d[0].w, L)) != 0;
      if (!d0) issue_dma(r0, 0xffffffffu, 0u);
    }
    for (;;) {
      __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0): copies of g, entries of g, record of g + 1
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      const bool direct = __builtin_amdgcn_readfirstlane(__shfl(r0.d[0].w, L)) != 0;
      const bool more = g1 < gend;
      if (more) {
        const bool d1 = __builtin_amdgcn_readfirstlane(__shfl(r1.d[0].w, L)) != 0;
        if (!d1) issue_dma(r1, 0xffffffffu, (cur ^ 1u) * wbytes);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        load_ent(g1, r1, e1);
        load_rec(g2, r2);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!direct) products(r0, e0, cur * wbytes);
      g = g1; g1 = g2; g2 = g2 + G;
      if (!more) break;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // every window read of this group has returned before the window is filled again
      r0 = r1; e0 = e1; r1 = r2;
      cur ^= 1u;
    }
    return;
  }
  for (;;) {
    // aux of row slot 1 = the group's flag (row slot 1 is lane L of pass 0), of row slot 2 = the octets to copy
    const bool direct = __builtin_amdgcn_readfirstlane(__shfl(r0.d[0].w, L)) != 0;
    const unsigned mask = slide ? (unsigned)__builtin_amdgcn_readfirstlane(__shfl(r0.d[0].w, 2 * L)) : 0xffffffffu;
    if (!direct) issue_dma(r0, mask, 0u);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    load_ent(g1, r1, e1);
    load_rec(g2, r2);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (!direct) {
      // the DMA is older than the NPASS (F + 1) + NPASS + NL loads just issued: wait for it alone (nothing but the issuing
      // wave's vmcnt orders a ds_read behind an LDS-DMA).  The builtin, not inline asm: hipcc's own wait-count bookkeeping sees
      // it and does not add vmcnt(0) at the first use of the entries loaded one iteration ago; gfx9 encoding: vmcnt in bits
      // 3:0 and 15:14, expcnt 6:4, lgkmcnt 11:8.
      constexpr int N = S::NPASS * (S::F + 1) + S::NPASS + NL;
      __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      products(r0, e0, 0u);
    }                                                        // (flagged groups: spmm_tile_direct_kernel, launched next)
    if (slide) { g = g1; g1 = g2; g2 = next_of(run2, t2); } else { g += G; g1 = g + G; g2 = g + 2 * G; }
    if (g >= gend) break;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // every window read of this group has returned before the next DMA
    r0 = r1; e0 = e1; r1 = r2;
  }
}

// TWO WAVES PER WINDOW (round 4, `spmm_tile_pair`).  The window of a group is what limits the residency (18 KB at 144 panel rows:
// 7 one-wave workgroups per CU, 1.75 waves per SIMD), and a wave alternates between waiting for its copies and ~2 us of products
// (304 M VALU wave-instructions per launch).  Here a workgroup is two waves sharing ONE window: each issues half of the copies,
// they meet at a barrier when the copies have landed, and each runs the products of half of the group's row passes (L = 4: one
// pass of 16 rows each) -- twice the waves per CU on the same LDS, half the serial work per wave and group.  The record / entry
// prefetch pipeline is per wave as before.  Same arithmetic, same order per row: Y bit-identical.
// AHEAD (round 6) is a separate instantiation: the look-ahead loop keeps three records and two entry sets live across a
// barrier and would cost the round-4 loop its fourth wave per SIMD if both sat in one kernel (161 instead of 120 VGPRs at NL = 3).
// experiment build (-DKHIP_TILE2_NT=1): the two-wave kernel's matrix-side streams (records, entries) and its Y stores carry the
// non-temporal hint, so that they do not push the panel rows out of the Infinity Cache (tools/spmm_chunk_ab.sh builds it as build_nt)
#ifdef KHIP_TILE2_NT
#define KHIP_T2_LD(ptr) __builtin_nontemporal_load(ptr)
#define KHIP_T2_ST(ptr, val) __builtin_nontemporal_store(val, ptr)
#else
#define KHIP_T2_LD(ptr) (*(ptr))
#define KHIP_T2_ST(ptr, val) (*(ptr) = (val))
#endif
#ifdef KHIP_TILE_WPE                 // experiment build: ask for KHIP_TILE_WPE waves per SIMD (a register budget of 512 / WPE)
#define KHIP_TILE_WPE_ATTR __attribute__((amdgpu_waves_per_eu(KHIP_TILE_WPE, KHIP_TILE_WPE)))
#else
#define KHIP_TILE_WPE_ATTR
#endif
template <int L, bool DIST, int NL, bool AHEAD>
__global__ __launch_bounds__(128) KHIP_TILE_WPE_ATTR void spmm_tile2_kernel(SpmvArgs a, TileArgs w) {
  using S = TileShape<L>;
  static_assert(S::NPASS >= 2, "two waves per window need two row passes per group");
  constexpr int NP = S::NPASS / 2;                   // row passes per wave
  extern __shared__ dbl2 tile_win[];
  typedef __attribute__((address_space(3))) char lds_char;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, sub = lane / L, c = lane % L;
  const int64_t last = w.groups - 1;
  // Which workgroup takes which groups (runs).  Workgroup b runs on XCD b % 8.  Three orders:
  //   eighths (default on grid tiles until round 6): XCD x walks the x-th eighth of the groups -- eight fronts, neighbours share an L2;
  //   exp & 8   : plain round-robin, workgroup b takes b, b + G, ... -- ONE front, neighbours on different XCDs;
  //   exp & 128 : one front in chunks -- per sweep of G groups XCD x takes the x-th contiguous chunk of G / 8, so that neighbours
  //               share an L2 AND all XCDs stay within G consecutive groups (their common panel rows meet in the Infinity Cache).
  const bool eighths = !(w.exp & (8 | 128)) && (gridDim.x & 7) == 0;
  const int64_t lin = ((w.exp & 128) && (gridDim.x & 7) == 0) ? (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : (int64_t)blockIdx.x;
  int64_t G = gridDim.x, g = lin, gend = w.groups;
  if (eighths) {
    const int64_t x = blockIdx.x & 7;
    G = gridDim.x >> 3;
    g = x * w.per_xcd + (blockIdx.x >> 3);
    gend = (x + 1) * w.per_xcd < w.groups ? (x + 1) * w.per_xcd : w.groups;
  }
  // sliding windows (w.run_len > 0): as in spmm_tile_kernel -- the workgroup walks whole runs of groups and copies only the
  // octets of panel rows its window does not hold yet (mask = aux word of row slot 2)
  const bool slide = w.run_len > 0;
  int64_t run = 0;
  int tpos = 0;
  if (slide) {
    const int64_t x = blockIdx.x & 7;
    G = eighths ? (gridDim.x >> 3) : gridDim.x;
    run = eighths ? x * w.runs_per_xcd + (blockIdx.x >> 3) : lin;
    const int64_t run_end = eighths ? ((x + 1) * w.runs_per_xcd < w.runs ? (x + 1) * w.runs_per_xcd : w.runs) : w.runs;
    g = run * w.run_len;
    gend = run_end * w.run_len;
  }
  auto next_of = [&](int64_t &rr, int &tt) -> int64_t {
    if (++tt == w.run_len) { tt = 0; rr += G; }
    return rr * w.run_len + tt;
  };
  if (g >= gend) return;
  const int hq = (sub & 2) ? 16 * L : 0;
  const char *xa = reinterpret_cast<const char *>(tile_win) + 16 * c + hq;
  const char *xb_ = reinterpret_cast<const char *>(tile_win) + 16 * c + (16 * L - hq);
  const int64_t vlast = a.nnz_bound - 2;
  const int slot_off = kTileDescBytes + 4 * w.cap;
  struct Rec { int4v d[NP]; int lw[NL]; int flag; int mask; int look; };
  struct Ent { dbl2 v[NP][S::F]; int sw[NP][S::SW]; };

  auto load_rec = [&](int64_t gg, Rec &r) {
    const char *rec = w.meta + (gg < last ? gg : last) * (int64_t)w.stride;
    const int4v *desc = reinterpret_cast<const int4v *>(rec);
    const int32_t *lst = reinterpret_cast<const int32_t *>(rec + kTileDescBytes);
#pragma unroll
    for (int q = 0; q < NP; ++q) r.d[q] = KHIP_T2_LD(desc + (wv * NP + q) * S::RP + sub);
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int q = lane + 64 * j;
      r.lw[j] = KHIP_T2_LD(lst + (q < w.cap ? q : w.cap - 1));
    }
    r.flag = reinterpret_cast<const int *>(rec)[7];                      // aux word of row slot 1: the group's direct-path flag
    r.mask = reinterpret_cast<const int *>(rec)[11];                     // ... of row slot 2: the octets of slots to copy (sliding windows)
    r.look = reinterpret_cast<const int *>(rec)[15];                     // ... of row slot 3: 1 = the new rows avoid the predecessor's slots (look-ahead)
  };
  auto load_ent = [&](int64_t gg, const Rec &r, Ent &e) {
    const char *slots = w.meta + (gg < last ? gg : last) * (int64_t)w.stride + slot_off;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
#pragma unroll
      for (int f = 0; f < S::F; ++f) {
        int64_t i0 = (int64_t)r.d[q].y + 2 * c + 2 * L * f;
        i0 = i0 < vlast ? i0 : vlast;
        if (w.exp & 4) i0 = 2 * c + 2 * L * f;                             // phase experiment: no entry stream
        e.v[q][f] = KHIP_T2_LD(reinterpret_cast<const dbl2u *>(a.val + i0));
      }
      const char *sp = slots + ((wv * NP + q) * S::RP + sub) * kTileLen + 4 * S::SW * c;
      if (S::SW == 1) {
        e.sw[q][0] = KHIP_T2_LD(reinterpret_cast<const int *>(sp));
      } else {
        const int2v t = KHIP_T2_LD(reinterpret_cast<const int2v *>(sp));
        e.sw[q][0] = t.x; e.sw[q][S::SW - 1] = t.y;
      }
    }
  };
  const unsigned win_lds = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(lds_char *)tile_win);
  auto issue_dma = [&](const Rec &r, unsigned mask) {                     // this wave's half: the instructions wq with wq % 2 == wv
    constexpr int PER = 64 / S::EPI;
    int col[NL * PER];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
#pragma unroll
      for (int u = 0; u < PER; ++u) col[j * PER + u] = __builtin_amdgcn_ds_bpermute(4 * (u * S::EPI + lane / (2 * L)), r.lw[j]);
    }
#pragma unroll
    for (int wq = 0; wq < NL * PER; ++wq) {
      constexpr unsigned OCT = S::EPI >= 8 ? (1u << (S::EPI / 8)) - 1u : 1u;
      if ((wq & 1) == wv && S::EPI * wq < w.cap && !(w.exp & 1) && ((mask >> ((S::EPI * wq) >> 3)) & OCT) != 0) {
        const char *src = reinterpret_cast<const char *>(a.x);
        uint64_t rr = (unsigned)col[wq];
        if (DIST) {
          const bool own = (int64_t)col[wq] < a.n_owned;
          src = own ? src : reinterpret_cast<const char *>(a.ghost);
          rr = own ? rr : rr - (uint64_t)a.n_owned;
        }
        const char *gsrc = src + (rr << w.gshift) + w.coff + 16 * (lane % (2 * L));
        const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(win_lds + 1024u * (unsigned)wq));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
      }
    }
  };
  auto products = [&](const Rec &r, const Ent &e) {
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const int4v &d = r.d[q];
      const int len = d.z;
      const int len0 = __builtin_amdgcn_readfirstlane(len);
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      if (w.exp & 2) {                                                      // phase experiment: no products
        acc[0] = e.v[q][0].x + e.v[q][S::F - 1].y + (double)e.sw[q][0];
      } else if (__ballot(len != len0) == 0) {
        tile_rows<L, false, true>(e.v[q], e.sw[q], xa, xb_, len, len0, acc);
      } else {
        int nmax = len;
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) { const int o = __shfl_xor(nmax, sft); nmax = o > nmax ? o : nmax; }
        nmax = __builtin_amdgcn_readfirstlane(nmax);
        tile_rows<L, true>(e.v[q], e.sw[q], xa, xb_, len, nmax, acc);
      }
      if (d.x >= 0) {
        double *yr = a.y + ((int64_t)d.x << (w.gshift - 3)) + (w.coff >> 3) + 2 * c;
        KHIP_T2_ST(reinterpret_cast<dbl2 *>(yr + (hq >> 3)), (dbl2{acc[0], acc[1]}));
        KHIP_T2_ST(reinterpret_cast<dbl2 *>(yr + 2 * L - (hq >> 3)), (dbl2{acc[2], acc[3]}));
      }
    }
  };

  Rec r0, r1, r2;
  Ent e0, e1;
  int64_t run1 = run, run2;
  int t1 = tpos, t2;
  int64_t g1 = slide ? next_of(run1, t1) : g + G;
  run2 = run1; t2 = t1;
  int64_t g2 = slide ? next_of(run2, t2) : g + 2 * G;
  load_rec(g, r0);
  load_rec(g1, r1);
  load_ent(g, r0, e0);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  if constexpr (AHEAD) {
    // LOOK-AHEAD ON SLIDING WINDOWS (round 6).  Per group: everything this wave has in flight has landed (vmcnt(0): the copies of
    // g, its entries, the record of g + 1) -> ONE barrier (the partner's half of the copies has landed too, and both waves are
    // past the products of g - 1) -> issue the copies of g + 1 (its new rows avoid every slot g reads: record flag `look`), its
    // entries and the record of g + 2 -> products of g while those fly.  One memory round trip per group is overlapped with the
    // products instead of following them, one barrier per group instead of two, and the copies are half a window (the rows the
    // window does not hold yet).  A group whose record does not allow it (first of a run, or its new rows did not fit) is
    // copied after a second barrier behind the products, as in the loop below.  vmcnt(0) keeps hipcc's own wait bookkeeping
    // exact (the asm copies are outside it): nothing is in flight across the top of the loop.
    if (__builtin_amdgcn_readfirstlane(r0.flag) == 0) issue_dma(r0, (unsigned)__builtin_amdgcn_readfirstlane(r0.mask));
    for (;;) {
      __builtin_amdgcn_s_waitcnt(0x0F70);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      const bool direct = __builtin_amdgcn_readfirstlane(r0.flag) != 0;
      const bool more = g1 < gend;
      bool direct1 = true, early = false;
      if (more) {
        direct1 = __builtin_amdgcn_readfirstlane(r1.flag) != 0;
        early = !direct1 && __builtin_amdgcn_readfirstlane(r1.look) != 0;
        if (early) issue_dma(r1, (unsigned)__builtin_amdgcn_readfirstlane(r1.mask));
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        load_ent(g1, r1, e1);
        load_rec(g2, r2);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!direct) products(r0, e0);
      if (!more) break;
      if (!early && !direct1) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // both waves are done with the window: it may be filled anew
        issue_dma(r1, (unsigned)__builtin_amdgcn_readfirstlane(r1.mask));
      }
      g = g1; g1 = g2; g2 = next_of(run2, t2);
      r0 = r1; e0 = e1; r1 = r2;
    }
    return;
  } else {
  for (;;) {
    const bool direct = __builtin_amdgcn_readfirstlane(r0.flag) != 0;      // the same for both waves: barriers stay matched
    const unsigned mask = slide ? (unsigned)__builtin_amdgcn_readfirstlane(r0.mask) : 0xffffffffu;
    if (!direct) issue_dma(r0, mask);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    load_ent(g1, r1, e1);
    load_rec(g2, r2);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (!direct) {
      constexpr int N = NP * (S::F + 1) + NP + NL + 2;                      // loads issued after this wave's copies
      __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
      asm volatile("s_barrier" ::: "memory");                               // both halves of the window have landed
      __builtin_amdgcn_sched_barrier(0);
      products(r0, e0);
    }
    if (slide) { g = g1; g1 = g2; g2 = next_of(run2, t2); } else { g += G; g1 = g + G; g2 = g + 2 * G; }
    if (g >= gend) break;
    if (!direct) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // both waves are done with the window before the next copies
    r0 = r1; e0 = e1; r1 = r2;
  }
  }
}

// The flagged groups (a row longer than 32 entries, or more distinct columns than the window holds): direct gathers, same
// order of operations.  One workgroup per flagged group, launched after the main kernel over the handle's list of such groups
// (kept out of the main kernel: a second arm with loads of its own made hipcc's wait-count merging drain the prefetches there).
// A row of thousands of entries must not be one serial chain of dependent gathers (four such rows cost 1.5 ms at 10 M rows,
// profiles/r03_bench_irregular.jsonl): the 256 threads form the products of 256 / L entries at a time in parallel (thread t:
// entry t / L, columns 4 (t % L) .. + 3), park them in LDS, and L lanes add them up in stored order -- the same rounded
// multiply and rounded add per entry and column as everywhere else.
template <int L, bool DIST>
__global__ __launch_bounds__(kBlock) void spmm_tile_direct_kernel(SpmvArgs a, TileArgs w, const int32_t *glist, int64_t count) {
  constexpr int CH = kBlock / L, P = 4 * L;
  __shared__ double prod[CH][P];
  const int tid = threadIdx.x, e = tid / L, c = tid % L;
  if ((int64_t)blockIdx.x >= count) return;
  const int64_t g = glist[blockIdx.x];
  const int4v *desc = reinterpret_cast<const int4v *>(w.meta + g * (int64_t)w.stride);
  for (int t = 0; t < kTileR; ++t) {
    const int4v d = desc[t];
    if (d.x < 0) continue;                                  // (uniform)
    const int64_t s = d.y, len = d.z;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t base = 0; base < len; base += CH) {
      const int cnt = (int)((len - base) < CH ? (len - base) : CH);
      if (e < cnt) {
        const double vv = a.val[s + base + e];
        const int32_t cc = a.col[s + base + e];
        const bool own = !DIST || cc < a.n_owned;
        const double *src = own ? a.x : a.ghost;
        const int64_t rr = own ? (int64_t)cc : (int64_t)cc - a.n_owned;
        const double *xr = src + (rr << (w.gshift - 3)) + (w.coff >> 3) + 4 * c;
        const dbl2 x0 = *reinterpret_cast<const dbl2 *>(xr);
        const dbl2 x1 = *reinterpret_cast<const dbl2 *>(xr + 2);
        prod[e][4 * c + 0] = vv * x0.x;
        prod[e][4 * c + 1] = vv * x0.y;
        prod[e][4 * c + 2] = vv * x1.x;
        prod[e][4 * c + 3] = vv * x1.y;
      }
      __syncthreads();
      if (tid < L) {
        for (int k = 0; k < cnt; ++k) {
          acc[0] = acc[0] + prod[k][4 * c + 0];
          acc[1] = acc[1] + prod[k][4 * c + 1];
          acc[2] = acc[2] + prod[k][4 * c + 2];
          acc[3] = acc[3] + prod[k][4 * c + 3];
        }
      }
      __syncthreads();
    }
    if (tid < L) {
      double *yr = a.y + ((int64_t)d.x << (w.gshift - 3)) + (w.coff >> 3) + 4 * c;
      *reinterpret_cast<dbl2 *>(yr) = dbl2{acc[0], acc[1]};
      *reinterpret_cast<dbl2 *>(yr + 2) = dbl2{acc[2], acc[3]};
    }
  }
}

// list of the flagged groups (order irrelevant)
__global__ __launch_bounds__(kBlock) void spmm_tile_flagged_kernel(const char *meta, int stride, int64_t groups, int32_t *glist,
                                                                    unsigned long long *count) {
  const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (g >= groups) return;
  const int4v d1 = reinterpret_cast<const int4v *>(meta + g * (int64_t)stride)[1];
  if (d1.w != 0) glist[atomicAdd(count, 1ull)] = (int32_t)g;
}

// ---------------------------------------------------------------- metadata ----------
// Row of slot t (0..31) of group g.  Tile order (s1 > 0): the group is the 4 x 4 x 2 (or, for one plane, 8 x 4 x 1) tile
// (ti, tj, tk) of the n1 x n2 x n3 grid whose row index is i + s1 j + s2 k; identity order: rows 32 g .. 32 g + 31.
struct TileOrder {
  int64_t m;
  int64_t s1, s2;       // s1 == 0: identity order
  int n1, n2, n3;
  int bi, bj, bk;       // tile extents (bi * bj * bk == 32)
  int gi, gj, gk;       // tiles along i, j, k
  int pj;               // pencil width in tiles along j (group order: i fastest, then j inside a pencil, then k, then the pencils)
  int run_pieces;       // runs per line of tiles along the sliding direction (1: whole lines)
  int run_len;          // > 0: sliding windows -- the groups are numbered in runs of run_len along the slowest tile direction
                        // (k; j on a single plane), run r = (ti, tj) with ti fastest; identity order: runs of consecutive groups
};
__device__ __forceinline__ int64_t tile_row_of(const TileOrder &o, int64_t g, int t) {
  if (o.s1 == 0) { const int64_t r = g * kTileR + t; return r < o.m ? r : -1; }
  if (o.run_len > 0) {
    const int64_t rid0 = g / o.run_len, nlines = o.gk > 1 ? (int64_t)o.gi * o.gj : (int64_t)o.gi;
    const int64_t rid = rid0 % nlines, pos = (rid0 / nlines) * o.run_len + g % o.run_len;      // pieces of a line: slowest, so that consecutive runs are neighbouring lines
    int64_t ti, tj, tk;
    if (o.gk > 1) { ti = rid % o.gi; tj = rid / o.gi; tk = pos; }
    else { ti = rid; tj = pos; tk = 0; }
    if (ti >= o.gi || tj >= o.gj || tk >= o.gk) return -1;
    const int di = t % o.bi, dj = (t / o.bi) % o.bj, dk = t / (o.bi * o.bj);
    const int64_t i = ti * o.bi + di, j = tj * o.bj + dj, k = tk * o.bk + dk;
    if (i >= o.n1 || j >= o.n2 || k >= o.n3) return -1;
    const int64_t r = i + o.s1 * j + o.s2 * k;
    return r < o.m ? r : -1;
  }
  // pencil order: the groups of pj tile rows are walked through ALL planes before the next pj tile rows, so that the
  // k-neighbours of a tile are pj * gi groups away (inside the set of groups an XCD has in flight) instead of gi * gj
  const int64_t per_pencil = (int64_t)o.gi * o.pj * o.gk;          // groups of a full-width pencil
  const int64_t pen = g / per_pencil;
  const int wj = (pen + 1) * o.pj <= o.gj ? o.pj : o.gj - (int)(pen * o.pj);      // the last pencil may be narrower
  const int64_t rem = g - pen * per_pencil;
  const int64_t ti = rem % o.gi, tjj = (rem / o.gi) % wj, tk = rem / ((int64_t)o.gi * wj);
  const int64_t tj = pen * o.pj + tjj;
  if (tk >= o.gk) return -1;
  const int di = t % o.bi, dj = (t / o.bi) % o.bj, dk = t / (o.bi * o.bj);
  const int64_t i = ti * o.bi + di, j = tj * o.bj + dj, k = tk * o.bk + dk;
  if (i >= o.n1 || j >= o.n2 || k >= o.n3) return -1;
  const int64_t r = i + o.s1 * j + o.s2 * k;
  return r < o.m ? r : -1;
}

constexpr int kTileKeys = kTileR * kTileLen;   // 1024 column indices of a group at most
constexpr int kTileEmpty = 0x7fffffff;

struct TileBuildShared {
  int keys[kTileKeys];
  int uniq[kTileKeys];
  int scan[kBlock];
  int rrow[kTileR], rstart[kTileR], rlen[kTileR];
  int too_long;
};

// The distinct columns of group g, ascending, in S.uniq[0 .. n); returns n, or -1 when a row has more than 32 entries.  Fills
// S.rrow / rstart / rlen.  Called by all threads of the workgroup; ends with a barrier.
__device__ int tile_group_unique(TileBuildShared &S, const int32_t *rowptr, const int32_t *col, const TileOrder &o, int64_t g) {
  const int tid = threadIdx.x;
  __syncthreads();                     // the previous group's arrays are no longer read
  if (tid == 0) S.too_long = 0;
  __syncthreads();
  if (tid < kTileR) {
    const int64_t r = tile_row_of(o, g, tid);
    int s = 0, len = 0;
    if (r >= 0) { s = rowptr[r]; len = rowptr[r + 1] - s; }
    S.rrow[tid] = (int)r; S.rstart[tid] = s; S.rlen[tid] = len;
    if (len > kTileLen) S.too_long = 1;
  }
  __syncthreads();
  const bool longrow = S.too_long != 0;
  for (int q = tid; q < kTileKeys; q += kBlock) {
    const int t = q / kTileLen, k = q % kTileLen;
    S.keys[q] = (!longrow && k < S.rlen[t]) ? col[(int64_t)S.rstart[t] + k] : kTileEmpty;
  }
  __syncthreads();
  if (longrow) return -1;
  for (int size = 2; size <= kTileKeys; size <<= 1) {
    for (int strd = size >> 1; strd > 0; strd >>= 1) {
      for (int q = tid; q < kTileKeys / 2; q += kBlock) {
        const int lo = 2 * q - (q & (strd - 1));      // index with bit `strd` clear
        const int hi = lo + strd;
        const bool up = (lo & size) == 0;
        const int x = S.keys[lo], y = S.keys[hi];
        if ((x > y) == up) { S.keys[lo] = y; S.keys[hi] = x; }
      }
      __syncthreads();
    }
  }
  // unique: thread t owns keys KPT t .. KPT t + KPT - 1
  constexpr int KPT = kTileKeys / kBlock;
  int mine = 0;
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const int q = KPT * tid + j;
    const int k = S.keys[q];
    mine += (k != kTileEmpty && (q == 0 || S.keys[q - 1] != k)) ? 1 : 0;
  }
  S.scan[tid] = mine;
  __syncthreads();
  for (int d = 1; d < kBlock; d <<= 1) {
    const int add = tid >= d ? S.scan[tid - d] : 0;
    __syncthreads();
    S.scan[tid] += add;
    __syncthreads();
  }
  const int n_uniq = S.scan[kBlock - 1];
  int pos = S.scan[tid] - mine;
#pragma unroll
  for (int j = 0; j < KPT; ++j) {
    const int q = KPT * tid + j;
    const int k = S.keys[q];
    if (k != kTileEmpty && (q == 0 || S.keys[q - 1] != k)) S.uniq[pos++] = k;
  }
  __syncthreads();
  return n_uniq;
}

// One workgroup per group: the group's column indices sorted (bitonic, LDS), made unique, counted.  FILL = false:
// cnt[g] = number of distinct columns, or -1 when a row has more than 32 entries; statistics for the choice of the order
// and of the window size.  FILL = true: the group's record (slots = ranks of the columns: every group fills its window anew).
template <bool FILL>
__global__ __launch_bounds__(kBlock) void spmm_tile_build_kernel(const int32_t *rowptr, const int32_t *col, TileOrder o, int cap,
                                                                  int stride, char *meta, int32_t *cnt,
                                                                  unsigned long long *stat /* [0] long-row groups, [1] sum cnt, [2..34] histogram of ceil(cnt / 8) */) {
  __shared__ TileBuildShared S;
  const int tid = threadIdx.x;
  const int64_t g = blockIdx.x;
  const int n_uniq0 = tile_group_unique(S, rowptr, col, o, g);
  const bool longrow = n_uniq0 < 0;
  const int n_uniq = longrow ? 0 : n_uniq0;
  if (!FILL) {
    if (tid == 0) {
      cnt[g] = longrow ? -1 : n_uniq;
      if (longrow) atomicAdd(&stat[0], 1ull);
      else {
        atomicAdd(&stat[1], (unsigned long long)n_uniq);
        atomicAdd(&stat[2 + (n_uniq + 7) / 8], 1ull);
      }
    }
    return;
  }
  char *rec = meta + g * (int64_t)stride;
  const bool direct = longrow || n_uniq > cap;
  if (tid < kTileR) {
    int4v d;
    d.x = S.rrow[tid]; d.y = S.rstart[tid]; d.z = S.rlen[tid];
    d.w = tid == 0 ? n_uniq : (tid == 1 ? (direct ? 1 : 0) : (tid == 2 ? -1 : 0));
    reinterpret_cast<int4v *>(rec)[tid] = d;
  }
  int32_t *lst = reinterpret_cast<int32_t *>(rec + kTileDescBytes);
  uint8_t *slots = reinterpret_cast<uint8_t *>(rec + kTileDescBytes + 4 * cap);
  if (direct) {
    for (int q = tid; q < cap; q += kBlock) lst[q] = 0;
    for (int q = tid; q < kTileSlotBytes; q += kBlock) slots[q] = 0;
    return;
  }
  for (int q = tid; q < cap; q += kBlock) lst[q] = n_uniq > 0 ? S.uniq[q < n_uniq ? q : n_uniq - 1] : 0;   // padding: the last column again
  for (int q = tid; q < kTileKeys; q += kBlock) {
    const int t = q / kTileLen, k = q % kTileLen;
    int sl = 0;
    if (k < S.rlen[t]) {
      const int key = col[(int64_t)S.rstart[t] + k];
      int lo = 0, hi = n_uniq - 1;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (S.uniq[mid] < key) lo = mid + 1; else hi = mid;
      }
      sl = lo;
    }
    slots[q] = (uint8_t)sl;
  }
}

// Sliding windows: one workgroup per RUN, its groups one after the other.  slot_col[s] = the column whose panel row sits in
// slot s of the wave's window once the group's copies have landed.  A column the group shares with what the window holds
// keeps its slot; the others take, in ascending order, the slots of columns the group does not need (ascending too), and the
// octets of slots that received one are the group's copy mask (aux word of row slot 2).  The list keeps naming, for EVERY
// slot, the column that sits there, so an octet that is copied again for one new row rewrites its other seven rows with
// themselves.  A group on the direct-gather path (or an empty one) ends the chain: the next group fills the window anew.
//
// LOOK-AHEAD (ahead != 0, round 6).  The kernel wants to issue the copies of group t BEFORE the products of group t - 1 have
// read the window: the new columns of t then must not land in a slot that t - 1 still reads.  So they take only slots that
// neither t (kept columns) nor t - 1 (used_prev) uses -- the window needs |columns(t - 1)| + |new columns(t)| slots -- and the
// record says so (aux word of row slot 3 = 1).  Where the free slots do not suffice, or at the start of a chain, the old rule
// applies and the flag is 0: the kernel copies that group after the products of its predecessor, as before.  (An octet that is
// copied for one new row rewrites its other rows with THEMSELVES, also those t - 1 is reading: same bytes, no hazard.)
__global__ __launch_bounds__(kBlock) void spmm_tile_build_run_kernel(const int32_t *rowptr, const int32_t *col, TileOrder o, int cap,
                                                                      int stride, char *meta, int run_len, int ahead) {
  __shared__ TileBuildShared S;
  __shared__ int slot_col[kTileCapMax], uslot[kTileCapMax], freelist[kTileCapMax], flag[kBlock], used_prev[kTileCapMax], used_now[kTileCapMax];
  __shared__ unsigned mask_sh;
  const int tid = threadIdx.x;
  for (int q = tid; q < kTileCapMax; q += kBlock) { slot_col[q] = kTileEmpty; used_prev[q] = 0; }
  bool chain = false;                  // the previous group of this run left a window behind (uniform)
  for (int t = 0; t < run_len; ++t) {
    const int64_t g = (int64_t)blockIdx.x * run_len + t;
    const int n_uniq0 = tile_group_unique(S, rowptr, col, o, g);
    const bool longrow = n_uniq0 < 0;
    const int n_uniq = longrow ? 0 : n_uniq0;
    char *rec = meta + g * (int64_t)stride;
    const bool direct = longrow || n_uniq > cap;
    int32_t *lst = reinterpret_cast<int32_t *>(rec + kTileDescBytes);
    uint8_t *slots = reinterpret_cast<uint8_t *>(rec + kTileDescBytes + 4 * cap);
    if (tid == 0) mask_sh = 0u;
    int look = 0;
    if (direct) {
      for (int q = tid; q < cap; q += kBlock) lst[q] = 0;
      for (int q = tid; q < kTileSlotBytes; q += kBlock) slots[q] = 0;
      for (int q = tid; q < kTileCapMax; q += kBlock) { slot_col[q] = kTileEmpty; used_prev[q] = 0; }       // the chain ends here
      chain = false;
      __syncthreads();
    } else {
      // (a) slots whose column the group needs again keep it
      for (int q = tid; q < kTileCapMax; q += kBlock) uslot[q] = -1;
      __syncthreads();
      int keep = 0;
      if (tid < cap) {
        const int c = slot_col[tid];
        if (c != kTileEmpty && n_uniq > 0) {
          int lo = 0, hi = n_uniq - 1;
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (S.uniq[mid] < c) lo = mid + 1; else hi = mid;
          }
          if (S.uniq[lo] == c) { uslot[lo] = tid; keep = 1; }
        }
      }
      __syncthreads();
      // (b) how many columns are new, and how many slots are free of this group AND of the previous one
      const int isnew = (tid < n_uniq && uslot[tid] < 0) ? 1 : 0;
      flag[tid] = isnew;
      __syncthreads();
      for (int d = 1; d < kBlock; d <<= 1) {
        const int add = tid >= d ? flag[tid - d] : 0;
        __syncthreads();
        flag[tid] += add;
        __syncthreads();
      }
      const int new_rank = flag[tid], n_new = flag[kBlock - 1];
      __syncthreads();
      const int free2 = (tid < cap && !keep && !used_prev[tid]) ? 1 : 0;
      flag[tid] = free2;
      __syncthreads();
      for (int d = 1; d < kBlock; d <<= 1) {
        const int add = tid >= d ? flag[tid - d] : 0;
        __syncthreads();
        flag[tid] += add;
        __syncthreads();
      }
      const int n_free2 = flag[kBlock - 1];
      look = (ahead && chain && n_free2 >= n_new) ? 1 : 0;
      // (c) the free slots, ascending: free of this group (and, with look-ahead, of the previous one)
      int isfree = (tid < cap && !keep) ? 1 : 0;
      if (look) isfree = free2;
      __syncthreads();
      if (!look) {
        flag[tid] = isfree;
        __syncthreads();
        for (int d = 1; d < kBlock; d <<= 1) {
          const int add = tid >= d ? flag[tid - d] : 0;
          __syncthreads();
          flag[tid] += add;
          __syncthreads();
        }
      }
      if (isfree) freelist[flag[tid] - 1] = tid;
      used_now[tid] = keep;
      __syncthreads();
      // (d) the new columns, ascending, into them
      if (isnew) {
        const int sl = freelist[new_rank - 1];
        uslot[tid] = sl;
        slot_col[sl] = S.uniq[tid];
        used_now[sl] = 1;
        atomicOr(&mask_sh, 1u << (sl >> 3));
      }
      __syncthreads();
      used_prev[tid] = used_now[tid];
      chain = true;
      // (e) the record: list in slot order (an empty slot names the group's last column: any valid row will do), slot bytes
      for (int q = tid; q < cap; q += kBlock) lst[q] = slot_col[q] != kTileEmpty ? slot_col[q] : (n_uniq > 0 ? S.uniq[n_uniq - 1] : 0);
      for (int q = tid; q < kTileKeys; q += kBlock) {
        const int tt = q / kTileLen, k = q % kTileLen;
        int sl = 0;
        if (k < S.rlen[tt]) {
          const int key = col[(int64_t)S.rstart[tt] + k];
          int lo = 0, hi = n_uniq - 1;
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (S.uniq[mid] < key) lo = mid + 1; else hi = mid;
          }
          sl = uslot[lo];
        }
        slots[q] = (uint8_t)sl;
      }
      __syncthreads();
    }
    if (tid < kTileR) {
      int4v d;
      d.x = S.rrow[tid]; d.y = S.rstart[tid]; d.z = S.rlen[tid];
      d.w = tid == 0 ? n_uniq : (tid == 1 ? (direct ? 1 : 0) : (tid == 2 ? (int)mask_sh : (tid == 3 ? look : 0)));
      reinterpret_cast<int4v *>(rec)[tid] = d;
    }
  }
}

void csr_free_tiles(khip_csr *A) {
  (void)hipFree(A->tile_meta);
  (void)hipFree(A->tile_direct_list);
  A->tile_meta = nullptr;
  A->tile_direct_list = nullptr;
  A->tile_state = 0;
  A->tile_run_len = 0;
  A->tile_ahead = 0;
  A->tile_runs = 0;
}

static TileOrder tile_order_for(const khip_csr *A, bool tiles) {
  TileOrder o{};
  o.m = A->m;
  if (!tiles) return o;
  const int shape = A->ctx ? A->ctx->tune.spmm_tile_shape : 0;
  const int64_t s1 = A->line_rows, s2 = A->plane_rows > A->line_rows ? A->plane_rows : A->m;
  o.s1 = s1; o.s2 = s2;
  o.n1 = (int)s1; o.n2 = (int)((s2 + s1 - 1) / s1); o.n3 = (int)((A->m + s2 - 1) / s2);
  if (o.n3 > 1) {
    switch (shape) {                       // 4 x 4 x 2 (default): 6 x 6 x 4 = 144 panel rows of the 27-point box, runs of 4 consecutive rows
      case 1: o.bi = 8; o.bj = 2; o.bk = 2; break;
      case 2: o.bi = 2; o.bj = 4; o.bk = 4; break;
      case 3: o.bi = 4; o.bj = 2; o.bk = 4; break;
      case 4: o.bi = 8; o.bj = 4; o.bk = 1; break;
      case 5: o.bi = 32; o.bj = 1; o.bk = 1; break;
      default: o.bi = 4; o.bj = 4; o.bk = 2; break;
    }
    if (kTileR == 64) { if (shape == 6) o.bi *= 2; else if (shape == 7) o.bj *= 2; else o.bk *= 2; }      // 64-row groups: 4 x 4 x 4 by default (6: 8 x 4 x 2, 7: 4 x 8 x 2)
  } else { o.bi = 8; o.bj = kTileR / 8; o.bk = 1; }
  o.gi = (o.n1 + o.bi - 1) / o.bi;
  o.gj = (o.n2 + o.bj - 1) / o.bj;
  o.gk = (o.n3 + o.bk - 1) / o.bk;
  o.pj = A->ctx && A->ctx->tune.spmm_tile_pencil > 0 ? A->ctx->tune.spmm_tile_pencil : 4;      // 3..16 measure alike (1.20-1.31 ms at 216^3 x 16), 1 / 2 / whole planes 1.35 (profiles/r03g_spmm_tile_sweep.log)
  if (o.pj > o.gj) o.pj = o.gj;
  return o;
}
// sliding windows (ctx option spmm_tile_slide): runs along k (along j on a single plane); identity order: 64 consecutive groups
static void tile_order_set_runs(const khip_csr *A, TileOrder &o) {
  int slide = A->ctx ? A->ctx->tune.spmm_tile_slide : 0;
  o.run_len = 0;
  if (slide < 0) slide = o.s1 != 0 ? 27 : 0;      // default: runs of <= 27 groups on grids (with two waves per window: -3 ... -5 % at cfg 5, -16 % on 7-point grids), none without a grid (+3 % there); profiles/r04t_spmm_pair_slide.log, r04u_spmm_pair3.log
  if (!slide) return;
  if (o.s1 == 0) o.run_len = slide > 1 ? slide : 64;
  else {
    const int full = o.gk > 1 ? o.gk : o.gj;
    o.run_len = full;
    if (slide > 1 && slide < full) {           // shorter runs: pieces of (nearly) equal length
      const int pieces = (full + slide - 1) / slide;
      o.run_len = (full + pieces - 1) / pieces;
    }
    o.run_pieces = (full + o.run_len - 1) / o.run_len;
  }
  if (o.run_len < 2) o.run_len = 0;
}
static int64_t tile_runs_for(const TileOrder &o) {
  if (o.run_len <= 0) return 0;
  if (o.s1 == 0) return ((o.m + kTileR - 1) / kTileR + o.run_len - 1) / o.run_len;
  return (o.gk > 1 ? (int64_t)o.gi * o.gj : (int64_t)o.gi) * o.run_pieces;
}
static int64_t tile_groups_for(const TileOrder &o) {
  if (o.run_len > 0) return tile_runs_for(o) * o.run_len;        // padded to whole runs (rows past the end: -1)
  if (o.s1 == 0) return (o.m + kTileR - 1) / kTileR;
  return (int64_t)o.gi * o.gj * o.gk;
}

// Builds the group records of A for the p = 16 tile kernel.  On return A->tile_state is 1 (usable) or -1 (the operator
// lacks the locality, or memory is short: the caller keeps the other kernels and does not ask again).
int spmm_tile_build(khip_ctx *ctx, khip_csr *A) {
  csr_free_tiles(A);
  A->tile_state = -1;
  if (A->m <= 0 || A->nnz <= 0) return KHIP_OK;
  constexpr int NSTAT = 2 + kTileCapMax / 8 + 1 + 128;        // histogram bins for up to 1024 distinct columns
  int32_t *cnt = nullptr;
  unsigned long long *stat = nullptr;
  struct Scratch { int32_t *&c; unsigned long long *&s; ~Scratch() { (void)hipFree(c); (void)hipFree(s); } } scratch{cnt, stat};
  KHIP_CHECK_HIP(hipMalloc(&stat, NSTAT * sizeof(unsigned long long)));
  // candidate orders: identity, and the grid tiles when csr_finalize saw a line / plane structure
  const bool grid_ok = A->line_rows >= 4 && A->line_rows < A->m &&
                       (A->plane_rows <= A->line_rows || (A->plane_rows % A->line_rows == 0 && A->plane_rows / A->line_rows >= 4));
  TileOrder best{};
  std::vector<unsigned long long> best_st;
  double best_score = -1.0;
  for (int cand = 0; cand < (grid_ok ? 2 : 1); ++cand) {
    TileOrder o = tile_order_for(A, cand == 1);
    tile_order_set_runs(A, o);
    const int64_t groups = tile_groups_for(o);
    if (groups <= 0 || groups > ((int64_t)1 << 30)) continue;
    (void)hipFree(cnt); cnt = nullptr;
    KHIP_CHECK_HIP(hipMalloc(&cnt, sizeof(int32_t) * (size_t)groups));
    KHIP_CHECK_HIP(hipMemsetAsync(stat, 0, NSTAT * sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL((spmm_tile_build_kernel<false>), dim3((unsigned)groups), dim3(kBlock), 0, ctx->stream, A->rowptr, A->col, o, 0, 0,
                       (char *)nullptr, cnt, stat);
    KHIP_CHECK_HIP(hipGetLastError());
    std::vector<unsigned long long> st(NSTAT);
    KHIP_CHECK_HIP(hipMemcpyAsync(st.data(), stat, NSTAT * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    // score: references per distinct column (higher is better); long-row groups count as no reuse
    const double score = st[1] > 0 ? (double)A->nnz / ((double)st[1] + 32.0 * (double)st[0]) : 0.0;
    if (score > best_score) { best_score = score; best = o; best_st = st; }
  }
  if (best_score < 1.5) return KHIP_OK;                       // fewer than 1.5 references per distinct column: no reuse to stage
  const int64_t groups = tile_groups_for(best);
  // window size: the smallest multiple of 8 that takes 99 % of the groups (the others go down the direct path)
  unsigned long long fit = 0;
  int cap = -1;
  for (int bin = 0; bin <= kTileCapMax / 8; ++bin) {
    fit += best_st[2 + (size_t)bin];
    if (100 * fit >= 99 * (unsigned long long)groups) { cap = 8 * bin; break; }
  }
  if (cap < 0) {
    if (2 * fit < (unsigned long long)groups) return KHIP_OK;          // most groups exceed the largest window
    cap = kTileCapMax;
  }
  if (cap < 8) cap = 8;
  // look-ahead on sliding windows (tune.spmm_tile_ahead): the window also holds the NEW panel rows of the next group while the
  // current one is being read -- half a window more covers two shared planes out of four (4 x 4 x 2 tiles: 144 + 72 = 216
  // slots); a group whose new rows do not fit keeps the old rule (its record says so)
  const int ahead_opt = ctx->tune.spmm_tile_ahead;
  const bool ahead = best.run_len > 0 && ahead_opt != 0 && ctx->tune.spmm_tile_pair != 0 && cap + cap / 2 <= kTileCapMax;
  if (ahead) cap = ((cap + cap / 2) + 7) & ~7;
  const int stride = kTileDescBytes + 4 * cap + kTileSlotBytes;
  size_t free_b = 0, total_b = 0;
  KHIP_CHECK_HIP(hipMemGetInfo(&free_b, &total_b));
  if ((size_t)groups * (size_t)stride + ((size_t)1 << 30) > free_b) return KHIP_OK;
  KHIP_CHECK_HIP(hipMalloc(&A->tile_meta, (size_t)groups * (size_t)stride));
  if (best.run_len > 0)
    hipLaunchKernelGGL(spmm_tile_build_run_kernel, dim3((unsigned)tile_runs_for(best)), dim3(kBlock), 0, ctx->stream, A->rowptr, A->col, best, cap,
                       stride, A->tile_meta, best.run_len, ahead ? 1 : 0);
  else
    hipLaunchKernelGGL((spmm_tile_build_kernel<true>), dim3((unsigned)groups), dim3(kBlock), 0, ctx->stream, A->rowptr, A->col, best, cap, stride,
                       A->tile_meta, (int32_t *)nullptr, (unsigned long long *)nullptr);
  KHIP_CHECK_HIP(hipGetLastError());
  unsigned long long over = 0;
  for (size_t bin = (size_t)cap / 8 + 1; bin + 2 < best_st.size(); ++bin) over += best_st[2 + bin];
  const int64_t n_direct = (int64_t)(best_st[0] + over);
  if (n_direct > 0) {
    KHIP_CHECK_HIP(hipMalloc(&A->tile_direct_list, sizeof(int32_t) * (size_t)n_direct));
    KHIP_CHECK_HIP(hipMemsetAsync(stat, 0, sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(spmm_tile_flagged_kernel, dim3((unsigned)((groups + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream,
                       A->tile_meta, stride, groups, A->tile_direct_list, stat);
    KHIP_CHECK_HIP(hipGetLastError());
    unsigned long long got = 0;
    KHIP_CHECK_HIP(hipMemcpyAsync(&got, stat, sizeof(got), hipMemcpyDeviceToHost, ctx->stream));
    KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    if ((int64_t)got != n_direct) { set_error("spmm_tile_build: %lld flagged groups listed, %lld counted", (long long)got, (long long)n_direct); csr_free_tiles(A); A->tile_state = -1; return KHIP_ERR_HIP; }
  }
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  A->tile_state = 1;
  A->tile_cap = cap;
  A->tile_stride = stride;
  A->tile_groups = groups;
  A->tile_run_len = best.run_len;
  A->tile_ahead = ahead ? 1 : 0;
  A->tile_runs = tile_runs_for(best);
  A->tile_grid = best.s1 != 0 ? 1 : 0;
  A->tile_direct = n_direct;
  A->tile_reuse = best_score;
  return KHIP_OK;
}

template <int L>
static int launch_tile_L(khip_ctx *ctx, const khip_csr *A, const SpmvArgs &a, int gshift, int coff) {
  TileArgs w;
  w.run_len = A->tile_run_len; w.runs = A->tile_runs; w.runs_per_xcd = (A->tile_runs + 7) / 8;
  w.meta = A->tile_meta; w.groups = A->tile_groups; w.per_xcd = (A->tile_groups + 7) / 8; w.cap = A->tile_cap; w.stride = A->tile_stride; w.exp = ctx->tune.spmm_tile_exp; w.gshift = gshift; w.coff = coff;
  // two windows per wave where six waves per CU still fit with them (windows of <= 104 panel rows at p = 16: 7-point grids 1.12 ->
  // 0.99 ms); at cfg 5 (144 rows: four waves per CU, one per SIMD) the products' own latency shows: 1.84 against 1.36 ms
  // (profiles/r04r_spmm_dbuf.log).  spmm_tile_dbuf: -1 by that rule, 0 never, 1 always.  (Sliding windows carry ONE window.)
  const int dbuf_opt = ctx->tune.spmm_tile_dbuf;
  const bool use_pair = ctx->tune.spmm_tile_pair && L >= 4 && !ctx->tune.spmm_tile_nt;
  {
    // the order of the groups over the XCDs rides on bit 3 of w.exp (the kernels' round-robin switch)
    const int xo = ctx->tune.spmm_tile_xcd;
    if (xo == 2) w.exp |= 128;                       // one front in chunks of G / 8 per XCD
    else if (xo == 1 || (xo < 0 && A->tile_grid == 0 && A->tile_run_len == 0)) w.exp |= 8;
  }
  w.ahead = (use_pair && A->tile_ahead && w.run_len > 0 && ctx->tune.spmm_tile_ahead != 0) ? 1 : 0;
  w.dbuf = (!use_pair && (dbuf_opt > 0 || (dbuf_opt < 0 && (size_t)w.cap * 32 * L * 2 <= (size_t)26 * 1024))) ? 1 : 0;
  if (w.dbuf) w.run_len = 0;      // records built for sliding windows serve every other scheme too: their list names the column of EVERY slot, so a kernel that copies all octets of every group (in any order of the groups) fills a consistent window
  const size_t lds = (size_t)w.cap * 32 * L * (w.dbuf ? 2 : 1);
  int per_cu = (int)((size_t)(160 * 1024) / lds);                    // LDS-limited residency of the one-wave workgroups
  if (per_cu >= 5) --per_cu;                                         // one wave short of the LDS limit measures 3 % faster (7 instead of 8 at 144 panel rows)
  if (per_cu > 16) per_cu = 16;
  if (!use_pair) {
    // ... and the registers: the one-wave kernel holds 216-344 VGPRs (one or two waves per SIMD); a persistent grid beyond what
    // a CU takes leaves whole workgroups waiting for a slot (round 6; the two-wave kernel asks the same question below)
    const bool dist0 = a.ghost != a.x, nt0 = ctx->tune.spmm_tile_nt != 0;
    const int NL0 = (w.cap + 63) / 64;
    const void *fn = nullptr;
#define KHIP_TILE1_FN(D, N) fn = nt0 ? (const void *)spmm_tile_kernel<L, D, N, true> : (const void *)spmm_tile_kernel<L, D, N, false>
    if (dist0) { switch (NL0) { case 1: KHIP_TILE1_FN(true, 1); break; case 2: KHIP_TILE1_FN(true, 2); break; case 3: KHIP_TILE1_FN(true, 3); break; default: KHIP_TILE1_FN(true, 4); break; } }
    else       { switch (NL0) { case 1: KHIP_TILE1_FN(false, 1); break; case 2: KHIP_TILE1_FN(false, 2); break; case 3: KHIP_TILE1_FN(false, 3); break; default: KHIP_TILE1_FN(false, 4); break; } }
#undef KHIP_TILE1_FN
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 64, lds) == hipSuccess && nb > 0 && nb < per_cu) per_cu = nb;
  }
  if (per_cu < 1) per_cu = 1;
  int64_t grid = (int64_t)ctx->num_cu * per_cu * (ctx->tune.spmm_tile_waves > 0 ? ctx->tune.spmm_tile_waves : 1);
  if (ctx->tune.spmm_tile_grid > 0) grid = ctx->tune.spmm_tile_grid;
  if (grid > w.groups) grid = w.groups;
  if (w.run_len > 0 && ctx->tune.spmm_tile_grid <= 0) {
    // sliding windows: a wave takes whole runs, so the waves must get equally many -- the largest grid below the residency
    // limit that gives every wave of an XCD ceil(runs per XCD / waves per XCD) runs with no wave left half empty
    const int64_t per_x = w.runs_per_xcd, max_wx = grid / 8 > 0 ? grid / 8 : 1;
    const int64_t k = (per_x + max_wx - 1) / max_wx;                 // runs per wave
    if (k < 4) grid = 8 * ((per_x + k - 1) / k);                     // from four runs per wave on: the full grid (as for the two-wave kernel below)
  }
  if (w.run_len > 0 && grid > w.runs) grid = w.runs;
  if (grid >= 64) grid &= ~(int64_t)7;                               // whole waves per XCD
  const bool dist = a.ghost != a.x;
  const int NL = (w.cap + 63) / 64;
  if (use_pair) {
    // two waves per window (spmm_tile2_kernel): the residency is counted in windows as before, every window now carries two waves
    const size_t lds1 = (size_t)w.cap * 32 * L;
    int wg_per_cu = (int)((size_t)(160 * 1024) / lds1);
    {
      // how many of the kernel's workgroups a CU takes (LDS AND registers: the chunked product loop of round 6 holds 141 VGPRs at
      // NL = 3, three waves per SIMD) -- a persistent grid larger than that leaves whole workgroups waiting for a slot.  The
      // look-ahead kernel keeps its waves a multiple of the four SIMDs (5 workgroups = 10 waves ran 45 % slower than 4,
      // profiles/r06b_spmm_ahead_ab.jsonl); the round-4 loop stays one workgroup short of an LDS limit of 5 and more (3 % faster).
      int nb = 0;
      const void *fn = nullptr;
#define KHIP_TILE2_FN(D, N) fn = w.ahead ? (const void *)spmm_tile2_kernel<(L >= 4 ? L : 4), D, N, true> : (const void *)spmm_tile2_kernel<(L >= 4 ? L : 4), D, N, false>
      if (dist) { switch (NL) { case 1: KHIP_TILE2_FN(true, 1); break; case 2: KHIP_TILE2_FN(true, 2); break; case 3: KHIP_TILE2_FN(true, 3); break; default: KHIP_TILE2_FN(true, 4); break; } }
      else      { switch (NL) { case 1: KHIP_TILE2_FN(false, 1); break; case 2: KHIP_TILE2_FN(false, 2); break; case 3: KHIP_TILE2_FN(false, 3); break; default: KHIP_TILE2_FN(false, 4); break; } }
#undef KHIP_TILE2_FN
      if (!w.ahead && wg_per_cu >= 5) --wg_per_cu;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 128, lds1) == hipSuccess && nb > 0 && nb < wg_per_cu) wg_per_cu = nb;
      if (w.ahead && wg_per_cu > 2) wg_per_cu &= ~1;
    }
    if (wg_per_cu > 8) wg_per_cu = 8;
    if (wg_per_cu < 1) wg_per_cu = 1;
    int64_t grid2 = ctx->tune.spmm_tile_grid > 0 ? ctx->tune.spmm_tile_grid : (int64_t)ctx->num_cu * wg_per_cu;
    if (w.run_len > 0 && ctx->tune.spmm_tile_grid <= 0) {
      // sliding windows: a workgroup takes whole runs.  With few runs per workgroup they must get equally many (the grid shrinks to
      // the largest one that does it); from four runs per workgroup on, every slot of every CU filled is worth more than an even
      // last round (cfg 5, 6 workgroups per CU: 1536 workgroups of 7 or 8 runs 1.140 ms, 1464 of 8 runs each 1.190 ms,
      // profiles/r06e_spmm_chunk_ab.log)
      const int64_t per_x = w.runs_per_xcd, max_wx = grid2 / 8 > 0 ? grid2 / 8 : 1;
      const int64_t kk = (per_x + max_wx - 1) / max_wx;
      if (kk < 4) grid2 = 8 * ((per_x + kk - 1) / kk);
    }
    if (w.run_len > 0 && grid2 > w.runs) grid2 = w.runs;
    if (grid2 > w.groups) grid2 = w.groups;
    if (grid2 >= 64) grid2 &= ~(int64_t)7;
    const dim3 gd2((unsigned)grid2), bd2(128);
#define KHIP_TILE2(D, N)                                                                                                     \
  do {                                                                                                                       \
    if (w.ahead) hipLaunchKernelGGL((spmm_tile2_kernel<(L >= 4 ? L : 4), D, N, true>), gd2, bd2, lds1, ctx->stream, a, w);   \
    else hipLaunchKernelGGL((spmm_tile2_kernel<(L >= 4 ? L : 4), D, N, false>), gd2, bd2, lds1, ctx->stream, a, w);          \
  } while (0)
    if (dist) { switch (NL) { case 1: KHIP_TILE2(true, 1); break; case 2: KHIP_TILE2(true, 2); break; case 3: KHIP_TILE2(true, 3); break; default: KHIP_TILE2(true, 4); break; } }
    else      { switch (NL) { case 1: KHIP_TILE2(false, 1); break; case 2: KHIP_TILE2(false, 2); break; case 3: KHIP_TILE2(false, 3); break; default: KHIP_TILE2(false, 4); break; } }
#undef KHIP_TILE2
    if (A->tile_direct > 0) {
      const dim3 gdd((unsigned)A->tile_direct), bdd(kBlock);
      if (dist) hipLaunchKernelGGL((spmm_tile_direct_kernel<L, true>), gdd, bdd, 0, ctx->stream, a, w, A->tile_direct_list, A->tile_direct);
      else hipLaunchKernelGGL((spmm_tile_direct_kernel<L, false>), gdd, bdd, 0, ctx->stream, a, w, A->tile_direct_list, A->tile_direct);
    }
    KHIP_CHECK_HIP(hipGetLastError());
    return KHIP_OK;
  }
  const dim3 gd((unsigned)grid), bd(64);
#define KHIP_TILE_LAUNCH1(D, N, T)                                                                                      \
  do {                                                                                                                  \
    if (lds > 64 * 1024)                                                                                                \
      (void)hipFuncSetAttribute((const void *)spmm_tile_kernel<L, D, N, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((spmm_tile_kernel<L, D, N, T>), gd, bd, lds, ctx->stream, a, w);                                 \
  } while (0)
#define KHIP_TILE_LAUNCH(D, N)                                                                                          \
  do {                                                                                                                  \
    if (nt) KHIP_TILE_LAUNCH1(D, N, true); else KHIP_TILE_LAUNCH1(D, N, false);                                         \
  } while (0)
  const bool nt = ctx->tune.spmm_tile_nt != 0;
  if (dist) {
    switch (NL) { case 1: KHIP_TILE_LAUNCH(true, 1); break; case 2: KHIP_TILE_LAUNCH(true, 2); break; case 3: KHIP_TILE_LAUNCH(true, 3); break; default: KHIP_TILE_LAUNCH(true, 4); break; }
  } else {
    switch (NL) { case 1: KHIP_TILE_LAUNCH(false, 1); break; case 2: KHIP_TILE_LAUNCH(false, 2); break; case 3: KHIP_TILE_LAUNCH(false, 3); break; default: KHIP_TILE_LAUNCH(false, 4); break; }
  }
#undef KHIP_TILE_LAUNCH
#undef KHIP_TILE_LAUNCH1
  if (A->tile_direct > 0) {
    const dim3 gdd((unsigned)A->tile_direct), bdd(kBlock);
    if (dist) hipLaunchKernelGGL((spmm_tile_direct_kernel<L, true>), gdd, bdd, 0, ctx->stream, a, w, A->tile_direct_list, A->tile_direct);
    else hipLaunchKernelGGL((spmm_tile_direct_kernel<L, false>), gdd, bdd, 0, ctx->stream, a, w, A->tile_direct_list, A->tile_direct);
  }
  KHIP_CHECK_HIP(hipGetLastError());
  return KHIP_OK;
}

// p = 8, 16 or 32 right-hand sides in one launch (L = p / 4 lanes per row), or any multiple of 16 as p / 16 launches of the
// 16-column kernel over column slices of the panels (tune.spmm_tile_slices; the matrix stream is read once per slice, the
// window reads and the residency are those of p = 16).  The group records do not depend on p.
int launch_spmm_tile(khip_ctx *ctx, const khip_csr *A, const SpmvArgs &a, int p, bool slices) {
  if (slices && p % 16 == 0 && (p & (p - 1)) == 0 && p > 16) {
    int gshift = 3;
    while ((1 << gshift) < 8 * p) ++gshift;                    // 8 p bytes per panel row
    for (int c = 0; c < p / 16; ++c) KHIP_TRY(launch_tile_L<4>(ctx, A, a, gshift, 128 * c));
    return KHIP_OK;
  }
  if ((size_t)A->tile_cap * 8 * (size_t)p > (size_t)160 * 1024) { set_error("spmm_tile: window of %d panel rows x %d columns exceeds the LDS", A->tile_cap, p); return KHIP_ERR_UNSUPPORTED; }
  switch (p) {
    case 8: return launch_tile_L<2>(ctx, A, a, 6, 0);
    case 16: return launch_tile_L<4>(ctx, A, a, 7, 0);
    case 32: return launch_tile_L<8>(ctx, A, a, 8, 0);
    default: set_error("spmm_tile: p = 8, 16, 32 or a power of two above"); return KHIP_ERR_INVALID;
  }
}

}  // namespace khip
