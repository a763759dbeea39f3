// khip_internal.hpp -- shared host-side declarations of libkrylov_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <unistd.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/krylov_hip.h"

namespace khip {

void set_error(const char *fmt, ...);
// the hipError_t of the last failed HIP call on this thread, recorded AT the failing call by KHIP_CHECK_HIP (the runtime's own
// "last error" is overwritten by every later successful call on ROCm < 7: a builder's clean-up hipFree would hide an
// out-of-memory failure from optional_build -- ADVICE r05); take_hip_error() returns it and resets it to hipSuccess
void note_hip_error(int e);
int take_hip_error();

// options.verbose rows: stdout (log_fd = 0) or the caller's file descriptor (the reference's `iostream`, src/cg.jl:24)
inline void klogf(int fd, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  if (fd <= 0) vprintf(fmt, ap); else vdprintf(fd, fmt, ap);
  va_end(ap);
}
// Lazily built optional accelerators (coded column stream, SpMM tiles) set their state to -1 ("not usable") before they
// start: a build that fails -- typically hipMalloc on a full device -- must not fail the user's product, the other kernels
// are still there.  The sticky HIP error of the failed call is cleared.  Only an out-of-memory failure (as recorded at
// the failing call, note_hip_error) is silent:
// anything else (a launch error, an inconsistency a builder detects) is a defect of the builder and is reported on stderr and
// counted (khip_test_optional_build_failures: the GPU test session asserts the count is zero) -- api.cpp.
void optional_build(int rc);
inline void klog_flush(int fd) { if (fd <= 0) fflush(stdout); }

#define KHIP_CHECK_HIP(expr)                                                              \
  do {                                                                                    \
    hipError_t e__ = (expr);                                                              \
    if (e__ != hipSuccess) {                                                              \
      ::khip::note_hip_error((int)e__);                                                   \
      ::khip::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, \
                        __LINE__);                                                        \
      return KHIP_ERR_HIP;                                                                \
    }                                                                                     \
  } while (0)

#define KHIP_REQUIRE(cond, ...)          \
  do {                                   \
    if (!(cond)) {                       \
      ::khip::set_error(__VA_ARGS__);    \
      return KHIP_ERR_INVALID;           \
    }                                    \
  } while (0)

#define KHIP_TRY(expr)                 \
  do {                                 \
    int rc__ = (expr);                 \
    if (rc__ != KHIP_OK) return rc__;  \
  } while (0)

// double-double partial of a compensated reduction: value = hi + lo
struct dd {
  double hi, lo;
};

constexpr int kMaxNout = 4;           // outputs one reduction launch may produce (dot2 = 2)
constexpr int kMaxRedOut = 64;        // scalars one all-reduce call may carry (multi-GPU)
constexpr int kResultSlots = 256;     // device-resident scalar ring (chained MGS coefficients)

struct Comm;   // comm.cpp

struct Tuning {
  int spmv_kernel = 0;      // 0 = auto, 1 = stream (LDS-staged), 2 = vector (sub-wave per row), 3 = ordered sub-wave
  int spmv_rows = 256;      // rows per workgroup of the stream kernel
  int spmv_vec = 1;         // nnz per lane per load in the stream kernel (1, 2)
  int spmv_nt = 0;          // non-temporal loads for the val/col streams (measured slower on MI355X)
  int spmv_xcd = 0;         // XCD-aware tile remap: R > 0 runs of R tiles per XCD, -1 contiguous eighths, 0 off
  int spmv_sweep_s = 0;     // plane sweep (spmv_xcd = -2): tiles per grid plane (0 = from the handle's band width)
  int spmv_sweep_w = 16;    // plane sweep: consecutive tiles per XCD column
  int spmv_nty = -1;        // y store of the SpMV kernels: -1 = non-temporal when y is >= 512 MiB (beyond the Infinity Cache), else plain; 0 plain, 1 non-temporal, 2 write-through (sc1)
  int spmv_dot_early = 0;   // staged kernels with a fused dot: load dotw[row] before the row block's windows
  int spmv_fake_gather = 0; // tuning experiment (wrong results): coalesced x reads
  int spmv_tiles = 1;       // staged kernel: consecutive row blocks per workgroup
  int spmv_template = 1;    // use the row-template kernel on handles that khip_csr_compress compressed
  int spmv_tmpl_rows = 8;   // template kernel: rows per lane (amortises the per-workgroup table load; 1.02 -> 0.86 ms at 512^3)
  int spmv_cap = 0;         // staged kernel LDS window in entries (0 = sized to the widest row block)
  int spmv_lds_pad = 0;     // experiment: extra dynamic LDS bytes per workgroup (lowers occupancy)
  int spmv_blockptr = 1;    // use the L2-resident block-pointer table in the stream kernel
  int spmv_pipe = 0;        // staged kernel: software-pipelined form with this many consecutive row blocks per workgroup (0 = one block per workgroup, no pipeline; measured no faster: profiles/r02b_sweep_pipe.log)
  int cg_setup_fused = 1;   // cg! (fused paths, M = I, no warm start): x = 0, r = p = b, gamma = b.b in one pass (khip_cg_setup) instead of four primitives
  int spmv_sell = 2;        // coded operators with 8-bit codes: the sliced (64-row transposed) form -- every lane loads its own row's entries with coalesced 8-byte loads, no LDS window, no barrier in the row walk (1: default load policy, 2: non-temporal loads of the matrix words, 0: off = spmv_code_kernel); 512^3: 2.21 -> 1.97-2.00 ms fused, CG 275 -> 290-297 it/s (profiles/r06ap, r06aq)
  int spmv_sell_pair = 1;   // sliced form of a coded operator: the row's words in 16-byte pairs (half the vector-memory instructions of the matrix stream): 7-point 512^3 fused 1.92-2.00 -> 1.77-1.84 ms, 27-point 216^3 plain 0.53 -> 0.49 ms; 2 = for the int32 form too (96 instead of 88 B per 7-point row: no gain, off)
  int spmv_sell_narrow = 0; // sliced form: 1 = 4-bit codes in one 32-bit word per row where the operator allows (<= 15 diagonals, <= 8 entries per row): 60 instead of 64 B per 7-point row -- and 8-10 % SLOWER at 512^3 (fused 2.13-2.15 against 1.93-1.98 ms: slices of 7 units are no longer 4 KB blocks, and the codes are a second stream of 256-byte wave loads; profiles/r06au_spmv_sell_narrow_ab.log): off
  int spmv_stream_nt = 0;   // 16-byte-load stream kernel: matrix stream loaded with the non-temporal policy
  int spmv_blk_pub = 0;     // fused dots of the staged / coded / delta SpMV kernels: 1 = workgroup-level fold in LDS, one wave runs the double-double tree (block_publish); measured equal to the per-wave trees (profiles/r04b_sweep_headline.log): off
  int spmv_delta = 0;       // stream kernel: block-delta column stream (coldelta.hip: 1 or 2 B per entry + 6 B per escape) -- 0: never (default: 18 % fewer bytes but no faster on the banded + random operator, slower on stencils; profiles/r04a_sweep_delta.log); 1: operators of >= 4 M entries where it saves at least a sixth of the column bytes; 2: whatever the size; 8 / 16: that width, always
  int spmv_wide = 0;        // stream kernel: 1 = the 16-byte-load form (spmv_delta_kernel<int32_t>) where the delta stream is not used (measured equal on the irregular operator, 8 % slower on the 27-point one); 0: the 8 + 4 byte loads of spmv_stream_kernel
  int spmv_codes = 1;       // staged kernel: stream dictionary-coded columns (1 or 2 B per entry) when the operator has <= 2048 diagonals -- 1: for operators of >= 4 M entries; 2: whatever the size; 16: two-byte codes; 0: plain int32 columns
  int spmv_lanes = 0;       // vector kernel lanes per row (0 = auto)
  int spmv_persist = 0;     // 1 = persistent grid (<= 8 workgroups per CU) instead of one row block per workgroup
  int compensated = 1;      // Dot2 (TwoSum/TwoProd) reductions
  int nt_min_elems = 1 << 22;  // BLAS-1 vectors at least this long use non-temporal accesses (32 MiB)
  int mgs_keep = -1;        // MGS cascade: keep q and the freshly dotted basis vector cacheable for the next step (-1 auto: when they fit the Infinity Cache; 0 off; 1 on; 2 nothing streamed)
  int hist_window = 1 << 14;   // device-resident loops: residual-history entries kept on the device between drains
  int red_u = 0;            // reductions: 16-byte accesses per lane (0 auto: 4 for long vectors, else 1)
  int ilu_blocks = 1;        // ILU(0) solves: block schedule where the pattern is a structured grid (0: level scheduling always; 2: block schedule on packed entry lists only, no row records; 3: blocks from the level-sorted row sequence even where a grid is recognised)
  int panel_multi_tiles = 2; // X += sum V_i Y_i: factor blocks in LDS, this many 16-row tiles per wave (0 = the one-tile kernel that re-reads the factors per tile)
  int gmres_sstep = 4;      // gmres! variant 2: inner iterations per block of the s-step form (1..8)
  int panel_nt = 0;         // fused Gram-Schmidt kernel: non-temporal accesses to the panels (1: panels of at least nt_min_elems doubles, 2: always); measured no effect (3.263 vs 3.263 ms per three-panel sweep), off
  int panel_a_lds = 1;      // fused Gram-Schmidt step at p = 16: the A operand (V_i tile) by two coalesced loads per lane + an LDS transpose instead of four loads that each touch all 16 lines of the tile
  int panel_qr_tsqr = 0;    // panel QR: R factor by TSQR (block Householder QRs out of LDS, tree of triangles) instead of the Cholesky of the Gram matrix: any conditioning, no shifted pass (block.cpp)
  int panel_signs = 1;      // panel QR: LAPACK's Householder signs / tau from the top p x p block (block.cpp); 0 = positive diagonal of R
  int panel_fuse = 1;       // block Gram-Schmidt: apply Psi_i and form Psi_{i+1} in one pass (panel.hip, single rank)
  int spmm_wide = 1;        // SpMM: two panel columns per lane (16-byte gathers) when p is even
  int spmm_window_grid = 0; // workgroups of the persistent window kernel (0 = CUs x LDS-limited residency)
  int spmm_tile = 1;        // SpMM with 16 columns: wave-private LDS windows filled by LDS-DMA, grid-tile row groups (spmm_tile.hip)
  int spmm_tile_exp = 0;    // experiments on the tile kernel (WRONG results): 1 no panel-row copies, 2 no products, 4 no (val, slot) loads, 8 round-robin XCD order
  int spmm_tile_waves = 0;  // persistent waves of the tile kernel = this multiple of the LDS-limited residency (0 = 1)
  int spmm_tile_grid = 0;   // ... or this many waves outright
  int spmm_tile_pair = 1;   // two waves per window (spmm_tile2_kernel, p >= 16): twice the waves per CU on the same LDS, each wave half of a group's row passes; 0: one wave per window
  int spmm_tile_dbuf = -1;  // two windows per wave, the copies of the next group overlap this group's products: -1 = where six waves per CU still fit (small windows), 0 never, 1 always
  int spmm_tile_xcd = -1;   // which XCD takes which groups of the tile SpMM: 0 = XCD x the x-th eighth (neighbouring groups share an L2: grid tiles), 1 = round-robin over all persistent workgroups (ONE front over the matrix: long-range columns find their panel rows in the Infinity Cache; banded + random 1.90 -> 1.55 ms, profiles/r06h_spmm_xcd_order.log), 2 = one front in chunks (per sweep of G groups XCD x takes the x-th chunk of G / 8: within 1 % of 1 without a grid, worse on grid tiles), -1 = 1 for handles without a grid and without runs, else 0
  int spmm_tile_ahead = 0;  // sliding windows with ONE GROUP OF LOOK-AHEAD (round 6): the new panel rows of group g + 1 take window slots that neither g nor g + 1 needs, so their copies are issued BEFORE the products of g and land while they run (windows of |g| + |new(g + 1)| rows: 216 instead of 144 at cfg 5); -1 = where the records have runs and two waves share a window; 0 never; 1 = as -1 (read at build time for the window size, at launch for the loop)
  int spmm_tile_slide = -1; // sliding windows: a wave (pair) walks a run of groups along the slowest grid direction and copies only the panel rows its window does not hold yet; -1 = runs of <= 27 groups on grid operators, none otherwise; 0 never; 1 whole grid lines; N > 1 runs of N groups (also without a grid)
  int spmm_tile_pencil = 0; // tile rows per pencil in the group order of grid operators (0 = 4)
  int spmm_tile_slices = 0; // panels of 32 columns and more as 16-column slices through the p = 16 tile kernel: 0 = where it measures faster, 1 = always, -1 = never
  int spmm_tile_nt = 0;     // non-temporal hints on the record / entry / Y streams of the tile kernel
  int spmm_tile_shape = 0;  // grid tile of a 32-row group at build time: 0 = 4x4x2, 1 = 8x2x2, 2 = 2x4x4, 3 = 4x2x4, 4 = 8x4x1, 5 = 32x1x1
  int spmm_window = 1;      // SpMM: stage the distinct panel rows of a row group in LDS (csr_aux.hip, spmm_window_kernel)
  int spmm_sweep = 0;       // SpMM direct kernel: plane-sweep tile order when the band is wider than the L2 can hold (csr_aux.hip; measured no faster)
  int spmm_win_sweep = 0;   // SpMM window kernel: plane-sweep order of the row groups (each XCD walks columns of spmm_sweep_w groups through all planes); measured 15 % slower although it removes the re-fetches (profiles/r02_spmm_experiments.log)
  int spmm_sweep_s = 0;     // tiles per plane (0 = from the handle's band width)
  int spmm_sweep_w = 64;    // tiles per XCD column
  int overlap_halo = 1;     // overlap halo exchange with interior rows
  int halo_mode = 0;        // distributed operators: 1 = exchange only the needed remote x entries (neighbour Send/Recv), 2 = all-gather x before every product, 0 = choose per operator (gather when a rank needs more than halo_gather_pct % of its own row count from its peers)
  int halo_gather_pct = 50; // see halo_mode
  int halo_self = 0;        // TEST / MEASUREMENT hook: with a ONE-rank communicator, a row slab [row0, row0 + m) of a larger operator gets a real halo plan whose off-slab columns wrap onto the slab's own rows (the slab becomes periodic) and are exchanged with grouped ncclSend / ncclRecv to the rank ITSELF: pack kernel, grouped call on the halo stream, interior / boundary split, 16-byte all-gather + combine all run as on N ranks (tools/slab_iteration.py).  Set BEFORE khip_comm_init
  int comm_priority = 1;    // 1: the communication stream has the highest stream priority (default; env KHIP_COMM_PRIORITY=0 starts with 0); 0: default priority
  int profile_spmv = 0;     // record HIP events around every SpMV launch (bench.py roofline leg)
};

// Device-resident loop control attached to the launches of a "fused = 2" solve (solver_device.hpp).
struct SeqCtl {
  const long long *stop_seq = nullptr;   // device word; kernels with seq >= *stop_seq return immediately
  long long seq = 0;
  int epi = 0;                           // scalar epilogue run by the kernel that finalises a reduction
  void *epi_state = nullptr;
};

}  // namespace khip

struct khip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipStream_t comm_stream = nullptr;     // = comm_stream_hi or comm_stream_lo (ctx option comm_priority; both exist for an A/B inside one process)
  hipStream_t comm_stream_hi = nullptr, comm_stream_lo = nullptr;
  // halo exchange hand-off events main stream <-> comm stream: a ring, so that an exchange enqueued while
  // earlier ones are still pending (device-resident loops run several iterations ahead) never re-records
  // an event something still waits on
  static constexpr int kEvRing = 16;
  hipEvent_t ev_a[kEvRing] = {}, ev_b[kEvRing] = {};
  unsigned ev_cur = 0;
  hipEvent_t ev_fetch = nullptr;       // results_copy_begin / _end (look-ahead fetch of device scalars)
  hipEvent_t ev_red = nullptr;         // comm_allreduce_dd_device_begin / _end (all-reduce on the communication stream)
  bool allreduce_pending = false;      // a _begin on this context still waits for its _end
  int num_cu = 256;
  // reduction scratch (grown on demand by ensure_reduction_scratch)
  khip::dd *partials = nullptr;        // [kMaxNout][red_cap1]  one per wave of the streaming kernel
  khip::dd *partials2 = nullptr;       // [kMaxNout][256]       one per workgroup of the finish kernel
  unsigned *tickets = nullptr;         // ticket word (+ spare words)
  int64_t red_cap1 = 0;
  int *scratch_word = nullptr;         // spare device word (row statistics)
  double *results = nullptr;           // device scalar ring [kResultSlots]   (value)
  khip::dd *results_dd = nullptr;      // device scalar ring [kResultSlots]   (hi, lo) for multi-GPU
  double *results_pinned = nullptr;    // pinned host mirror [kResultSlots * 2]
  int next_slot = 0;
  khip::Tuning tune;
  khip::SeqCtl ctl;                    // empty except inside a device-resident solver loop
  khip::Comm *comm = nullptr;
  void *panel_scratch = nullptr;       // panel.hip: V^T Q partial tiles + Psi staging ring
  // SpMV launch profiling (events recorded on `stream`, resolved lazily)
  std::vector<hipEvent_t> prof_events;   // pairs: start, stop
  std::vector<int> prof_tags;            // one per pair: which kernel family the bracket belongs to (ProfTag)
  int prof_spmv_tag = 0;                 // tag of the next SpMV launch's bracket (kProfSpmv; kProfSpmvBoundary for the boundary rows)
  size_t prof_used = 0;
};

struct khip_csr {
  khip_ctx *ctx = nullptr;
  int64_t m = 0, n = 0, nnz = 0;       // local rows, (global) cols, local nnz
  int32_t *rowptr = nullptr;           // device, m+1, 0-based
  int32_t *blockptr = nullptr;         // device, ceil(m/256)+1 : rowptr[min(256 i, m)]
  int32_t *col = nullptr;              // device, nnz (+pad), 0-based; remapped when distributed
  double *val = nullptr;               // device, nnz (+pad)
  int64_t max_row_nnz = 0;
  int64_t band = 0;                    // max |column - row| (local indices)
  int64_t plane_rows = 0;              // estimated distance (rows) between the outermost coupling planes of a 3-D operator (0 = unknown; csr_finalize)
  int64_t line_rows = 0;               // estimated distance (rows) between neighbouring grid lines (0 = unknown; csr_finalize)
  double mean_row_nnz = 0;
  // distributed state (null / zero when single GPU)
  bool dist = false;
  int64_t n_global = 0, row0 = 0;
  int64_t n_ghost = 0;
  double *ghost = nullptr;             // device, n_ghost : received remote x entries
  double *sendbuf = nullptr;           // device, n_send
  int32_t *send_idx = nullptr;         // device, n_send : owned indices to pack
  double *ghost_w = nullptr, *sendbuf_w = nullptr;   // panel versions (n_ghost / n_send rows of halo_w_cap doubles)
  int halo_w_cap = 0;
  int64_t n_send = 0;
  std::vector<int64_t> send_off, recv_off;   // per-peer offsets (size nranks+1)
  std::vector<int64_t> part_starts;          // row partition of the global operator (size nranks+1), kept for khip_csr_transpose
  std::vector<int32_t> ghost_gid;            // neighbour mode: global column of every ghost slot (ascending), kept for khip_csr_transpose
  // gather mode (comm.cpp): x is all-gathered before the product instead of exchanging the needed entries only
  bool self_halo = false;                    // ctx option "halo_self": the one rank exchanges its halo with itself
  bool gather = false;
  int64_t gather_maxm = 0;                   // slice stride of the receive buffer (largest local row count)
  std::vector<int64_t> gather_rows;          // local row count of every rank
  int64_t interior_lo = 0, interior_hi = 0;  // rows [lo,hi) reference no ghost column
  // optional row-template compression (template.hip): one 16-bit template id per row + a small table
  uint16_t *tmpl_id = nullptr;
  int32_t *tmpl_off = nullptr;         // [T][K] column - row
  double *tmpl_val = nullptr;          // [T][K]
  int32_t *tmpl_cnt = nullptr;         // [T]
  int tmpl_T = 0, tmpl_K = 0;
  // optional SpMM panel-row window (csr_aux.hip, built on the first SpMM that can use it)
  int win_L = 0;                       // lanes per row the metadata was built for; -L = tried, not usable; 0 = not tried
  int32_t *win_list = nullptr;         // [groups][STRIDE(L)] distinct columns of a row group, padded with 0
  int32_t *win_flag = nullptr;         // per row group: 1 = direct-gather group (too many panel rows or nonzeros for the window)
  uint16_t *win_slot = nullptr;        // per nonzero: position of its column in the group's list
  // optional group records of the p = 16 tile SpMM (spmm_tile.hip, built on the first SpMM with 16 columns)
  int tile_state = 0;                  // 0 = not tried, 1 = built, -1 = tried, not usable
  char *tile_meta = nullptr;           // [tile_groups] records of tile_stride bytes
  int32_t *tile_direct_list = nullptr; // [tile_direct] groups on the direct-gather path
  int tile_cap = 0, tile_stride = 0, tile_grid = 0;   // window size (panel rows); tile_grid: 1 = groups are grid tiles, 0 = consecutive rows
  int tile_run_len = 0;                // sliding windows: groups per run (0: every group fills its window anew)
  int tile_ahead = 0;                  // 1: the records carry look-ahead flags (aux word of row slot 3) and the window has room for them
  int64_t tile_runs = 0;
  int64_t tile_groups = 0, tile_direct = 0;           // tile_direct: groups on the direct-gather path
  double tile_reuse = 0;               // references per distinct column of a group
  // optional dictionary-coded column stream of the staged SpMV (colcode.hip, built on the first product that can use it)
  int code_state = 0;                  // 0 = not tried, 1 = built, -1 = tried, not usable (too many distinct diagonals)
  int code_bits = 0;                   // 8 or 16
  int code_T = 0;                      // distinct (column - row) offsets
  void *code = nullptr;                // uint8_t / uint16_t [nnz + pad]
  int32_t *code_tab = nullptr;         // [code_T], ascending
  // optional SLICED form of the coded operator (colcode.hip csr_build_sell, built on the first product that can use it): the entries of
  // every 64 consecutive rows transposed so that lane l of a wave finds entry k of ITS row at word (W + k) * 64 + l of the slice --
  // W = ceil(L / 8) words of eight 1-byte codes per row first (0xFF = no entry), then L = longest row of the slice words of values
  int sell_state = 0;                  // 0 = not tried, 1 = built, -1 = tried, not usable / not worth the padding
  unsigned long long *sell = nullptr;  // 64-bit words, 64 per unit
  uint32_t *sell_off = nullptr;        // [slices + 1] first unit of every slice (null: every slice has sell_units units)
  int sell_units = 0;                  // uniform slices: W + L of every slice (0: per-slice offsets)
  int64_t sell_total_units = 0;
  int sell32_pair = 0;                 // the int32 form in 16-byte pairs (2: general pair layout, see sell_pair)
  int sell_pair = 0;                   // 2: general pair layout -- head words (codes / columns) padded to an even count, then the values padded to an even count, lane l's words 2e, 2e + 1 one 16-byte element; 1: rows of at most 8 entries, the words of a row (code word, values) interleaved in PAIRS: lane l's words 2e, 2e + 1 are one 16-byte element at (e * 64 + l) of the slice -- 16-byte lane loads
  uint32_t *sell_c4 = nullptr;         // narrow codes (at most 15 diagonals, rows of at most 8 entries): ONE 32-bit word of eight 4-bit codes per row (0xF = no entry), indexed by the row; the slices then hold values only
  // ... and the same for the int32 column stream (operators that are not coded, or spmv_codes = 0): W = ceil(L / 2) words of two
  // int32 columns per row (-1 = no entry), then L values
  int sell32_state = 0;
  unsigned long long *sell32 = nullptr;
  uint32_t *sell32_off = nullptr;
  int sell32_units = 0;
  int64_t sell32_total_units = 0;
  // optional block-delta column stream of the stream SpMV (coldelta.hip, built on the first product that can use it)
  int delta_state = 0;                 // 0 = not tried, 1 = built, -1 = tried, not usable / not worth it
  int delta_bits = 0;                  // 8 or 16
  int delta_rows = 0;                  // rows per block (the kernel's row block)
  int64_t delta_esc = 0;               // escapes (entries kept as int32 columns)
  void *dcode = nullptr;               // uint8_t / uint16_t [nnz + pad]
  int32_t *dbase = nullptr;            // [blocks]
  int32_t *desc_ptr = nullptr;         // [blocks + 1]
  uint16_t *desc_pos = nullptr;        // [delta_esc]
  int32_t *desc_col = nullptr;         // [delta_esc]
};

namespace khip {

// ---- slot ring -------------------------------------------------------------
inline int take_slots(khip_ctx *ctx, int count) {
  if (ctx->next_slot + count > kResultSlots) ctx->next_slot = 0;
  int s = ctx->next_slot;
  ctx->next_slot += count;
  return s;
}

// blas1.hip
int results_copy_begin(khip_ctx *ctx, int slot, int count);
int results_copy_end(khip_ctx *ctx, int count, double *out_host);
int launch_divcopy_dev(khip_ctx *ctx, int64_t n, const double *sumsq_dev, const double *x, double *y);   // y = x / sqrt(*sumsq)
int ensure_reduction_scratch(khip_ctx *ctx, int64_t nwaves, int nout);
int launch_finish(khip_ctx *ctx, int64_t nwaves, int nout, int slot);
int launch_dot(khip_ctx *ctx, int64_t n, const double *x, const double *y, int slot);
int launch_nrm2sq(khip_ctx *ctx, int64_t n, const double *x, int slot);
int launch_dot2(khip_ctx *ctx, int64_t n, const double *x, const double *y, int slot);  // slot: x.y, slot+1: x.x
int launch_axpy2_dot(khip_ctx *ctx, int64_t n, double a, const double *p, const double *q, double *x,
                     double *r, int slot);
int launch_axpy_sqnorm(khip_ctx *ctx, int64_t n, double a, const double *x, double *y, int slot);   // y += a x ; y.y
int launch_cg_setup(khip_ctx *ctx, int64_t n, const double *b, double *x, double *r, double *p, int slot);   // x = 0 ; r = p = b ; b.b
int launch_cg_update(khip_ctx *ctx, int64_t n, double a, double b, const double *r, double *p, double *x);
// same with alpha / beta / solved read from a CgDevState in device memory (solver_device.hpp)
int launch_cg_update_dev(khip_ctx *ctx, int64_t n, const void *cg_state_dev, long long seq, const double *r, double *p,
                         double *x);
// run the scalar epilogue in ctx->ctl on results[slot..] (1-thread kernel; used when the reduction result was
// produced outside a finish kernel) / fold nranks gathered (hi, lo) partials per scalar and run it
int launch_epilogue_only(khip_ctx *ctx, int slot);
int launch_combine(khip_ctx *ctx, const dd *gathered_dev, int nranks, int count, int slot, hipStream_t stream = nullptr);
// single-reduction CG: p, s, x, r updated in one pass with alpha / beta from a CgcgDevState
int launch_cgcg_update(khip_ctx *ctx, int64_t n, const void *st_dev, long long seq, const double *w, double *r, double *p,
                       double *s, double *x);
// pipelined CG (Ghysels-Vanroose): z, s, p, x, r, w updated in one pass with alpha / beta from a CgcgDevState
int launch_pcg_update(khip_ctx *ctx, int64_t n, const void *st_dev, long long seq, const double *q, double *z, double *s, double *p,
                      double *x, double *r, double *w);
// fused elementwise passes of one bicgstab! iteration (blas1.hip)
// st_dev != null: alpha / omega / beta and the stop word are read from a BicgDevState (solver_device.hpp)
int launch_bicg_sx(khip_ctx *ctx, int64_t n, double alpha, const double *r, const double *v, const double *y, double *s,
                   double *x, const void *st_dev = nullptr, long long seq = 0);
int launch_bicg_xr(khip_ctx *ctx, int64_t n, double omega, const double *s, const double *t, const double *z,
                   const double *c, double *x, double *r, int slot, const void *st_dev = nullptr);   // slot: c.r, slot+1: r.r
int launch_bicg_p(khip_ctx *ctx, int64_t n, double omega, double beta, const double *v, const double *r, double *p,
                  const void *st_dev = nullptr, long long seq = 0);
// y <- y - (*coef_dev) x ; out[slot] = z . y (z == y -> ||y||^2), coef read from device memory
int launch_axpy_dev_dot(khip_ctx *ctx, int64_t n, const double *coef_dev, const double *x, double *y,
                        const double *z, int slot);
int launch_map(khip_ctx *ctx, int op, int64_t n, double a, double b, const double *x, double *y,
               double *w);
int launch_multi_axpy(khip_ctx *ctx, int64_t n, int k, const double *coef_host,
                      const double *const *V_host, double *x);
// classical Gram-Schmidt pieces (gmres! variant 1): results[slot + i] = V_i . q (four per launch); x -= sum coef_dev[j] V_j
int launch_multi_dot(khip_ctx *ctx, int64_t n, int k, const double *const *V_host, const double *q, int slot);
int launch_multi_axpy_dev(khip_ctx *ctx, int64_t n, int k, const double *coef_dev, const double *const *V_host, double *x);
// fetch `count` results starting at slot into host memory (synchronises the stream; all-reduces
// across ranks when a communicator is attached).
// already_global: results[slot..] were all-reduced on the device (comm_allreduce_dd_device): plain copy
int fetch_results(khip_ctx *ctx, int slot, int count, double *out_host, bool already_global = false);

// spmv.hip
// dot_slot >= 0: results[dot_slot] = dotw . y (dotw = x when null); dot_sq: also results[dot_slot + 1] = y . y
int launch_spmv(khip_ctx *ctx, const khip_csr *A, const double *x, double *y, int dot_slot /* -1 = none */,
                int64_t row_lo, int64_t row_hi, int64_t *wave_cursor = nullptr, bool finish = true,
                const double *dotw = nullptr, int dot_sq = 0,      // dot_sq: 0 none, 1 second output y.y, 2 second output dotw.dotw
                int64_t hole_lo = 0, int64_t hole_hi = 0);         // hole_hi > hole_lo: rows [hole_lo, hole_hi) are left out (two ranges, one launch where the kernel can)
int spmv_kernel_choice(const khip_ctx *ctx, const khip_csr *A);
int launch_spmm(khip_ctx *ctx, const khip_csr *A, const double *X, double *Y, int p);
int csr_finalize(khip_ctx *ctx, khip_csr *A);   // row statistics after arrays are resident
int csr_transpose(khip_ctx *ctx, const khip_csr *A, khip_csr *T);   // T = A' (fresh handle, deterministic entry order)
int launch_gather(khip_ctx *ctx, int64_t n, const int32_t *idx, const double *x, double *out, int width = 1);
int launch_col_remap(khip_ctx *ctx, khip_csr *A, const int32_t *ghost_sorted_dev, int64_t n_ghost);
int launch_col_remap_gather(khip_ctx *ctx, khip_csr *A, const int64_t *row_starts_dev, int nranks, int64_t maxm);
int launch_collect_offrank(khip_ctx *ctx, const khip_csr *A, int64_t row0, int64_t row1, int32_t *out_dev,
                           unsigned long long *count_dev, int64_t cap);
int launch_row_ghost_range(khip_ctx *ctx, const khip_csr *A, int64_t *lo_hi_host);
int launch_index_shift(khip_ctx *ctx, int32_t *data, int64_t n, int32_t delta);
int launch_diagonal(khip_ctx *ctx, const khip_csr *A, double *diag);

// template.hip
void csr_free_templates(khip_csr *A);
void csr_free_window(khip_csr *A);
void csr_free_tiles(khip_csr *A);                  // spmm_tile.hip
int spmm_tile_build(khip_ctx *ctx, khip_csr *A);   // spmm_tile.hip: sets A->tile_state to 1 or -1
void csr_free_delta(khip_csr *A);                  // coldelta.hip
int csr_build_delta(khip_ctx *ctx, khip_csr *A, int rows);   // coldelta.hip: sets A->delta_state to 1 or -1
void csr_free_codes(khip_csr *A);                  // colcode.hip
int csr_build_codes(khip_ctx *ctx, khip_csr *A);   // colcode.hip: sets A->code_state to 1 or -1
void csr_free_sell(khip_csr *A);                   // colcode.hip
int csr_build_sell(khip_ctx *ctx, khip_csr *A);    // colcode.hip: sets A->sell_state to 1 or -1 (needs 8-bit codes)
void csr_free_sell32(khip_csr *A);                 // colcode.hip
int csr_build_sell32(khip_ctx *ctx, khip_csr *A);  // colcode.hip: the sliced form with int32 columns; sets A->sell32_state to 1 or -1
int panel_multi_nn(khip_ctx *ctx, int64_t n, int p, int k, const double *const *V_host, const double *Y_host, double beta,
                   double *X);   // panel.hip
int panel_tsqr_r(khip_ctx *ctx, int64_t n, int p, const double *Q, double *R_host_rowmajor);   // panel.hip: R factor by TSQR
int panel_scale_gram(khip_ctx *ctx, int64_t n, int p, double *Q, const double *Ri_host, double *G_host);   // panel.hip
int panel_mgs_gram(khip_ctx *ctx, int64_t n, int p, int k, const double *const *V_host, double *Q, double *Psi_host, int accumulate,
                   double *G_host, bool *have_gram);   // panel.hip: khip_panel_mgs that also returns Q^T Q of the swept panel
int panel_fill_columns(khip_ctx *ctx, int64_t n, int p, double *Q, unsigned mask, double scale, unsigned long long seed);   // panel.hip

// api.cpp: the MGS cascade of khip_mgs in two halves (enqueue: launches only; the k coefficients and ||q||^2 end up in
// results[slot .. slot + k], all-reduced on the device when there are several ranks)
int mgs_enqueue(khip_ctx *ctx, int64_t n, int k, const double *const *V_host, double *q, int *slot_out);

// api.cpp: y = A x (dot_slot >= 0: also results[dot_slot] = x . y) incl. halo exchange; launches only, no host sync
int spmv_any(khip_ctx *ctx, const khip_csr *A, const double *x, double *y, int dot_slot, const double *dotw = nullptr,
             int dot_sq = 0);

// panel.hip
void panel_scratch_destroy(khip_ctx *ctx);

// comm.cpp
int comm_nranks(const khip_ctx *ctx);
int comm_rank_of(const khip_ctx *ctx);
int comm_allreduce_dd(khip_ctx *ctx, dd *vals_dev, int count, double *out_host);
// all-reduce results_dd[slot..slot+count) into results[slot..] ON THE DEVICE (no host sync with RCCL), then run ctx->ctl's epilogue
int comm_allreduce_dd_device(khip_ctx *ctx, int slot, int count);
// the same in two halves, for recurrences that put work between the reduction and its first use (pipelined CG): `begin`
// issues the all-gather and the combine (+ epilogue) on the COMMUNICATION stream behind an event of the main stream and
// returns; `end` makes the main stream wait for them.  What is enqueued on the main stream in between overlaps with the
// collective.  Falls back to the in-order form (all in `begin`) where there is no second communicator to keep the two streams'
// collectives apart, with the in-process backend, and without a communicator.
int comm_allreduce_dd_device_begin(khip_ctx *ctx, int slot, int count);
int comm_allreduce_dd_device_end(khip_ctx *ctx);
int comm_halo_exchange_begin(khip_ctx *ctx, const khip_csr *A, const double *x, int width = 1);   // width p: row-major panel
int comm_allreduce_sum_host(khip_ctx *ctx, double *vals, int count);
int comm_halo_exchange_end(khip_ctx *ctx, const khip_csr *A);
int comm_transpose_dist(khip_ctx *ctx, const khip_csr *A, khip_csr **out);   // A' of a row-partitioned handle, same partition
int comm_build_plan(khip_ctx *ctx, khip_csr *A, int local_rc = 0);   // local_rc != 0 / A == null: this rank failed before the plan; every rank then returns an error

// "time limit exceeded" decided COLLECTIVELY: every stopping test of the solver loops comes from all-reduced scalars and
// is therefore identical on all ranks -- except the wall clock.  A rank that alone ran out of time would leave its peers
// blocked in the next all-gather / Send / Recv, so with a communicator attached the flag is summed over the ranks (any
// rank over the limit stops all of them, at the same iteration).  Costs a host all-gather per test, hence only when the
// caller set a finite timemax.  All ranks reach every call site together (same control flow), as the collective needs.
inline bool time_limit_reached(khip_ctx *ctx, double elapsed_s, double timemax) {
  if (!(timemax < 1e300)) return false;                 // no limit (the default): no collective either
  double over = elapsed_s > timemax ? 1.0 : 0.0;
  if (comm_nranks(ctx) > 1 && comm_allreduce_sum_host(ctx, &over, 1) != KHIP_OK) return true;   // a broken communicator ends the solve
  return over > 0.0;
}

// rows of the GLOBAL operator (= n on one GPU): a distributed handle knows it, otherwise the local counts are summed
inline int64_t global_rows(khip_ctx *ctx, const khip_operator *A, int64_t n_local) {
  if (A && A->csr && !A->apply && A->csr->dist) return A->csr->n_global;
  if (comm_nranks(ctx) > 1) {
    double v = (double)n_local;
    if (comm_allreduce_sum_host(ctx, &v, 1) == KHIP_OK) return (int64_t)v;
  }
  return n_local;
}


// HIP-event brackets around single kernel launches on the context's stream (ctx option "profile_spmv" = 1; khip_profile_spmv /
// khip_profile_kernels read them): bench.py's roofline figures are averages of these, measured inside the timed solve.
enum ProfTag { kProfSpmv = 0, kProfSpmm = 1, kProfPanelTn = 2, kProfPanelNnTn = 3, kProfPanelMultiNn = 4, kProfPanelNn = 5, kProfPanelQr = 6,
               kProfHaloPack = 7,       // pack kernel of the neighbour exchange (main stream)
               kProfHaloXfer = 8,       // grouped ncclSend / ncclRecv (or the all-gather of x), on the stream it runs on
               kProfDotGather = 9,      // 16-byte all-gather of a dot's (hi, lo) partials + combine kernel
               kProfSpmvBoundary = 10,  // the boundary rows' launch of a row-partitioned product (after the halo has arrived)
               kProfTags = 11 };
struct ProfScope {
  khip_ctx *ctx;
  hipEvent_t stop = nullptr;
  hipStream_t stream = nullptr;
  ProfScope(khip_ctx *c, int tag, hipStream_t st = nullptr) : ctx(c) {
    stream = st ? st : c->stream;
    if (!c->tune.profile_spmv) return;
    if (c->prof_used + 2 > c->prof_events.size()) {
      for (int i = 0; i < 64; ++i) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        c->prof_events.push_back(e);
      }
    }
    if (c->prof_tags.size() < c->prof_events.size() / 2) c->prof_tags.resize(c->prof_events.size() / 2, 0);
    if (hipEventRecord(c->prof_events[c->prof_used], stream) != hipSuccess) return;
    stop = c->prof_events[c->prof_used + 1];
    c->prof_tags[c->prof_used / 2] = tag;
    c->prof_used += 2;
  }
  ~ProfScope() { if (stop) (void)hipEventRecord(stop, stream); }
};
}  // namespace khip
