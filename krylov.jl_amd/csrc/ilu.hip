// ilu.hip -- ILU(0) / IC(0) preconditioner resident on the device: factorisation and the two sparse
// triangular solves of M^{-1} = U^{-1} L^{-1}  (SURVEY.md section 8f, row N1).
//
// What it replaces: the reference has no such code of its own -- its GPU examples and tests build the
// preconditioner with the vendor library (`ic02` / `ilu02` + `ldiv!` on triangular views,
// docs/src/gpu.md:74-163, test/gpu/nvidia.jl:37-100: IC(0)-CG on sparse_laplacian(16) must converge in
// <= 19 iterations) and hand it to cg!/bicgstab!/gmres! as the `M` / `N` operator (src/cg.jl:160,241).
// Here the same operator is a khip_operator (khip_ilu0_create) on the pattern of a khip_csr.
//
// Algorithm: ILU(0), IKJ variant (Saad, alg. 10.4); for SPD A its factors are L and U = D L^T, i.e. the
// IC(0) preconditioner in exact arithmetic.  Parallelism: LEVEL SCHEDULING.  Row i of the lower solve
// depends on the rows j < i of its pattern; level(i) = 1 + max level(j).  Rows of one level are
// independent: one kernel launch per level, one lane per row, and every lane walks its row in stored order
// with one rounded multiply and one rounded subtract per entry -- the serial CPU loop, so factors and
// solves are BIT-IDENTICAL to the serial restatement the tests compare against (ko_ilu0 / ko_ilu0_solve).
// The level analysis (setup, once per pattern) runs on the host from the downloaded index arrays;
// the whole apply (2 x #levels launches) is captured once into a hipGraph and replayed.
// A 7-point grid in natural ordering has n1+n2+n3-2 levels (hyperplanes): level-scheduled solves are
// launch-latency bound, not HBM bound -- see DESIGN.md for measured numbers.  Where the pattern IS such a grid
// (5- / 7-point stencils) the solves therefore run the BLOCK SCHEDULE further down (ilu_block_solve_kernel):
// one persistent launch per triangle, blocks of 8 x 8 x 8 grid points solved out of LDS, per-block flags.
//
// Distributed handles: block-Jacobi ILU(0) of the rank's diagonal block (ghost columns are ignored), the
// usual domain-decomposition preconditioner; it needs no communication.
#include <algorithm>
#include <map>
#include <functional>
#include <thread>
#include <unistd.h>

#include "khip_internal.hpp"

namespace khip {

constexpr int kIluBlock = 256;

// Row i of the factor occupies lu[row_lo[i] .. row_hi[i]) with the diagonal at diag[i]; for a
// non-distributed handle these are rowptr[i], rowptr[i+1]; for a distributed one the owned, sorted part.
struct IluView {
  const int32_t *col;
  const int32_t *row_lo, *diag, *row_hi;
  double *lu;
};

__global__ __launch_bounds__(kIluBlock) void ilu0_factor_level_kernel(IluView v, const int32_t *perm, int64_t lo,
                                                                        int64_t hi, int *bad_row) {
  const int64_t idx = lo + (int64_t)blockIdx.x * kIluBlock + threadIdx.x;
  if (idx >= hi) return;
  const int32_t i = perm[idx];
  const int32_t re = v.row_hi[i], di = v.diag[i];
  for (int32_t kk = v.row_lo[i]; kk < di; ++kk) {
    const int32_t k = v.col[kk];
    const int32_t dk = v.diag[k];
    const double piv = v.lu[dk];
    if (piv == 0.0) { atomicMin(bad_row, (int)k); return; }
    const double lik = v.lu[kk] / piv;
    v.lu[kk] = lik;
    int32_t p = kk + 1;
    const int32_t ke = v.row_hi[k];
    for (int32_t q = dk + 1; q < ke; ++q) {
      const int32_t j = v.col[q];
      while (p < re && v.col[p] < j) ++p;
      if (p < re && v.col[p] == j) {
        const double t = lik * v.lu[q];
        v.lu[p] = v.lu[p] - t;
      }
    }
  }
  if (v.lu[di] == 0.0) atomicMin(bad_row, (int)i);
}

// y[i] = x[i] - sum_{q in [row_lo, diag)} lu[q] * y[col[q]]          (unit lower triangle)
__global__ __launch_bounds__(kIluBlock) void trsv_lower_level_kernel(IluView v, const int32_t *perm, int64_t lo,
                                                                       int64_t hi, const double *x, double *y) {
  const int64_t idx = lo + (int64_t)blockIdx.x * kIluBlock + threadIdx.x;
  if (idx >= hi) return;
  const int32_t i = perm[idx];
  double acc = x[i];
  const int32_t di = v.diag[i];
  for (int32_t q = v.row_lo[i]; q < di; ++q) {
    const double t = v.lu[q] * y[v.col[q]];
    acc = acc - t;
  }
  y[i] = acc;
}

// y[i] = (y[i] - sum_{q in (diag, row_hi)} lu[q] * y[col[q]]) / lu[diag]
__global__ __launch_bounds__(kIluBlock) void trsv_upper_level_kernel(IluView v, const int32_t *perm, int64_t lo,
                                                                       int64_t hi, double *y) {
  const int64_t idx = lo + (int64_t)blockIdx.x * kIluBlock + threadIdx.x;
  if (idx >= hi) return;
  const int32_t i = perm[idx];
  double acc = y[i];
  const int32_t di = v.diag[i], re = v.row_hi[i];
  for (int32_t q = di + 1; q < re; ++q) {
    const double t = v.lu[q] * y[v.col[q]];
    acc = acc - t;
  }
  y[i] = acc / v.lu[di];
}

// Runs of consecutive SMALL levels (<= kIluBlock rows each: the tips of a stencil's wavefront pyramid, or every level of
// a chain-like operator such as a tridiagonal matrix) are executed by ONE workgroup that steps through them with a
// workgroup barrier in between: one launch instead of one per level.  lvl = device copy of the level pointers.
template <int KIND>     // 0: factorisation, 1: lower solve, 2: upper solve
__global__ __launch_bounds__(kIluBlock) void ilu_small_levels_kernel(IluView v, const int32_t *perm, const int64_t *lvl, int l0,
                                                                      int l1, const double *x, double *y, int *bad_row) {
  for (int l = l0; l < l1; ++l) {
    const int64_t idx = lvl[l] + threadIdx.x;
    if (idx < lvl[l + 1]) {
      const int32_t i = perm[idx];
      const int32_t re = v.row_hi[i], di = v.diag[i];
      if (KIND == 0) {
        bool dead = false;
        for (int32_t kk = v.row_lo[i]; kk < di && !dead; ++kk) {
          const int32_t k = v.col[kk];
          const int32_t dk = v.diag[k];
          const double piv = v.lu[dk];
          if (piv == 0.0) { atomicMin(bad_row, (int)k); dead = true; break; }
          const double lik = v.lu[kk] / piv;
          v.lu[kk] = lik;
          int32_t p = kk + 1;
          const int32_t ke = v.row_hi[k];
          for (int32_t q = dk + 1; q < ke; ++q) {
            const int32_t j = v.col[q];
            while (p < re && v.col[p] < j) ++p;
            if (p < re && v.col[p] == j) {
              const double t = lik * v.lu[q];
              v.lu[p] = v.lu[p] - t;
            }
          }
        }
        if (!dead && v.lu[di] == 0.0) atomicMin(bad_row, (int)i);
      } else if (KIND == 1) {
        double acc = x[i];
        for (int32_t q = v.row_lo[i]; q < di; ++q) {
          const double t = v.lu[q] * y[v.col[q]];
          acc = acc - t;
        }
        y[i] = acc;
      } else {
        double acc = y[i];
        for (int32_t q = di + 1; q < re; ++q) {
          const double t = v.lu[q] * y[v.col[q]];
          acc = acc - t;
        }
        y[i] = acc / v.lu[di];
      }
    }
    __threadfence_block();
    __syncthreads();
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Block schedule of the triangular solves (operators whose pattern is a structured grid in natural ordering).
//
// Level scheduling pays one kernel boundary per level -- 2 (n1 + n2 + n3 - 2) of them for a 7-point grid, ~7 us each: the
// solves are launch-latency bound.  Here the rows are cut into BLOCKS of a few hundred rows (8 x 8 x 8
// lattice points of a basis in which all dependencies point towards smaller coordinates -- the grid's axes for the 5- / 7-point
// stencils, a skewed basis for the 9- / 27-point ones, see detect_grid; 16 x 16 on a 2-D grid), so that a block depends on
// blocks with smaller block coordinates only.  One persistent launch per triangle: a workgroup of ONE wave takes the next
// block off a ticket counter (tickets run along the block wavefronts bx + by + bz, a topological order of the block graph),
// copies the block's rows to LDS -- 48-byte records for rows with up to 3 entries, 176-byte records up to 16, packed (value,
// slot) lists beyond -- while it waits for the blocks it depends on, fetches the y values of their faces, and then walks
// the block's OWN levels out of LDS: a dependent step costs an LDS round trip instead of a kernel boundary, and device-wide
// traffic (a flag per block, y written through and read past the L2s) happens once per block, not once per level.  Every row is
// still computed by one lane that walks its entries in stored order with a rounded multiply and a rounded subtract per
// entry: y is bit-identical to the level-scheduled kernels and to the oracle's serial loops (ko_ilu0_solve).
// No deadlock: a block waits only for blocks with SMALLER tickets; tickets are handed out in order to running workgroups,
// so the smallest unfinished ticket never waits.  The spins are bounded all the same (a.fail is set instead of hanging).
#ifndef KHIP_ILU_SPIN
#define KHIP_ILU_SPIN (1 << 22)
#endif
// -DKHIP_ILU_TRACE: shader-clock stamps of the phases of 64 consecutive blocks of the lower solve (tools/archive/ilu_trace.py)
#ifdef KHIP_ILU_TRACE
__device__ unsigned long long g_ilu_trace[64 * 8];
#define ILU_STAMP(slot) do { if (KIND == 1 && lane == 0 && t >= a.nb / 2 && t < a.nb / 2 + 64) g_ilu_trace[(t - a.nb / 2) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ILU_STAMP(slot) do { } while (0)
#endif
constexpr int kBlkRows = 512;          // rows of a block at most
constexpr int kBlkThreads = 64;        // one wave: its levels need no s_barrier between waves

// Row record of the fast path (every row of the triangle has at most 3 off-diagonal entries: the 5- and 7-point stencils):
// 48 bytes = v0, v1, v2, pivot, {slot0, slot1, slot2, count} as 4 x u16, the row's number in y -- one LDS round trip of three
// 16-byte reads brings everything of a row that does not depend on y.
constexpr int kRecDoubles = 6;
constexpr int kWideDoubles = 22;      // wide row record (rows with 4..16 entries): 16 values, 16 x u16 slots, pivot, row number = 176 bytes
constexpr int kRecExtCap = 256;        // fast path: ext slots of the LDS y array; the last two are a constant 0.0 and a dump
typedef double dbl2 __attribute__((ext_vector_type(2)));

struct IluBlockHdr {
  int64_t ent0;                        // first packed entry of the block
  int32_t row0, nrows;                 // its rows: row_gid[row0 .. row0 + nrows), in local level order
  int32_t nent;
  int32_t lvl0, nlvl;                  // local level pointers lvl[lvl0 .. lvl0 + nlvl]
  int32_t ext0, next;                  // rows of other blocks whose y it reads
  int32_t dep0, ndep;                  // tickets of the blocks it waits for
  int32_t pad;
};

struct IluBlkArgs {
  const IluBlockHdr *hdr;
  const int32_t *row_gid;              // [n]
  const uint16_t *row_eptr;            // [n + nb]: per block nrows + 1 local entry offsets, at row0 + ticket
  const uint16_t *lvl;
  const int32_t *ext_gid, *dep;
  const uint16_t *ent_slot;            // per packed entry: local row (< rows_cap) or rows_cap + index into the block's ext list
  const double *ent_val;
  const double *diag_val;              // upper solve: lu[diag] per row, in block order
  const double *rec;                   // row records (kRecDoubles per row, block order) where every row has <= 3 entries, else null
  const double *recw;                  // wide row records (kWideDoubles per row) where every row has <= 16 entries, else null
  int *done;                           // [nb]: epoch of the last solve that finished the block
  unsigned *ticket;
  int *fail;                           // device word: a wait of THIS solve gave up (checked by the other waits of the same solve)
  int *fail_host;                      // pinned host mirror (device pointer to mapped host memory): read by the host without a sync
  int nb, max_ent, max_ext, max_lvl;
  int rows_cap;                        // rows of the largest block, rounded up to the wave: the LDS arrays and the slot numbers are laid out for it
};

template <int KIND>     // 1: lower solve y = L^{-1} x, 2: upper solve y = U^{-1} y
__global__ __launch_bounds__(kBlkThreads) void ilu_block_solve_kernel(IluBlkArgs a, const double *x, double *y, int epoch,
                                                                       unsigned ticket_base) {
  extern __shared__ double ilu_sm[];
  __shared__ int s_ticket;
  const int RC = a.rows_cap;                                // <= kBlkRows
  double *yl = ilu_sm;                                      // [RC + max_ext + 1]: y of the block's rows, of the ext list, a spare
  double *xv = yl + RC + a.max_ext + 1;                     // [RC]: right-hand side
  double *dv = xv + RC;                                     // [RC]: pivots (upper solve)
  double *ev = dv + (KIND == 2 ? RC : 0);                   // [max_ent]
  uint16_t *es = reinterpret_cast<uint16_t *>(ev + a.max_ent);   // [max_ent rounded up to 4]
  uint16_t *ep = es + ((a.max_ent + 3) & ~3);               // [RC + 2]
  uint16_t *lv = ep + RC + 2;                               // [max_lvl + 2]
  int32_t *gl = reinterpret_cast<int32_t *>(lv + ((a.max_lvl + 2 + 1) & ~1));      // [RC]: row numbers (general path)
  const int lane = threadIdx.x;
  for (;;) {
    // lane 0 draws the ticket; every lane reads it back from LDS
    if (lane == 0) s_ticket = (int)(atomicAdd(a.ticket, 1u) - ticket_base);
    __syncthreads();
    const int t = __builtin_amdgcn_readfirstlane(s_ticket);
    __syncthreads();
    if (t >= a.nb) break;
    const IluBlockHdr h = a.hdr[t];
    ILU_STAMP(0);
    const bool fast = a.rec != nullptr && h.pad != 0;        // row records, and no level wider than the wave
    const bool wide = !fast && a.recw != nullptr && h.pad != 0;
    int my_lv = 0;                                           // record paths: lane l holds level pointer l
    // stage what does not depend on other blocks
    if (fast || wide) {
      // kBlkRows x 48 (176) bytes at most, as 16-byte pieces: all loads of a batch are in flight before the first LDS store
      const int rd = fast ? kRecDoubles : kWideDoubles;
      const dbl2 *src = reinterpret_cast<const dbl2 *>((fast ? a.rec : a.recw) + (int64_t)h.row0 * rd);
      dbl2 *dst = reinterpret_cast<dbl2 *>(ev);
      const int pieces = h.nrows * (rd / 2);
      for (int b0 = 0; b0 < pieces; b0 += 8 * kBlkThreads) {
        dbl2 tmp[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = b0 + u * kBlkThreads + lane;
          tmp[u] = src[i < pieces ? i : 0];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = b0 + u * kBlkThreads + lane;
          if (i < pieces) dst[i] = tmp[u];
        }
      }
      my_lv = a.lvl[h.lvl0 + (lane <= h.nlvl ? lane : 0)];
      int32_t gid[kBlkRows / kBlkThreads];
#pragma unroll
      for (int u = 0; u < kBlkRows / kBlkThreads; ++u) {
        const int r = u * kBlkThreads + lane;
        gid[u] = a.row_gid[h.row0 + (r < h.nrows ? r : 0)];       // no branch around a load: all of them are in flight together
      }
      double xs[kBlkRows / kBlkThreads];
#pragma unroll
      for (int u = 0; u < kBlkRows / kBlkThreads; ++u) xs[u] = KIND == 1 ? x[gid[u]] : y[gid[u]];
#pragma unroll
      for (int u = 0; u < kBlkRows / kBlkThreads; ++u) {
        const int r = u * kBlkThreads + lane;
        if (r < h.nrows) xv[r] = xs[u];
      }
    } else {
      // batches of 16 elements per lane: all loads of a batch are in flight before its first LDS store (one element per trip
      // made the staging of a 27-point block, 6656 entries, the longest phase of the block)
      for (int b0 = 0; b0 < h.nent; b0 += 16 * kBlkThreads) {
        double tv[16];
        uint16_t ts[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int e = b0 + u * kBlkThreads + lane;
          const int64_t g = h.ent0 + (e < h.nent ? e : 0);
          tv[u] = a.ent_val[g];
          ts[u] = a.ent_slot[g];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int e = b0 + u * kBlkThreads + lane;
          if (e < h.nent) { ev[e] = tv[u]; es[e] = ts[u]; }
        }
      }
      for (int r = lane; r <= h.nrows; r += kBlkThreads) ep[r] = a.row_eptr[(int64_t)h.row0 + t + r];
      for (int l = lane; l <= h.nlvl; l += kBlkThreads) lv[l] = a.lvl[h.lvl0 + l];
      for (int r = lane; r < h.nrows; r += kBlkThreads) {
        const int32_t gid = a.row_gid[h.row0 + r];
        gl[r] = gid;
        xv[r] = KIND == 1 ? x[gid] : y[gid];
        if (KIND == 2) dv[r] = a.diag_val[h.row0 + r];
      }
    }
    // the row numbers of the faces it will read: fetched now, so that one round trip (the y values) is left after the wait
    constexpr int kExtRegs = 16;
    int32_t eg[kExtRegs];
#pragma unroll
    for (int u = 0; u < kExtRegs; ++u) {
      const int k = u * kBlkThreads + lane;
      // lanes past the face list (all of them when the block has no external dependency: ext_gid may then be empty) read
      // the block's own first row instead of an entry that does not exist
      eg[u] = k < h.next ? a.ext_gid[h.ext0 + k] : a.row_gid[h.row0];
    }
    ILU_STAMP(1);
    // the blocks this one reads from
    for (int d = lane; d < h.ndep; d += kBlkThreads) {
      const int *flag = a.done + a.dep[h.dep0 + d];
      int spins = 0;
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > KHIP_ILU_SPIN || __hip_atomic_load(a.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch) {
          // give up: mark THIS solve (the word holds the epoch, so a past failure does not cut later solves' waits short)
          // and tell the host, which checks at the next application / block_info and falls back to level scheduling
          __hip_atomic_store(a.fail, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (a.fail_host) __hip_atomic_store(a.fail_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          break;
        }
      }
    }
    __syncthreads();
    ILU_STAMP(2);
    asm volatile("" ::: "memory");                           // the loads below stay below the spins
    // y of other blocks: written through and read past the L2s (agent scope), so that neither side needs a cache-wide
    // write-back or invalidate per block
    {
      double ye[kExtRegs];
#pragma unroll
      for (int u = 0; u < kExtRegs; ++u) ye[u] = __hip_atomic_load(y + eg[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int u = 0; u < kExtRegs; ++u) {
        const int k = u * kBlkThreads + lane;
        if (k < h.next) yl[RC + k] = ye[u];
      }
      for (int k = kExtRegs * kBlkThreads + lane; k < h.next; k += kBlkThreads)       // more than 1024 face rows
        yl[RC + k] = __hip_atomic_load(y + a.ext_gid[h.ext0 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    ILU_STAMP(3);
    if (fast) {
      // one row per lane and level; the row's record and right-hand side are read one level AHEAD (they do not depend on y),
      // so a level costs one dependent LDS round trip: the (at most three) y values
      const dbl2 *recs = reinterpret_cast<const dbl2 *>(ev);
      if (lane == 0) yl[RC + kRecExtCap - 2] = 0.0;
      auto row_of = [&](int l) {                                    // this lane's row of level l, or -1
        const int r = __builtin_amdgcn_readlane(my_lv, l) + lane;
        return r < __builtin_amdgcn_readlane(my_lv, l + 1) ? r : -1;
      };
      int r = row_of(0);
      int rr = r >= 0 ? r : 0;
      dbl2 c0 = recs[rr * 3], c1 = recs[rr * 3 + 1], c2 = recs[rr * 3 + 2];
      double rhs = xv[rr];
      for (int l = 0; l < h.nlvl; ++l) {
        const int rn = l + 1 < h.nlvl ? row_of(l + 1) : -1;
        const int rrn = rn >= 0 ? rn : 0;
        const dbl2 n0 = recs[rrn * 3], n1 = recs[rrn * 3 + 1], n2 = recs[rrn * 3 + 2];      // next level's row
        const double nrhs = xv[rrn];
        const unsigned long long meta = (unsigned long long)__double_as_longlong(c2.x);
        const double y0 = yl[meta & 0xffff], y1 = yl[(meta >> 16) & 0xffff], y2 = yl[(meta >> 32) & 0xffff];
        // absent entries are (value 0.0, slot of a constant 0.0): acc - 0.0 * 0.0 == acc bit for bit (also for -0.0 and NaN),
        // so the three steps need no selects
        double acc = rhs;
        const double t0 = c0.x * y0;
        acc = acc - t0;
        const double t1 = c0.y * y1;
        acc = acc - t1;
        const double t2 = c1.x * y2;
        acc = acc - t2;
        const double yv = KIND == 2 ? acc / c1.y : acc;
        yl[r >= 0 ? r : RC + kRecExtCap - 1] = yv;              // idle lanes write the dump slot: no branch
        // ... and through the L2 to y at once: the stores are in flight while the remaining levels run, instead of a separate
        // pass over the block at the end whose latency sits on the critical path of the block wavefronts
        if (r >= 0) __hip_atomic_store(y + (long long)__double_as_longlong(c2.y), yv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // no s_barrier, no s_waitcnt: the workgroup is ONE wave and the LDS executes a wave's instructions in order, so the
        // reads of the next level see this write; only the compiler must not move them across it
        asm volatile("" ::: "memory");
        r = rn; c0 = n0; c1 = n1; c2 = n2; rhs = nrhs;
      }
    } else if (wide) {
      // rows with 4..16 entries: the same walk as above on 176-byte records -- 16 values (absent: 0.0), 16 slots (absent: the
      // slot of a constant 0.0), pivot, row number -- read with eleven 16-byte LDS loads one level ahead
      const dbl2 *recs = reinterpret_cast<const dbl2 *>(ev);
      const int zero_slot = RC + a.max_ext - 2, dump_slot = RC + a.max_ext - 1;
      if (lane == 0) yl[zero_slot] = 0.0;
      auto row_of = [&](int l) {
        const int r = __builtin_amdgcn_readlane(my_lv, l) + lane;
        return r < __builtin_amdgcn_readlane(my_lv, l + 1) ? r : -1;
      };
      struct WRow { dbl2 c[kWideDoubles / 2]; double rhs; };
      auto fetchw = [&](int rr, WRow &w) {
#pragma unroll
        for (int q = 0; q < kWideDoubles / 2; ++q) w.c[q] = recs[rr * (kWideDoubles / 2) + q];
        w.rhs = xv[rr];
      };
      int r = row_of(0);
      WRow cur;
      fetchw(r >= 0 ? r : 0, cur);
      for (int l = 0; l < h.nlvl; ++l) {
        const int rn = l + 1 < h.nlvl ? row_of(l + 1) : -1;
        WRow nxt;
        fetchw(rn >= 0 ? rn : 0, nxt);
        const unsigned long long sw[4] = {(unsigned long long)__double_as_longlong(cur.c[8].x), (unsigned long long)__double_as_longlong(cur.c[8].y),
                                          (unsigned long long)__double_as_longlong(cur.c[9].x), (unsigned long long)__double_as_longlong(cur.c[9].y)};
        double yy[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) yy[u] = yl[(sw[u >> 2] >> (16 * (u & 3))) & 0xffff];
        double acc = cur.rhs;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const double tt = ((u & 1) ? cur.c[u >> 1].y : cur.c[u >> 1].x) * yy[u];
          acc = acc - tt;
        }
        const double yv = KIND == 2 ? acc / cur.c[10].x : acc;
        yl[r >= 0 ? r : dump_slot] = yv;
        if (r >= 0) __hip_atomic_store(y + (long long)__double_as_longlong(cur.c[10].y), yv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" ::: "memory");
        r = rn;
        cur = nxt;
      }
    } else {
      for (int l = 0; l < h.nlvl; ++l) {
        const int r1 = lv[l + 1];
        for (int r = lv[l] + lane; r < r1; r += kBlkThreads) {
          // the row's first 16 (value, slot) pairs are read together, then the y values they point to, then the products are
          // subtracted one after the other in stored order (the selects keep absent entries out); longer rows go on one by one
          double acc = xv[r];
          const int e0 = ep[r], cnt = ep[r + 1] - e0;
          constexpr int kRowBatch = 16;
          double v[kRowBatch], yy[kRowBatch];
          int sl[kRowBatch];
#pragma unroll
          for (int u = 0; u < kRowBatch; ++u) {
            const int e = u < cnt ? e0 + u : 0;
            v[u] = ev[e];
            sl[u] = es[e];
          }
#pragma unroll
          for (int u = 0; u < kRowBatch; ++u) yy[u] = yl[sl[u]];
#pragma unroll
          for (int u = 0; u < kRowBatch; ++u) {
            const double tt = v[u] * yy[u];
            const double an = acc - tt;
            acc = u < cnt ? an : acc;
          }
          for (int e = e0 + kRowBatch; e < e0 + cnt; ++e) {
            const double tt = ev[e] * yl[es[e]];
            acc = acc - tt;
          }
          const double yv = KIND == 2 ? acc / dv[r] : acc;
          yl[r] = yv;
          __hip_atomic_store(y + gl[r], yv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // through the L2, in flight during the next levels
        }
        __syncthreads();
      }
    }
    ILU_STAMP(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the block's y has arrived before its flag is raised
    ILU_STAMP(5);
    __syncthreads();
    if (lane == 0) __hip_atomic_store(a.done + t, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();      // a convergent operation between this `if (lane == 0)` and the one at the top of the loop: without it
                          // the two are threaded into a private loop of lane 0 and the other lanes re-run block t for ever
  }
}

// row records of the fast path: src4 = positions in lu of the row's (up to three) entries and of its pivot (-1: none),
// meta = {slot0, slot1, slot2, count}
__global__ __launch_bounds__(256) void ilu_pack_records_kernel(const double *lu, const int32_t *src4, const unsigned long long *meta,
                                                               const int32_t *row_gid, int64_t rows, double *rec) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows) return;
  double *o = rec + i * kRecDoubles;
  for (int k = 0; k < 3; ++k) o[k] = src4[4 * i + k] >= 0 ? lu[src4[4 * i + k]] : 0.0;
  o[3] = lu[src4[4 * i + 3]];
  o[4] = __longlong_as_double((long long)meta[i]);
  o[5] = __longlong_as_double((long long)row_gid[i]);          // the row's number in y: the level loop writes y as it goes
}

// wide row records from the packed entry lists: one wave per block
__global__ __launch_bounds__(kBlkThreads) void ilu_pack_wide_kernel(const IluBlockHdr *hdr, const uint16_t *row_eptr, const double *ent_val,
                                                                    const uint16_t *ent_slot, const double *lu, const int32_t *diag,
                                                                    const int32_t *row_gid, int zero_slot, double *recw) {
  const int t = blockIdx.x;
  const IluBlockHdr h = hdr[t];
  for (int r = threadIdx.x; r < h.nrows; r += kBlkThreads) {
    const int e0 = row_eptr[(int64_t)h.row0 + t + r], e1 = row_eptr[(int64_t)h.row0 + t + r + 1];
    double *o = recw + ((int64_t)h.row0 + r) * kWideDoubles;
    unsigned long long sw[4] = {0, 0, 0, 0};
    for (int u = 0; u < 16; ++u) {
      const bool on = e0 + u < e1;
      o[u] = on ? ent_val[h.ent0 + e0 + u] : 0.0;
      const unsigned long long sl = on ? ent_slot[h.ent0 + e0 + u] : (unsigned)zero_slot;
      sw[u >> 2] |= sl << (16 * (u & 3));
    }
    for (int q = 0; q < 4; ++q) o[16 + q] = __longlong_as_double((long long)sw[q]);
    const int32_t gid = row_gid[h.row0 + r];
    o[20] = lu[diag[gid]];
    o[21] = __longlong_as_double((long long)gid);
  }
}

// ent_val[e] = lu[src[e]] (and the pivots of the upper solve) after the numeric factorisation
__global__ __launch_bounds__(256) void ilu_pack_values_kernel(const double *lu, const int32_t *src, int64_t n, double *out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = lu[src[i]];
}

}  // namespace khip

using namespace khip;

struct khip_ilu0 {
  khip_ctx *ctx = nullptr;
  const khip_csr *A = nullptr;
  int64_t n = 0, nnz = 0;
  double *lu = nullptr;
  int32_t *row_lo = nullptr, *diag = nullptr, *row_hi = nullptr;
  int32_t *perm_lo = nullptr, *perm_up = nullptr;
  std::vector<int64_t> lvl_lo, lvl_up;       // level pointers into perm_lo / perm_up
  int64_t *d_lvl_lo = nullptr, *d_lvl_up = nullptr;   // device copies (for the batched small levels)
  int *bad_row = nullptr;
  // block schedule (structured grids; see ilu_block_solve_kernel): one per triangle
  struct Blocks {
    int nb = 0, max_ent = 0, max_ext = 0, max_lvl = 0, grid = 0, rows_cap = kBlkRows;
    IluBlockHdr *hdr = nullptr;
    int32_t *row_gid = nullptr, *ext_gid = nullptr, *dep = nullptr;
    uint16_t *row_eptr = nullptr, *lvl = nullptr, *ent_slot = nullptr;
    double *ent_val = nullptr, *diag_val = nullptr, *rec = nullptr, *recw = nullptr;
    bool want_wide = false;
    int *done = nullptr;
    unsigned *ticket = nullptr;
    int epoch = 0;
    unsigned ticket_base = 0;
    size_t lds = 0;
  } blk_lo, blk_up;
  int *blk_fail = nullptr;
  int *blk_fail_host = nullptr, *blk_fail_host_dev = nullptr;   // pinned mirror of "a wait gave up" and its device address
  int64_t grid_dims[3] = {0, 0, 0};          // detected grid (0: none, level scheduling)
  int grid_skew[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};   // lattice basis of the block coordinates
  bool use_blocks = false;
  // cached hipGraph of one application, keyed by the (x, y) pointers it was captured with
  hipGraphExec_t graph = nullptr;
  const double *gx = nullptr;
  double *gy = nullptr;
  bool use_graph = true;
};

namespace {

IluView view_of(const khip_ilu0 *P) { return IluView{P->A->col, P->row_lo, P->diag, P->row_hi, P->lu}; }

unsigned grid_for(int64_t rows) { return (unsigned)((rows + kIluBlock - 1) / kIluBlock); }

// all levels of one triangle: runs of small levels in one single-workgroup launch, every other level its own launch
template <int KIND>
int enqueue_levels(khip_ilu0 *P, const std::vector<int64_t> &lvl, const int64_t *d_lvl, const int32_t *perm, const double *x,
                   double *y) {
  khip_ctx *ctx = P->ctx;
  const IluView v = view_of(P);
  const int nl = (int)lvl.size() - 1;
  int l = 0;
  while (l < nl) {
    int e = l;
    while (e < nl && lvl[e + 1] - lvl[e] <= kIluBlock) ++e;
    if (e - l >= 2) {                                  // a run of at least two small levels
      hipLaunchKernelGGL((ilu_small_levels_kernel<KIND>), dim3(1), dim3(kIluBlock), 0, ctx->stream, v, perm, d_lvl, l, e, x, y,
                         P->bad_row);
      l = e;
      continue;
    }
    const int64_t lo = lvl[l], hi = lvl[l + 1];
    if (KIND == 0) hipLaunchKernelGGL(ilu0_factor_level_kernel, dim3(grid_for(hi - lo)), dim3(kIluBlock), 0, ctx->stream, v, perm, lo, hi, P->bad_row);
    else if (KIND == 1) hipLaunchKernelGGL(trsv_lower_level_kernel, dim3(grid_for(hi - lo)), dim3(kIluBlock), 0, ctx->stream, v, perm, lo, hi, x, y);
    else hipLaunchKernelGGL(trsv_upper_level_kernel, dim3(grid_for(hi - lo)), dim3(kIluBlock), 0, ctx->stream, v, perm, lo, hi, y);
    ++l;
  }
  KHIP_CHECK_HIP(hipGetLastError());
  return KHIP_OK;
}

int enqueue_solve(khip_ilu0 *P, const double *x, double *y) {
  KHIP_TRY((enqueue_levels<1>(P, P->lvl_lo, P->d_lvl_lo, P->perm_lo, x, y)));
  return enqueue_levels<2>(P, P->lvl_up, P->d_lvl_up, P->perm_up, x, y);
}

template <int KIND>
int launch_blocks(khip_ilu0 *P, khip_ilu0::Blocks &B, const double *x, double *y) {
  IluBlkArgs a{B.hdr, B.row_gid, B.row_eptr, B.lvl, B.ext_gid, B.dep, B.ent_slot, B.ent_val, B.diag_val, B.rec, B.recw, B.done, B.ticket,
               P->blk_fail, P->blk_fail_host_dev, B.nb, B.max_ent, B.max_ext, B.max_lvl, B.rows_cap};
  ++B.epoch;
  hipLaunchKernelGGL((ilu_block_solve_kernel<KIND>), dim3((unsigned)B.grid), dim3(kBlkThreads), B.lds, P->ctx->stream, a, x, y, B.epoch,
                     B.ticket_base);
  B.ticket_base += (unsigned)(B.nb + B.grid);          // every workgroup draws one ticket past the end
  KHIP_CHECK_HIP(hipGetLastError());
  return KHIP_OK;
}

int ilu0_apply(void *self, const double *x, double *y) {
  khip_ilu0 *P = static_cast<khip_ilu0 *>(self);
  if (P->n == 0) return KHIP_OK;
  khip_ctx *ctx = P->ctx;
  if (P->use_blocks && P->blk_fail_host && __atomic_load_n(P->blk_fail_host, __ATOMIC_RELAXED) != 0) {
    // a wait of an earlier application timed out: its result was wrong.  Say so (the solver loop stops on the error), and
    // schedule by levels from now on.
    __atomic_store_n(P->blk_fail_host, 0, __ATOMIC_RELAXED);
    P->use_blocks = false;
    set_error("ilu0: a block of the triangular solves waited too long for the blocks it depends on (result invalid); "
              "this operator falls back to level scheduling");
    return KHIP_ERR_NUMERIC;
  }
  if (P->use_blocks && ctx->tune.ilu_blocks != 0) {
    KHIP_TRY((launch_blocks<1>(P, P->blk_lo, x, y)));
    return launch_blocks<2>(P, P->blk_up, x, y);
  }
  const size_t launches = P->lvl_lo.size() + P->lvl_up.size();
  if (!P->use_graph || launches < 8) return enqueue_solve(P, x, y);
  if (!P->graph || P->gx != x || P->gy != y) {
    if (P->graph) { (void)hipGraphExecDestroy(P->graph); P->graph = nullptr; }
    hipGraph_t g = nullptr;
    KHIP_CHECK_HIP(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
    int rc = enqueue_solve(P, x, y);
    hipError_t e = hipStreamEndCapture(ctx->stream, &g);
    if (rc != KHIP_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
    KHIP_CHECK_HIP(e);
    e = hipGraphInstantiate(&P->graph, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    KHIP_CHECK_HIP(e);
    P->gx = x;
    P->gy = y;
  }
  KHIP_CHECK_HIP(hipGraphLaunch(P->graph, ctx->stream));
  return KHIP_OK;
}

void blocks_free(khip_ilu0::Blocks &B) {
  for (void *p : {(void *)B.hdr, (void *)B.row_gid, (void *)B.ext_gid, (void *)B.dep, (void *)B.row_eptr, (void *)B.lvl,
                  (void *)B.ent_slot, (void *)B.ent_val, (void *)B.diag_val, (void *)B.rec, (void *)B.recw, (void *)B.done, (void *)B.ticket})
    if (p) (void)hipFree(p);
  B = khip_ilu0::Blocks();
}

void ilu0_free(khip_ilu0 *P) {
  if (!P) return;
  if (P->graph) (void)hipGraphExecDestroy(P->graph);
  blocks_free(P->blk_lo);
  blocks_free(P->blk_up);
  if (P->blk_fail) (void)hipFree(P->blk_fail);
  if (P->blk_fail_host) (void)hipHostFree(P->blk_fail_host);
  for (void *p : {(void *)P->lu, (void *)P->row_lo, (void *)P->diag, (void *)P->row_hi, (void *)P->perm_lo,
                  (void *)P->perm_up, (void *)P->bad_row, (void *)P->d_lvl_lo, (void *)P->d_lvl_up})
    if (p) (void)hipFree(p);
  delete P;
}

// counting sort of the rows by level -> perm, level pointers
void sort_by_level(const std::vector<int32_t> &level, int32_t nlev, std::vector<int32_t> &perm, std::vector<int64_t> &ptr) {
  const int64_t n = (int64_t)level.size();
  ptr.assign((size_t)nlev + 1, 0);
  for (int64_t i = 0; i < n; ++i) ptr[(size_t)level[i] + 1]++;
  for (int32_t l = 0; l < nlev; ++l) ptr[(size_t)l + 1] += ptr[l];
  std::vector<int64_t> cur(ptr.begin(), ptr.end() - 1);
  perm.resize((size_t)n);
  for (int64_t i = 0; i < n; ++i) perm[(size_t)cur[level[i]]++] = (int32_t)i;    // rows stay in increasing order inside a level
}

template <typename T>
int upload(khip_ctx *ctx, const std::vector<T> &h, T **dev) {
  KHIP_CHECK_HIP(hipMalloc(dev, sizeof(T) * std::max<size_t>(h.size(), 1)));
  if (!h.empty()) KHIP_CHECK_HIP(hipMemcpy(*dev, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
  else KHIP_CHECK_HIP(hipMemset(*dev, 0, sizeof(T)));          // an empty list still has one addressable, defined element
  return KHIP_OK;
}


// ---- block schedule: host analysis ------------------------------------------------------------------------------------
struct HostPattern {
  int64_t n;
  const std::vector<int32_t> &col, &row_lo, &diag, &row_hi;
};

// Is the pattern a stencil on a grid n1 x n2 x n3 in natural ordering (index = x + n1 (y + n2 z)), and is there a lattice
// basis in which every lower entry of every row points to a grid point with coordinates <= and every upper entry to one
// with coordinates >= ?  Blocks that are cubes in those coordinates then depend on blocks with smaller block coordinates
// only.  For the 5- / 7-point stencils (and their second-neighbour relatives) the grid's own axes do; the 9- / 27-point
// stencils reach (x + 1, y - 1, z) and (x + 1, y + 1, z - 1) in their lower triangle and need the skewed basis
// c = (x + y + 2 z, y + z, z), in which the cubes are parallelepipeds of the grid.
// The candidates for n1 and n1 n2 are the centres of the runs of consecutive row - column offsets of a few sample rows; the
// answer is checked on EVERY entry: its offset must be one of a few (signed digits |dx|, |dy| <= 2, 0 <= dz <= 2, from a
// small table -- no division per entry) and must stay inside the grid (no coupling around the faces).
constexpr int kSkews[2][9] = {{1, 0, 0, 0, 1, 0, 0, 0, 1}, {1, 1, 2, 0, 1, 1, 0, 0, 1}};

bool detect_grid(const HostPattern &H, int64_t dims[3], int skew[9]) {
  const int64_t n = H.n;
  if (n < 4096) return false;
  std::vector<int64_t> offs;
  for (int s = 0; s < 64; ++s) {
    const int64_t i = n / 2 + (n / 3) * (s % 2) - (n / 5) * (s % 3 == 2) + 37 * s;
    if (i < 0 || i >= n) continue;
    for (int32_t q = H.row_lo[(size_t)i]; q < H.diag[(size_t)i]; ++q) offs.push_back(i - H.col[(size_t)q]);
  }
  std::sort(offs.begin(), offs.end());
  offs.erase(std::unique(offs.begin(), offs.end()), offs.end());
  std::vector<int64_t> centre;                       // centres of the runs of consecutive offsets
  for (size_t a = 0; a < offs.size();) {
    size_t b = a;
    while (b + 1 < offs.size() && offs[b + 1] == offs[b] + 1) ++b;
    centre.push_back((offs[a] + offs[b]) / 2);
    a = b + 1;
  }
  auto floordiv = [](int64_t a, int64_t b) { int64_t q = a / b; if ((a % b != 0) && ((a < 0) != (b < 0))) --q; return q; };
  auto valid = [&](int64_t n1, int64_t n2, int64_t n3) {
    if (n1 < 6 || n2 < 6 || n3 < 1 || (n3 > 1 && n3 < 3) || n1 * n2 * n3 != n) return false;
    const int64_t s2 = n1 * n2;
    constexpr int kMaxOffsets = 96;
    int64_t od[kMaxOffsets];
    int odx[kMaxOffsets], ody[kMaxOffsets], odz[kMaxOffsets];
    bool lower_seen[kMaxOffsets], upper_seen[kMaxOffsets];
    int nod = 0;
    int64_t xi = 0, yi = 0, zi = 0;
    for (int64_t i = 0; i < n; ++i) {
      for (int32_t q = H.row_lo[(size_t)i]; q < H.row_hi[(size_t)i]; ++q) {
        const int64_t j = H.col[(size_t)q];
        if (j == i) continue;
        const int64_t d = j < i ? i - j : j - i;
        int k = 0;
        while (k < nod && od[k] != d) ++k;
        if (k == nod) {
          if (nod == kMaxOffsets) return false;            // not a stencil
          const int64_t dz = floordiv(d + s2 / 2, s2), rem = d - dz * s2;
          const int64_t dy = floordiv(rem + n1 / 2, n1), dx = rem - dy * n1;
          if (dz < 0 || dz > 2 || dy < -2 || dy > 2 || dx < -2 || dx > 2) return false;
          od[nod] = d; odx[nod] = (int)dx; ody[nod] = (int)dy; odz[nod] = (int)dz;
          lower_seen[nod] = upper_seen[nod] = false;
          ++nod;
        }
        const int sg = j < i ? -1 : 1;
        const int64_t xj = xi + sg * odx[k], yj = yi + sg * ody[k], zj = zi + sg * odz[k];
        if (xj < 0 || xj >= n1 || yj < 0 || yj >= n2 || zj < 0 || zj >= n3) return false;      // coupling around a face
        (j < i ? lower_seen : upper_seen)[k] = true;
      }
      if (++xi == n1) { xi = 0; if (++yi == n2) { yi = 0; ++zi; } }
    }
    for (const auto &M : kSkews) {
      bool ok = true;
      for (int k = 0; k < nod && ok; ++k) {
        const int v[3] = {odx[k], ody[k], odz[k]};                       // coordinates of j minus those of i, upper entry
        for (int r = 0; r < 3 && ok; ++r) {
          const int c = M[3 * r] * v[0] + M[3 * r + 1] * v[1] + M[3 * r + 2] * v[2];
          if ((upper_seen[k] && c < 0) || (lower_seen[k] && -c > 0)) ok = false;
        }
      }
      if (ok) { for (int e = 0; e < 9; ++e) skew[e] = M[e]; return true; }
    }
    return false;
  };
  int tried = 0;
  for (int64_t n1 : centre) {
    if (n1 < 2 || n % n1 != 0) continue;
    for (int64_t s2 : centre) {                      // 3-D: a second centre that is a multiple of n1 and divides n
      if (s2 <= n1 || s2 % n1 != 0 || n % s2 != 0) continue;
      if (++tried > 6) return false;
      if (valid(n1, s2 / n1, n / s2)) { dims[0] = n1; dims[1] = s2 / n1; dims[2] = n / s2; return true; }
    }
    if (++tried > 6) return false;
    if (valid(n1, n / n1, 1)) { dims[0] = n1; dims[1] = n / n1; dims[2] = 1; return true; }       // 2-D
  }
  return false;
}

struct HostBlocks {            // what the analysis of one triangle produces (host memory; the two triangles are analysed side by side)
  std::vector<IluBlockHdr> hdr;
  std::vector<int32_t> row_gid, ext_gid, dep, src, diag_src, rec_src;
  std::vector<uint16_t> row_eptr, lvl, ent_slot;
  std::vector<unsigned long long> rec_meta;
  bool rec_ok = true;
  int64_t nb = 0;
  int max_ent = 0, max_ext = 0, max_lvl = 0, max_row_ent = 0, rows_cap = kBlkRows, rc = KHIP_OK;
};

// A partition of the rows into blocks and an order of the blocks in which every block depends on earlier ones only.
struct Partition {
  int64_t nslots = 0;
  std::vector<int64_t> slot_ptr;                 // rows of slot b: slot_rows[slot_ptr[b] .. slot_ptr[b + 1]), ascending
  std::vector<int32_t> slot_rows, slot_of_row;
  std::vector<int32_t> order, ticket_of;         // ticket -> slot, slot -> ticket (-1: empty slot)
  int rows_cap = kBlkRows;                       // rows of the largest block, rounded up to a multiple of the wave
};

// cubes of 8 x 8 x 8 lattice points of the basis detect_grid found (16 x 16 on a 2-D grid), in wavefront order
int make_grid_partition(const HostPattern &H, const int64_t dims[3], const int skew[9], bool upper, Partition &P) {
  const int64_t n = H.n, n1 = dims[0], n2 = dims[1], n3 = dims[2];
  // block coordinates: c = skew (x, y, z) (entries >= 0, unimodular: a cube of T^3 lattice points holds T^3 grid points), cut
  // into cubes of T1 x T2 x T3
  const int T1 = n3 > 1 ? 8 : 16, T2 = T1, T3 = n3 > 1 ? 8 : 1;
  auto cmax = [&](int r) { return skew[3 * r] * (n1 - 1) + skew[3 * r + 1] * (n2 - 1) + skew[3 * r + 2] * (n3 - 1); };
  const int64_t B1 = cmax(0) / T1 + 1, B2 = cmax(1) / T2 + 1, B3 = cmax(2) / T3 + 1, nslots = B1 * B2 * B3;
  if (nslots > (int64_t)1 << 30) return KHIP_ERR_INVALID;
  auto slot_of_xyz = [&](int64_t x, int64_t y, int64_t z) {
    const int64_t c1 = skew[0] * x + skew[1] * y + skew[2] * z, c2 = skew[3] * x + skew[4] * y + skew[5] * z,
                  c3 = skew[6] * x + skew[7] * y + skew[8] * z;
    return ((c3 / T3) * B2 + c2 / T2) * B1 + c1 / T1;
  };
  P.nslots = nslots;
  P.rows_cap = std::min(kBlkRows, (T1 * T2 * T3 + kBlkThreads - 1) / kBlkThreads * kBlkThreads);
  // the rows of every block slot, ascending (counting sort by slot; the rows' coordinates from a running counter)
  P.slot_ptr.assign((size_t)nslots + 1, 0);
  P.slot_rows.assign((size_t)n, 0);
  P.slot_of_row.assign((size_t)n, 0);
  {
    int64_t x = 0, y = 0, z = 0;
    for (int64_t i = 0; i < n; ++i) {
      const int64_t sl = slot_of_xyz(x, y, z);
      P.slot_of_row[(size_t)i] = (int32_t)sl;
      P.slot_ptr[(size_t)sl + 1]++;
      if (++x == n1) { x = 0; if (++y == n2) { y = 0; ++z; } }
    }
    for (int64_t b = 0; b < nslots; ++b) P.slot_ptr[(size_t)b + 1] += P.slot_ptr[(size_t)b];
    std::vector<int64_t> cur(P.slot_ptr.begin(), P.slot_ptr.end() - 1);
    for (int64_t i = 0; i < n; ++i) P.slot_rows[(size_t)cur[(size_t)P.slot_of_row[(size_t)i]]++] = (int32_t)i;
  }
  // ticket order: wavefronts of the block grid (mirrored for the upper solve), slot number inside a wavefront; empty slots
  // (the corners of the skewed box) get no ticket
  P.ticket_of.assign((size_t)nslots, -1);
  std::vector<int64_t> cnt((size_t)(B1 + B2 + B3), 0);
  auto wave = [&](int64_t b) {
    const int64_t bz = b / (B1 * B2), r = b - bz * B1 * B2, by = r / B1, bx = r - by * B1;
    return upper ? (B1 - 1 - bx) + (B2 - 1 - by) + (B3 - 1 - bz) : bx + by + bz;
  };
  for (int64_t b = 0; b < nslots; ++b) if (P.slot_ptr[(size_t)b + 1] > P.slot_ptr[(size_t)b]) cnt[(size_t)wave(b) + 1]++;
  for (size_t w = 1; w < cnt.size(); ++w) cnt[w] += cnt[w - 1];
  P.order.assign((size_t)cnt.back(), 0);
  for (int64_t b = 0; b < nslots; ++b) {
    if (P.slot_ptr[(size_t)b + 1] == P.slot_ptr[(size_t)b]) continue;
    const int64_t t = cnt[(size_t)wave(b)]++;
    P.order[(size_t)t] = (int32_t)b;
    P.ticket_of[(size_t)b] = (int32_t)t;
  }
  return KHIP_OK;
}

// No grid: blocks are pieces of the level-sorted row sequence (perm, level pointers lvl) -- a level of 64 rows or more is cut
// into blocks of 64 rows (one local level, one row per lane), consecutive narrower levels are merged into a block while it
// stays within kLevelMergeRows rows and 48 levels.  A row's dependencies lie in earlier levels, i.e. in earlier blocks or in its
// own: the sequence order is topological.  Instead of one kernel boundary per level, a level costs the flag round trip
// between two blocks, and only between blocks that really depend on each other.
constexpr int kLevelMergeRows = 512;    // (128 -- smaller LDS footprint, more workgroups per CU -- was no faster: the path is bound by the chain of flags)

int make_level_partition(int64_t n, const std::vector<int32_t> &perm, const std::vector<int64_t> &lvl, Partition &P) {
  const int nl = (int)lvl.size() - 1;
  P.slot_ptr.assign(1, 0);
  P.slot_rows.assign(perm.begin(), perm.end());
  P.slot_of_row.assign((size_t)n, 0);
  int64_t cur_rows = 0;
  int cur_levels = 0;
  auto close = [&](int64_t end) { if (end > P.slot_ptr.back()) P.slot_ptr.push_back(end); cur_rows = 0; cur_levels = 0; };
  for (int l = 0; l < nl; ++l) {
    const int64_t a = lvl[(size_t)l], b = lvl[(size_t)l + 1], w = b - a;
    if (w >= kBlkThreads) {
      close(a);
      for (int64_t q = a; q < b; q += kBlkThreads) close(std::min<int64_t>(q + kBlkThreads, b));
    } else {
      if (cur_rows + w > kLevelMergeRows || cur_levels >= 48) close(a);
      cur_rows += w;
      ++cur_levels;
    }
  }
  close(n);
  P.nslots = (int64_t)P.slot_ptr.size() - 1;
  if (P.nslots > (int64_t)1 << 30) return KHIP_ERR_INVALID;
  P.order.resize((size_t)P.nslots);
  P.ticket_of.resize((size_t)P.nslots);
  int64_t biggest = 0;
  for (int64_t b = 0; b < P.nslots; ++b) biggest = std::max(biggest, P.slot_ptr[(size_t)b + 1] - P.slot_ptr[(size_t)b]);
  P.rows_cap = (int)std::min<int64_t>(kBlkRows, (biggest + kBlkThreads - 1) / kBlkThreads * kBlkThreads);
  for (int64_t b = 0; b < P.nslots; ++b) {
    P.order[(size_t)b] = P.ticket_of[(size_t)b] = (int32_t)b;
    std::sort(P.slot_rows.begin() + P.slot_ptr[(size_t)b], P.slot_rows.begin() + P.slot_ptr[(size_t)b + 1]);      // ascending inside a block
    for (int64_t q = P.slot_ptr[(size_t)b]; q < P.slot_ptr[(size_t)b + 1]; ++q) P.slot_of_row[(size_t)P.slot_rows[(size_t)q]] = (int32_t)b;
  }
  return KHIP_OK;
}

int analyse_blocks(const HostPattern &H, const Partition &part, bool upper, HostBlocks &hb) {
  std::vector<IluBlockHdr> &hdr = hb.hdr;
  std::vector<int32_t> &row_gid = hb.row_gid, &ext_gid = hb.ext_gid, &dep = hb.dep, &src = hb.src, &diag_src = hb.diag_src, &rec_src = hb.rec_src;
  std::vector<uint16_t> &row_eptr = hb.row_eptr, &lvl = hb.lvl, &ent_slot = hb.ent_slot;
  std::vector<unsigned long long> &rec_meta = hb.rec_meta;
  bool &rec_ok = hb.rec_ok;
  const int64_t n = H.n;
  const std::vector<int64_t> &slot_ptr = part.slot_ptr;
  const std::vector<int32_t> &slot_rows = part.slot_rows, &order = part.order, &ticket_of = part.ticket_of;
  auto block_of = [&](int64_t i) { return (int64_t)part.slot_of_row[(size_t)i]; };
  const int RC = part.rows_cap;
  hb.rows_cap = RC;
  const int64_t nb = (int64_t)order.size();
  hdr.assign((size_t)nb, IluBlockHdr());
  row_gid.reserve((size_t)n);
  row_eptr.reserve((size_t)(n + nb));
  src.clear(); diag_src.clear();
  std::vector<int32_t> lpos((size_t)n, -1), ext_mark((size_t)n, -1);
  std::vector<int32_t> rows, llev, sorted, deps_here;
  std::vector<int32_t> lcount;
  rec_src.reserve((size_t)n * 4);      // fast path: lu positions of a row's <= 3 entries + pivot
  rec_meta.reserve((size_t)n);
  int max_ent = 0, max_ext = 0, max_lvl = 0;
  for (int64_t t = 0; t < nb; ++t) {
    const int64_t b = order[(size_t)t];
    rows.assign(slot_rows.begin() + slot_ptr[(size_t)b], slot_rows.begin() + slot_ptr[(size_t)b + 1]);
    const int nr = (int)rows.size();                       // ascending row numbers
    // local levels from the dependencies inside the block
    llev.assign((size_t)nr, 0);
    for (int k = 0; k < nr; ++k) lpos[(size_t)rows[(size_t)k]] = k;       // temporary: position in ascending order
    int nl = 0;
    for (int kk = 0; kk < nr; ++kk) {
      const int k = upper ? nr - 1 - kk : kk;
      const int32_t i = rows[(size_t)k];
      int lv = 0;
      const int32_t qa = upper ? H.diag[(size_t)i] + 1 : H.row_lo[(size_t)i], qb = upper ? H.row_hi[(size_t)i] : H.diag[(size_t)i];
      for (int32_t q = qa; q < qb; ++q) {
        const int32_t j = H.col[(size_t)q];
        if (lpos[(size_t)j] >= 0) lv = std::max(lv, llev[(size_t)lpos[(size_t)j]] + 1);      // lpos >= 0: a row of this block
      }
      llev[(size_t)k] = lv;
      nl = std::max(nl, lv + 1);
    }
    lcount.assign((size_t)nl + 1, 0);
    for (int k = 0; k < nr; ++k) lcount[(size_t)llev[(size_t)k] + 1]++;
    for (int l = 0; l < nl; ++l) lcount[(size_t)l + 1] += lcount[(size_t)l];
    IluBlockHdr &h = hdr[(size_t)t];
    h.row0 = (int32_t)row_gid.size(); h.nrows = nr; h.ent0 = (int64_t)src.size();
    h.lvl0 = (int32_t)lvl.size(); h.nlvl = nl; h.ext0 = (int32_t)ext_gid.size(); h.dep0 = (int32_t)dep.size(); h.pad = 0;
    for (int l = 0; l <= nl; ++l) lvl.push_back((uint16_t)lcount[(size_t)l]);
    h.pad = nl < kBlkThreads ? 1 : 0;                       // 1: no level wider than the wave, level pointers fit its lanes
    for (int l = 0; l < nl; ++l) if (lcount[(size_t)l + 1] - lcount[(size_t)l] > kBlkThreads) h.pad = 0;
    sorted.assign((size_t)nr, 0);
    {
      std::vector<int32_t> cur(lcount.begin(), lcount.end() - 1);
      for (int k = 0; k < nr; ++k) sorted[(size_t)cur[(size_t)llev[(size_t)k]]++] = rows[(size_t)k];
    }
    for (int k = 0; k < nr; ++k) lpos[(size_t)sorted[(size_t)k]] = k;      // final: position in level order
    deps_here.clear();
    int ne = 0;
    for (int k = 0; k < nr; ++k) {
      const int32_t i = sorted[(size_t)k];
      row_gid.push_back(i);
      row_eptr.push_back((uint16_t)ne);
      diag_src.push_back(H.diag[(size_t)i]);
      const size_t rec_at = rec_src.size();
      rec_src.insert(rec_src.end(), {-1, -1, -1, H.diag[(size_t)i]});
      unsigned long long meta = 0;
      int in_row = 0;
      const int32_t qa = upper ? H.diag[(size_t)i] + 1 : H.row_lo[(size_t)i], qb = upper ? H.row_hi[(size_t)i] : H.diag[(size_t)i];
      for (int32_t q = qa; q < qb; ++q) {                   // stored order
        const int32_t j = H.col[(size_t)q];
        int slot;
        if (lpos[(size_t)j] >= 0) slot = lpos[(size_t)j];
        else {
          if (ext_mark[(size_t)j] < 0) {
            ext_mark[(size_t)j] = (int32_t)(ext_gid.size() - (size_t)h.ext0);
            ext_gid.push_back(j);
            const int32_t tj = ticket_of[(size_t)block_of(j)];
            if (std::find(deps_here.begin(), deps_here.end(), tj) == deps_here.end()) deps_here.push_back(tj);
          }
          slot = RC + ext_mark[(size_t)j];
        }
        ent_slot.push_back((uint16_t)slot);
        src.push_back(q);
        ++ne;
        if (in_row < 3) { rec_src[rec_at + (size_t)in_row] = q; meta |= (unsigned long long)(unsigned)slot << (16 * in_row); }
        ++in_row;
      }
      if (in_row > 3) rec_ok = false;
      hb.max_row_ent = std::max(hb.max_row_ent, in_row);
      for (int kq = in_row; kq < 3; ++kq) meta |= (unsigned long long)(RC + kRecExtCap - 2) << (16 * kq);   // absent: the 0.0 slot
      rec_meta.push_back(meta | (unsigned long long)std::min(in_row, 3) << 48);
    }
    row_eptr.push_back((uint16_t)ne);
    h.nent = ne; h.next = (int32_t)(ext_gid.size() - (size_t)h.ext0); h.ndep = (int32_t)deps_here.size();
    for (int32_t tj : deps_here) {
      if (tj >= t) { set_error("ilu0_create: block schedule is not topological (internal)"); return KHIP_ERR_INVALID; }
      dep.push_back(tj);
    }
    for (int32_t k = h.ext0; k < h.ext0 + h.next; ++k) ext_mark[(size_t)ext_gid[(size_t)k]] = -1;
    for (int k = 0; k < nr; ++k) lpos[(size_t)sorted[(size_t)k]] = -1;
    if (nr > RC || ne > 60000 || RC + h.next > 65535) return KHIP_ERR_INVALID;
    max_ent = std::max(max_ent, ne); max_ext = std::max(max_ext, h.next); max_lvl = std::max(max_lvl, nl);
  }
  hb.nb = nb; hb.max_ent = max_ent; hb.max_ext = max_ext; hb.max_lvl = max_lvl;
  return KHIP_OK;
}

// uploads the analysis of one triangle and builds what needs the factor values (the numeric factorisation is complete)
int upload_blocks(khip_ilu0 *P, bool upper, HostBlocks &hb, khip_ilu0::Blocks &B) {
  khip_ctx *ctx = P->ctx;
  const int64_t n = P->n, nb = hb.nb;
  std::vector<IluBlockHdr> &hdr = hb.hdr;
  std::vector<int32_t> &row_gid = hb.row_gid, &ext_gid = hb.ext_gid, &dep = hb.dep, &src = hb.src, &diag_src = hb.diag_src, &rec_src = hb.rec_src;
  std::vector<uint16_t> &row_eptr = hb.row_eptr, &lvl = hb.lvl, &ent_slot = hb.ent_slot;
  std::vector<unsigned long long> &rec_meta = hb.rec_meta;
  bool rec_ok = hb.rec_ok;
  int max_ent = hb.max_ent, max_ext = hb.max_ext, max_lvl = hb.max_lvl;
  if (max_ext > kRecExtCap - 2 || ctx->tune.ilu_blocks == 2) rec_ok = false;      // ilu_blocks = 2: packed lists only (tests)
  const int RC = hb.rows_cap;
  const bool wide = !rec_ok && ctx->tune.ilu_blocks != 2 && hb.max_row_ent <= 16 && (size_t)RC + max_ext + 2 < 65535;
  if (wide) {
    max_ent = std::max(max_ent, RC * kWideDoubles);                 // the wide records share the LDS region of the packed entries
    max_ext += 2;                                                   // ... and the y array gets a 0.0 and a dump slot behind the faces
    B.want_wide = true;
  }
  if (rec_ok) {
    max_ent = std::max(max_ent, RC * kRecDoubles);                  // the records share the LDS region of the packed entries
    max_ext = kRecExtCap;                                           // ... and the y array has its fixed 0.0 and dump slots
  }
  B.nb = (int)nb; B.max_ent = max_ent; B.max_ext = max_ext; B.max_lvl = max_lvl;
  B.rows_cap = RC;
  B.lds = sizeof(double) * ((size_t)RC + max_ext + 1 + RC + (upper ? RC : 0) + max_ent) +
          sizeof(uint16_t) * ((size_t)((max_ent + 3) & ~3) + RC + 2 + ((max_lvl + 2 + 1) & ~1)) + sizeof(int32_t) * RC;
  if (B.lds > (size_t)150 * 1024) return KHIP_ERR_INVALID;
  bool all_fast = rec_ok;                         // every block on the row-record path: the packed entry arrays are not needed
  for (const IluBlockHdr &hh : hdr) all_fast = all_fast && hh.pad != 0;
  int rc = upload(ctx, hdr, &B.hdr);
  if (!rc) rc = upload(ctx, row_gid, &B.row_gid);
  if (!rc) rc = upload(ctx, lvl, &B.lvl);
  if (!rc) rc = upload(ctx, ext_gid, &B.ext_gid);
  if (!rc) rc = upload(ctx, dep, &B.dep);
  if (!rc && !all_fast) rc = upload(ctx, row_eptr, &B.row_eptr);
  if (!rc && !all_fast) rc = upload(ctx, ent_slot, &B.ent_slot);
  if (rc) return rc;
  if (!all_fast) {
    KHIP_CHECK_HIP(hipMalloc(&B.ent_val, sizeof(double) * std::max<size_t>(src.size(), 1)));
    if (upper) KHIP_CHECK_HIP(hipMalloc(&B.diag_val, sizeof(double) * (size_t)std::max<int64_t>(n, 1)));
  } else {
    src.clear(); diag_src.clear();
  }
  if (rec_ok && n > 0) {          // row records from the factor values (the factorisation is complete)
    int32_t *d_src4 = nullptr;
    unsigned long long *d_meta = nullptr;
    struct Scratch { int32_t *&a; unsigned long long *&b; ~Scratch() { (void)hipFree(a); (void)hipFree(b); } } scratch{d_src4, d_meta};
    KHIP_TRY(upload(ctx, rec_src, &d_src4));
    KHIP_TRY(upload(ctx, rec_meta, &d_meta));
    KHIP_CHECK_HIP(hipMalloc(&B.rec, sizeof(double) * (size_t)n * kRecDoubles));
    hipLaunchKernelGGL(ilu_pack_records_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, P->lu, d_src4, d_meta, B.row_gid, n, B.rec);
    KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  }
  KHIP_CHECK_HIP(hipMalloc(&B.done, sizeof(int) * (size_t)nb));
  KHIP_CHECK_HIP(hipMalloc(&B.ticket, sizeof(unsigned)));
  KHIP_CHECK_HIP(hipMemsetAsync(B.done, 0, sizeof(int) * (size_t)nb, ctx->stream));
  KHIP_CHECK_HIP(hipMemsetAsync(B.ticket, 0, sizeof(unsigned), ctx->stream));
  const void *fn = upper ? (const void *)ilu_block_solve_kernel<2> : (const void *)ilu_block_solve_kernel<1>;
  if (B.lds > 64 * 1024) KHIP_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)B.lds));
  int per_cu = (int)((size_t)(160 * 1024) / std::max<size_t>(B.lds, 1));
  per_cu = std::max(1, std::min(per_cu, 16));
  B.grid = (int)std::min<int64_t>(nb, (int64_t)ctx->num_cu * per_cu);
  return KHIP_OK;
}

// after the numeric factorisation: the packed factor values
int pack_block_values(khip_ilu0 *P, khip_ilu0::Blocks &B, const std::vector<int32_t> &src, const std::vector<int32_t> &diag_src, bool upper) {
  khip_ctx *ctx = P->ctx;
  int32_t *d_src = nullptr;
  struct Scratch { int32_t *&p; ~Scratch() { (void)hipFree(p); } } scratch{d_src};
  if (!src.empty() && B.ent_val) {
    KHIP_TRY(upload(ctx, src, &d_src));
    hipLaunchKernelGGL(ilu_pack_values_kernel, dim3((unsigned)((src.size() + 255) / 256)), dim3(256), 0, ctx->stream, P->lu, d_src,
                       (int64_t)src.size(), B.ent_val);
    KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    (void)hipFree(d_src); d_src = nullptr;
  }
  if (upper && !diag_src.empty() && B.diag_val) {
    KHIP_TRY(upload(ctx, diag_src, &d_src));
    hipLaunchKernelGGL(ilu_pack_values_kernel, dim3((unsigned)((diag_src.size() + 255) / 256)), dim3(256), 0, ctx->stream, P->lu, d_src,
                       (int64_t)diag_src.size(), B.diag_val);
    KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  }
  if (B.want_wide && B.ent_val && B.nb > 0) {
    KHIP_CHECK_HIP(hipMalloc(&B.recw, sizeof(double) * (size_t)std::max<int64_t>(P->n, 1) * kWideDoubles));
    hipLaunchKernelGGL(ilu_pack_wide_kernel, dim3((unsigned)B.nb), dim3(kBlkThreads), 0, ctx->stream, B.hdr, B.row_eptr, B.ent_val, B.ent_slot,
                       P->lu, P->diag, B.row_gid, B.rows_cap + B.max_ext - 2, B.recw);
    KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  }
  KHIP_CHECK_HIP(hipGetLastError());
  return KHIP_OK;
}

}  // namespace

extern "C" {

int khip_ilu0_create(khip_ctx *ctx, const khip_csr *A, khip_operator *op_out) {
  KHIP_REQUIRE(ctx && A && op_out, "ilu0_create: null argument");
  KHIP_REQUIRE(A->m == A->n || A->dist, "ilu0_create: the operator must be square");
  const int64_t n = A->m, nnz = A->nnz;
  khip_ilu0 *P = new khip_ilu0();
  struct Guard { khip_ilu0 *p; ~Guard() { if (p) ilu0_free(p); } } guard{P};      // any early return frees P
  P->ctx = ctx; P->A = A; P->n = n; P->nnz = nnz;
  P->use_graph = true;
  // ---- host analysis of the pattern (index arrays only) ---------------------------------
  std::vector<int32_t> rowptr((size_t)n + 1), col((size_t)std::max<int64_t>(nnz, 1));
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  KHIP_CHECK_HIP(hipMemcpy(rowptr.data(), A->rowptr, sizeof(int32_t) * (size_t)(n + 1), hipMemcpyDeviceToHost));
  if (nnz) KHIP_CHECK_HIP(hipMemcpy(col.data(), A->col, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToHost));
  std::vector<int32_t> row_lo((size_t)n), diag((size_t)n), row_hi((size_t)n), lev_lo((size_t)n), lev_up((size_t)n);
  int32_t nlev_lo = 0, nlev_up = 0;
  for (int64_t i = 0; i < n; ++i) {
    // owned part of the row: columns < n, contiguous and sorted (ghost columns of a distributed handle,
    // renumbered >= n, sit before / after it)
    int32_t a = rowptr[i], b = rowptr[i + 1];
    while (a < b && col[a] >= n) ++a;
    while (b > a && col[b - 1] >= n) --b;
    int32_t d = -1, lv = 0;
    for (int32_t q = a; q < b; ++q) {
      const int32_t j = col[q];
      if (j >= n || (q > a && col[q - 1] >= j)) {
        set_error("ilu0_create: row %lld: column indices must be sorted and unique", (long long)i);
        return KHIP_ERR_INVALID;
      }
      if (j == i) d = q;
      if (j < i) lv = std::max(lv, lev_lo[j] + 1);
    }
    if (d < 0) {
      set_error("ilu0_create: row %lld has no diagonal entry (structurally zero pivot)", (long long)i);
      return KHIP_ERR_NUMERIC;
    }
    row_lo[i] = a; diag[i] = d; row_hi[i] = b; lev_lo[i] = lv;
    nlev_lo = std::max(nlev_lo, lv + 1);
  }
  for (int64_t i = n - 1; i >= 0; --i) {
    int32_t lv = 0;
    for (int32_t q = diag[i] + 1; q < row_hi[i]; ++q) lv = std::max(lv, lev_up[col[q]] + 1);
    lev_up[i] = lv;
    nlev_up = std::max(nlev_up, lv + 1);
  }
  std::vector<int32_t> perm_lo, perm_up;
  sort_by_level(lev_lo, n ? nlev_lo : 0, perm_lo, P->lvl_lo);
  sort_by_level(lev_up, n ? nlev_up : 0, perm_up, P->lvl_up);
  // ---- device state ------------------------------------------------------------------------
  int rc = upload(ctx, row_lo, &P->row_lo);
  if (!rc) rc = upload(ctx, diag, &P->diag);
  if (!rc) rc = upload(ctx, row_hi, &P->row_hi);
  if (!rc) rc = upload(ctx, perm_lo, &P->perm_lo);
  if (!rc) rc = upload(ctx, perm_up, &P->perm_up);
  if (!rc) rc = upload(ctx, P->lvl_lo, &P->d_lvl_lo);
  if (!rc) rc = upload(ctx, P->lvl_up, &P->d_lvl_up);
  if (rc) return rc;
  hipError_t e = hipMalloc(&P->lu, sizeof(double) * (size_t)std::max<int64_t>(nnz, 1));
  if (e == hipSuccess) e = hipMalloc(&P->bad_row, sizeof(int));
  if (e != hipSuccess) { set_error("ilu0_create: %s", hipGetErrorString(e)); return KHIP_ERR_HIP; }
  const int none = 0x7fffffff;
  KHIP_CHECK_HIP(hipMemcpyAsync(P->bad_row, &none, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  if (nnz) KHIP_CHECK_HIP(hipMemcpyAsync(P->lu, A->val, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToDevice, ctx->stream));
  // ---- numeric factorisation, level by level (same dependency graph as the lower solve) ------
  KHIP_TRY((enqueue_levels<0>(P, P->lvl_lo, P->d_lvl_lo, P->perm_lo, nullptr, nullptr)));
  int bad = none;
  e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(&bad, P->bad_row, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) { set_error("ilu0_create: %s", hipGetErrorString(e)); return KHIP_ERR_HIP; }
  if (bad != none) {
    set_error("ilu0_create: zero pivot in row %d", bad);
    return KHIP_ERR_NUMERIC;
  }
  // ---- block schedule of the solves where the pattern is a structured grid (else: level scheduling) ----
  if (ctx->tune.ilu_blocks != 0) {
    const HostPattern H{n, col, row_lo, diag, row_hi};
    const bool grid = ctx->tune.ilu_blocks != 3 && detect_grid(H, P->grid_dims, P->grid_skew);      // 3: level-sequence blocks even on a grid (for comparison)
    // without a grid: blocks from the level-sorted row sequence, where the levels are wide enough to be worth a flag each
    const int64_t nlev = (int64_t)P->lvl_lo.size() - 1 + (int64_t)P->lvl_up.size() - 1;
    // the analysis holds ~80 bytes per row and triangle in host memory (two triangles side by side): not on a host that is short
    const long avail_pages = sysconf(_SC_AVPHYS_PAGES), page = sysconf(_SC_PAGESIZE);
    const bool host_ok = avail_pages <= 0 || page <= 0 || (double)avail_pages * (double)page > 200.0 * (double)n;
    if (host_ok && (grid || (n >= 4096 && nlev > 0 && 2 * n / nlev >= 32))) {
      KHIP_CHECK_HIP(hipMalloc(&P->blk_fail, sizeof(int)));
      KHIP_CHECK_HIP(hipMemsetAsync(P->blk_fail, 0, sizeof(int), ctx->stream));
      if (hipHostMalloc(reinterpret_cast<void **>(&P->blk_fail_host), sizeof(int), hipHostMallocMapped) == hipSuccess) {
        *P->blk_fail_host = 0;
        if (hipHostGetDevicePointer(reinterpret_cast<void **>(&P->blk_fail_host_dev), P->blk_fail_host, 0) != hipSuccess) P->blk_fail_host_dev = nullptr;
      } else {
        (void)hipGetLastError();
        P->blk_fail_host = nullptr;
      }
      HostBlocks hlo, hup;                     // the two triangles are analysed side by side (pure host work)
      auto analyse = [&](bool upper, HostBlocks &hb) {           // host memory may run out on a very large slab: then level scheduling
        try {
          Partition part;
          hb.rc = grid ? make_grid_partition(H, P->grid_dims, P->grid_skew, upper, part)
                       : make_level_partition(n, upper ? perm_up : perm_lo, upper ? P->lvl_up : P->lvl_lo, part);
          if (hb.rc == KHIP_OK) hb.rc = analyse_blocks(H, part, upper, hb);
        } catch (const std::exception &) { hb.rc = KHIP_ERR_INVALID; }
      };
      std::thread tup(analyse, true, std::ref(hup));
      analyse(false, hlo);
      tup.join();
      int rb = hlo.rc != KHIP_OK ? hlo.rc : hup.rc;
      if (rb == KHIP_OK) rb = upload_blocks(P, false, hlo, P->blk_lo);
      if (rb == KHIP_OK) rb = pack_block_values(P, P->blk_lo, hlo.src, hlo.diag_src, false);
      if (rb == KHIP_OK) rb = upload_blocks(P, true, hup, P->blk_up);
      if (rb == KHIP_OK) rb = pack_block_values(P, P->blk_up, hup.src, hup.diag_src, true);
      if (rb == KHIP_OK) P->use_blocks = true;
      else {                                   // not representable (or out of memory): keep the level schedule
        (void)hipGetLastError();
        blocks_free(P->blk_lo);
        blocks_free(P->blk_up);
        P->grid_dims[0] = P->grid_dims[1] = P->grid_dims[2] = 0;
      }
    }
  }
  guard.p = nullptr;
  op_out->csr = nullptr;
  op_out->apply = ilu0_apply;
  op_out->self = P;
  return KHIP_OK;
}

int khip_ilu0_destroy(khip_operator *op) {
  if (!op || op->apply != ilu0_apply || !op->self) return KHIP_OK;
  ilu0_free(static_cast<khip_ilu0 *>(op->self));
  op->self = nullptr;
  return KHIP_OK;
}

int khip_ilu0_info(const khip_operator *op, int64_t *levels_lower, int64_t *levels_upper, const double **lu_dev) {
  KHIP_REQUIRE(op && op->apply == ilu0_apply && op->self, "ilu0_info: not an ILU(0) operator");
  const khip_ilu0 *P = static_cast<const khip_ilu0 *>(op->self);
  if (levels_lower) *levels_lower = (int64_t)P->lvl_lo.size() - 1;
  if (levels_upper) *levels_upper = (int64_t)P->lvl_up.size() - 1;
  if (lu_dev) *lu_dev = P->lu;
  return KHIP_OK;
}

int khip_ilu0_block_info(const khip_operator *op, int64_t *dims3, int64_t *blocks, int *failed) {
  KHIP_REQUIRE(op && op->apply == ilu0_apply && op->self, "ilu0_block_info: not an ILU(0) operator");
  const khip_ilu0 *P = static_cast<const khip_ilu0 *>(op->self);
  if (dims3) for (int k = 0; k < 3; ++k) dims3[k] = P->use_blocks ? P->grid_dims[k] : 0;
  if (blocks) *blocks = P->use_blocks ? P->blk_lo.nb : 0;
  if (failed) {
    *failed = 0;
    if (P->blk_fail) {
      KHIP_CHECK_HIP(hipStreamSynchronize(P->ctx->stream));
      int epoch_failed = 0;
      KHIP_CHECK_HIP(hipMemcpy(&epoch_failed, P->blk_fail, sizeof(int), hipMemcpyDeviceToHost));
      *failed = (epoch_failed != 0 || (P->blk_fail_host && *P->blk_fail_host != 0)) ? 1 : 0;
    }
  }
  return KHIP_OK;
}

#ifdef KHIP_ILU_TRACE
int khip_debug_ilu_trace(unsigned long long *out512) {
  return hipMemcpyFromSymbol(out512, HIP_SYMBOL(khip::g_ilu_trace), sizeof(unsigned long long) * 512) == hipSuccess ? 0 : -1;
}
#endif

// Host-only check of the block schedule's analysis (no device needed: tests/test_ilu_blocks_host.py): the pattern as CSR
// with sorted columns and a diagonal in every row; mode 1 = as khip_ilu0_create would (grid if recognised, else the level
// sequence), 3 = level sequence.  out: [0..2] grid dims (0 if none), [3] 1 = skewed basis, [4] blocks lower, [5] blocks upper,
// [6] largest face list, [7] 1 = 48-byte records possible (lower), [8] largest row (entries in a triangle), [9] rows_cap.
// Returns KHIP_ERR_INVALID if a block would depend on a later one (the analysis checks every dependency).
int khip_test_ilu_blocks_host(int64_t n, const int64_t *rowptr, const int32_t *colidx, int mode, int64_t *out10) {
  KHIP_REQUIRE(n > 0 && rowptr && colidx && out10, "test_ilu_blocks_host: bad arguments");
  const int64_t nnz = rowptr[n];
  std::vector<int32_t> col(colidx, colidx + nnz), row_lo((size_t)n), diag((size_t)n), row_hi((size_t)n), lev_lo((size_t)n), lev_up((size_t)n);
  int32_t nlev_lo = 0, nlev_up = 0;
  for (int64_t i = 0; i < n; ++i) {
    int32_t d = -1, lv = 0;
    for (int64_t q = rowptr[i]; q < rowptr[i + 1]; ++q) {
      const int32_t j = col[(size_t)q];
      if (j < 0 || j >= n || (q > rowptr[i] && col[(size_t)q - 1] >= j)) { set_error("test_ilu_blocks_host: row %lld not sorted", (long long)i); return KHIP_ERR_INVALID; }
      if (j == i) d = (int32_t)q;
      if (j < i) lv = std::max(lv, lev_lo[(size_t)j] + 1);
    }
    if (d < 0) { set_error("test_ilu_blocks_host: row %lld has no diagonal", (long long)i); return KHIP_ERR_INVALID; }
    row_lo[(size_t)i] = (int32_t)rowptr[i]; diag[(size_t)i] = d; row_hi[(size_t)i] = (int32_t)rowptr[i + 1]; lev_lo[(size_t)i] = lv;
    nlev_lo = std::max(nlev_lo, lv + 1);
  }
  for (int64_t i = n - 1; i >= 0; --i) {
    int32_t lv = 0;
    for (int32_t q = diag[(size_t)i] + 1; q < row_hi[(size_t)i]; ++q) lv = std::max(lv, lev_up[(size_t)col[(size_t)q]] + 1);
    lev_up[(size_t)i] = lv;
    nlev_up = std::max(nlev_up, lv + 1);
  }
  std::vector<int32_t> perm_lo, perm_up;
  std::vector<int64_t> lvl_lo, lvl_up;
  sort_by_level(lev_lo, nlev_lo, perm_lo, lvl_lo);
  sort_by_level(lev_up, nlev_up, perm_up, lvl_up);
  const HostPattern H{n, col, row_lo, diag, row_hi};
  int64_t dims[3] = {0, 0, 0};
  int skew[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  const bool grid = mode != 3 && detect_grid(H, dims, skew);
  for (int k = 0; k < 3; ++k) out10[k] = grid ? dims[k] : 0;
  out10[3] = grid && skew[1] != 0;
  HostBlocks hb[2];
  for (int up = 0; up < 2; ++up) {
    Partition part;
    int rc = grid ? make_grid_partition(H, dims, skew, up != 0, part) : make_level_partition(n, up ? perm_up : perm_lo, up ? lvl_up : lvl_lo, part);
    if (rc == KHIP_OK) rc = analyse_blocks(H, part, up != 0, hb[up]);
    if (rc != KHIP_OK) return rc;
    // every row in exactly one block
    std::vector<char> seen((size_t)n, 0);
    for (int32_t g : hb[up].row_gid) { if (g < 0 || g >= n || seen[(size_t)g]) { set_error("test_ilu_blocks_host: row %d twice or out of range", g); return KHIP_ERR_INVALID; } seen[(size_t)g] = 1; }
    if ((int64_t)hb[up].row_gid.size() != n) { set_error("test_ilu_blocks_host: %lld of %lld rows in blocks", (long long)hb[up].row_gid.size(), (long long)n); return KHIP_ERR_INVALID; }
  }
  out10[4] = hb[0].nb; out10[5] = hb[1].nb; out10[6] = std::max(hb[0].max_ext, hb[1].max_ext); out10[7] = hb[0].rec_ok;
  out10[8] = std::max(hb[0].max_row_ent, hb[1].max_row_ent); out10[9] = hb[0].rows_cap;
  return KHIP_OK;
}

int khip_ilu0_set_graph(khip_operator *op, int enable) {
  KHIP_REQUIRE(op && op->apply == ilu0_apply && op->self, "ilu0_set_graph: not an ILU(0) operator");
  static_cast<khip_ilu0 *>(op->self)->use_graph = enable != 0;
  return KHIP_OK;
}

}  // extern "C"
