// spmv_common.hpp -- argument block and load helpers shared by the CSR kernels.
#pragma once

#include "device_reduce.hpp"

namespace khip {

typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef int int2v __attribute__((ext_vector_type(2)));
typedef int int4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kBufRsrcWord3 = 0x00020000;   // gfx9 raw buffer descriptor: 32-bit data format, range checking on

constexpr int kPad = 8;   // val/col are over-allocated by this many zeroed entries

struct SpmvArgs {
  const int32_t *rowptr;
  const int32_t *blockptr;   // rowptr[256 * i] (i = 0..ceil(m/256)), or null
  const int32_t *col;
  const double *val;
  const double *x;
  const double *ghost;   // remote x entries (distributed), indexed col - n_owned
  double *y;
  int64_t n_owned;       // columns < n_owned read x, others read ghost
  int64_t row_lo, row_hi;
  int64_t hole_lo, hole_len;   // staged / coded kernels: the launch covers [row_lo, hole_lo) and [hole_lo + hole_len, row_hi) -- the two boundary ranges of a row-partitioned product in ONE launch (hole_lo - row_lo is a multiple of the row block; hole_len = 0: none)
  int xcd_remap;         // see chunk_id() in spmv.hip
  int sweep_s, sweep_w;  // plane sweep (xcd_remap == -2): tiles per plane, tiles per XCD column
  int nt_y;              // non-temporal store of y
  int tiles_per_block;   // staged kernel: consecutive row blocks per workgroup (software pipeline depth)
  int64_t nnz_bound;     // nnz + pad: prefetches beyond it are clamped
  int stage_cap;         // staged kernel: LDS window in entries (multiple of 4, <= 2048)
  const uint16_t *tmpl_id;   // row-template compressed handle (template kernel only)
  const int32_t *tmpl_off;
  const double *tmpl_val;
  const int32_t *tmpl_cnt;
  int tmpl_T, tmpl_K;
  const double *dotw;    // left vector of the fused dot: results[slot] = dotw . y   (x for p.Ap; another vector for c.(A p))
  int dot_sq;            // staged kernel: also results[slot + 1] = y . y
  int stream_nt;         // delta kernel: non-temporal policy on the val / column window loads
  int blk_pub;           // fused dots of the staged / coded / delta kernels: one double-double tree per workgroup (block_publish) instead of one per wave
  int dot_early;         // staged kernels: load dotw[row] before the row block's windows instead of after the row walk
  const long long *stop_seq;   // device-resident loop control (solver_device.hpp); null outside such loops
  long long seq;
  int fake_gather;       // experiment: coalesced x reads instead of x[col] (WRONG results)
  // dictionary-coded column indices (colcode.hip): col = row + code_tab[code[k]]
  const void *code;          // uint8_t[nnz + pad] or uint16_t[nnz + pad]
  const int32_t *code_tab;   // sorted distinct (column - row) offsets, code_T entries
  int code_T;
  // sliced form of the coded operator (csr_build_sell): 64-row slices, per slice W code words then L value words per lane
  const unsigned long long *sell;
  const uint32_t *sell_off;  // first unit (64 words) of every slice; null: sell_units units per slice
  int sell_units;
  int sell_cols;             // 0: code words (eight 1-byte codes), 1: column words (two int32 columns)
  int sell_pair;             // the words of a row interleaved in 16-byte pairs (rows of at most 8 entries, one code word)
  const uint32_t *sell_c4;   // narrow codes: eight 4-bit codes per row in one word indexed by the row (the slices hold values only); null otherwise
  // block-delta column stream (coldelta.hip): col = dbase[block] + dcode[k]; all-ones code = escape
  const void *dcode;         // uint8_t[nnz + pad] or uint16_t[nnz + pad]
  const int32_t *dbase;      // base column of every row block
  const int32_t *desc_ptr;   // [blocks + 1] first escape of every row block
  const uint16_t *desc_pos;  // escape: entry position inside its block
  const int32_t *desc_col;   // escape: column
  int stage_rows;            // coded / pipelined / delta kernels: rows per block (256, 128, 64 or 32)
  int max_row;               // pipelined kernel: longest row of the operator (uniform trip count of the row walk)
};

template <bool NT, typename T>
__device__ __forceinline__ T ld(const T *p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}

template <bool DIST>
__device__ __forceinline__ double gather_x(const SpmvArgs &a, int32_t c) {
  if (DIST) {
    const double *src = (c < a.n_owned) ? a.x : (a.ghost - a.n_owned);
    return src[c];
  }
  return a.x[c];
}

}  // namespace khip
