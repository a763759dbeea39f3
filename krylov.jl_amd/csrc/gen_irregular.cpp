// gen_irregular.cpp -- generator of the "banded + random, fixed seed" benchmark operator (the stand-in for
// the SuiteSparse matrices of the reference's benchmarks, benchmark/cg_bmark.jl:29-54, which are not available offline).
// Built on the host (a hash per candidate entry, a few seconds for 10 M rows), uploaded like khip_gen_stencil's output.
//
// Definition (the CPU checker under oracle/ restates it independently as ko_csr_banded_random; tests compare the arrays):
//   mix(z)        splitmix64 finaliser
//   key(i, j)     mix(seed ^ (i * 0x9E3779B97F4A7C15 + j))                                   (i < j)
//   band          (i, i + d), 1 <= d <= hb, present iff i + d < n and key(i, i + d) & 7 != 0   (7 of 8 kept; symmetric pattern)
//   links         k = 0 .. K-1: rows are paired inside blocks of 2^B rows (B = min(20, floor(log2 n)); rows past the last
//                 full block have none) by the involution  pi_k(x) = s ^ inv_k(fwd_k(x ^ s) ^ 1),  x = row mod 2^B,
//                 s = mix(seed + 0x51 + 131 k + 977 block) mod 2^B, fwd_k a two-round multiply / xor-shift bijection of B bits;
//                 the pair is dropped when |i - j| <= hb or when an earlier link of the row has the same partner
//   off-diagonal  -(1 + (key(i, j) >> 8 & 255) / 256)  for both (i, j) and (j, i); flag UNSYM halves the entries above the diagonal
//   dense rows    (UNSYM only) rows R_q = (q + 1) n / (ndense + 1), q < ndense, get 3000 further entries at the columns
//                 (R_q + 1 + t stride) mod n, stride = (n / 3001) | 1, value -(1 + (key >> 8 & 255) / 256) / 64, where not present already
//   diagonal      1/16 + sum of |off-diagonal entries of the row|   (dyadic rationals: exact in any order; eigenvalues >= 1/16)
//   columns ascending within a row.
// More than 2048 distinct diagonals (the links), rows of 20 .. 31 entries for hb = 13, K = 3: nothing stencil-specific applies.
#include <algorithm>
#include <thread>
#include <vector>

#include "khip_internal.hpp"

namespace {

constexpr int kPad = 8;   // val / col are over-allocated by this many zeroed entries (spmv_common.hpp)

inline uint64_t mix(uint64_t z) {
  z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ull;
  z ^= z >> 27; z *= 0x94d049bb133111ebull;
  z ^= z >> 31;
  return z;
}

struct IrregularSpec {
  int64_t n;
  int hb, K, B;
  uint64_t seed, mask;
  bool unsym;
  int ndense;
  int h;                                  // xor-shift distance: 2 h >= B makes x ^= x >> h its own inverse
  std::vector<uint64_t> a1, a2, a1i, a2i; // per link: odd multipliers and their inverses mod 2^B
  int64_t dense_stride;

  uint64_t key(int64_t i, int64_t j) const { return mix(seed ^ ((uint64_t)i * 0x9E3779B97F4A7C15ull + (uint64_t)j)); }
  bool band(int64_t i, int64_t j) const { return (key(i, j) & 7) != 0; }              // i < j
  double mag(int64_t i, int64_t j) const { return 1.0 + (double)((key(i, j) >> 8) & 255) / 256.0; }   // i < j
  uint64_t fwd(int k, uint64_t x) const {
    x = (x * a1[(size_t)k]) & mask; x ^= x >> h;
    x = (x * a2[(size_t)k]) & mask; x ^= x >> h;
    return x;
  }
  uint64_t inv(int k, uint64_t y) const {
    y ^= y >> h; y = (y * a2i[(size_t)k]) & mask;
    y ^= y >> h; y = (y * a1i[(size_t)k]) & mask;
    return y;
  }
  int64_t partner(int k, int64_t r) const {
    const int64_t blk = r >> B;
    if (blk >= (n >> B)) return -1;
    const uint64_t s = mix(seed + 0x51 + 131ull * (uint64_t)k + 977ull * (uint64_t)blk) & mask;
    const uint64_t x = (uint64_t)r & mask;
    return (blk << B) | (int64_t)(s ^ inv(k, fwd(k, x ^ s) ^ 1));
  }
  int dense_index(int64_t r) const {
    if (!unsym) return -1;
    for (int q = 0; q < ndense; ++q) if (r == (int64_t)(q + 1) * n / (ndense + 1)) return q;
    return -1;
  }
};

uint64_t inv_mod_pow2(uint64_t a, uint64_t mask) {          // a odd
  uint64_t x = a;                                           // correct to 3 bits; each step doubles them
  for (int it = 0; it < 6; ++it) x = x * (2 - a * x);
  return x & mask;
}

// the row's entries in ascending column order; returns the count (cols / vals may be null: count only)
int64_t build_row(const IrregularSpec &S, int64_t r, std::vector<int32_t> &cols, std::vector<double> &vals, bool fill) {
  struct Ent { int64_t c; double v; };
  Ent small[96];
  std::vector<Ent> big;
  int cnt = 0;
  auto push = [&](int64_t c, double v) {
    if (!big.empty() || cnt == 96) { if (big.empty()) big.assign(small, small + cnt); big.push_back({c, v}); ++cnt; return; }
    small[cnt++] = {c, v};
  };
  for (int d = S.hb; d >= 1; --d) {
    const int64_t c = r - d;
    if (c >= 0 && S.band(c, r)) push(c, -S.mag(c, r));
  }
  for (int d = 1; d <= S.hb; ++d) {
    const int64_t c = r + d;
    if (c < S.n && S.band(r, c)) push(c, -S.mag(r, c) * (S.unsym ? 0.5 : 1.0));
  }
  int64_t seen[16];
  int nseen = 0;
  for (int k = 0; k < S.K; ++k) {
    const int64_t c = S.partner(k, r);
    if (c < 0 || c >= S.n) continue;
    const int64_t dist = c > r ? c - r : r - c;
    if (dist <= S.hb) continue;
    bool dup = false;
    for (int q = 0; q < nseen; ++q) dup |= seen[q] == c;
    if (dup) continue;
    seen[nseen++] = c;
    const int64_t lo = r < c ? r : c, hi = r < c ? c : r;
    push(c, -S.mag(lo, hi) * ((S.unsym && c > r) ? 0.5 : 1.0));
  }
  const int dq = S.dense_index(r);
  if (dq >= 0) {
    std::vector<int64_t> have;
    const Ent *e0 = big.empty() ? small : big.data();
    for (int q = 0; q < cnt; ++q) have.push_back(e0[q].c);
    std::sort(have.begin(), have.end());
    std::vector<int64_t> extra;
    for (int t = 0; t < 3000; ++t) {
      const int64_t c = (r + 1 + (int64_t)t * S.dense_stride) % S.n;
      if (c == r || std::binary_search(have.begin(), have.end(), c)) continue;
      extra.push_back(c);
    }
    std::sort(extra.begin(), extra.end());
    extra.erase(std::unique(extra.begin(), extra.end()), extra.end());
    for (int64_t c : extra) {
      const int64_t lo = r < c ? r : c, hi = r < c ? c : r;
      push(c, -S.mag(lo, hi) / 64.0);
    }
  }
  Ent *e = big.empty() ? small : big.data();
  double diag = 0.0625;
  for (int q = 0; q < cnt; ++q) diag += -e[q].v;
  if (!fill) return cnt + 1;
  std::sort(e, e + cnt, [](const Ent &a, const Ent &b) { return a.c < b.c; });
  cols.clear(); vals.clear();
  bool placed = false;
  for (int q = 0; q < cnt; ++q) {
    if (!placed && e[q].c > r) { cols.push_back((int32_t)r); vals.push_back(diag); placed = true; }
    cols.push_back((int32_t)e[q].c); vals.push_back(e[q].v);
  }
  if (!placed) { cols.push_back((int32_t)r); vals.push_back(diag); }
  return cnt + 1;
}

template <class F>
void parallel_rows(int64_t m, F f) {
  unsigned nt = std::thread::hardware_concurrency();
  if (nt < 1) nt = 1;
  if (nt > 32) nt = 32;
  if (m < 4096) nt = 1;
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; ++t) th.emplace_back([=] { f(m * t / nt, m * (t + 1) / nt); });
  for (auto &x : th) x.join();
}

}  // namespace

using namespace khip;

// the rows [row0, row0 + m) of the operator in host memory (col / val padded by kPad zeros, as every CSR handle's arrays are)
static int build_banded_random_host(int64_t n, int half_band, int links, uint64_t seed, int flags, int dense_rows, int64_t row0, int64_t m,
                                    std::vector<int32_t> &rp32, std::vector<int32_t> &col, std::vector<double> &val, int64_t *nnz_out) {
  KHIP_REQUIRE(n >= 2 && n < (1ll << 31) && half_band >= 0 && half_band <= 64 && links >= 0 && links <= 16 && dense_rows >= 0 && dense_rows <= 64,
               "gen_banded_random: n in [2, 2^31), half_band <= 64, links <= 16, dense_rows <= 64 required");
  KHIP_REQUIRE(row0 >= 0 && m >= 0 && row0 + m <= n, "gen_banded_random: bad row range");
  IrregularSpec S;
  S.n = n; S.hb = half_band; S.K = links; S.seed = seed; S.unsym = (flags & 1) != 0; S.ndense = dense_rows;
  int B = 0;
  while ((2ll << B) <= n) ++B;
  S.B = B < 20 ? B : 20;
  S.mask = (1ull << S.B) - 1;
  S.h = (S.B + 1) / 2;
  if (S.h < 1) S.h = 1;
  for (int k = 0; k < links; ++k) {
    const uint64_t a1 = (mix(seed * 4 + 4ull * (uint64_t)k + 1) | 1) & S.mask, a2 = (mix(seed * 4 + 4ull * (uint64_t)k + 2) | 1) & S.mask;
    S.a1.push_back(a1); S.a2.push_back(a2);
    S.a1i.push_back(inv_mod_pow2(a1, S.mask)); S.a2i.push_back(inv_mod_pow2(a2, S.mask));
  }
  S.dense_stride = (n / 3001) | 1;
  std::vector<int64_t> rp((size_t)m + 1, 0);
  parallel_rows(m, [&](int64_t lo, int64_t hi) {
    std::vector<int32_t> c; std::vector<double> v;
    for (int64_t i = lo; i < hi; ++i) rp[(size_t)i + 1] = build_row(S, row0 + i, c, v, false);
  });
  for (int64_t i = 0; i < m; ++i) rp[(size_t)i + 1] += rp[(size_t)i];
  const int64_t total = rp[(size_t)m];
  if (total >= (1ll << 31) - 64) { set_error("gen_banded_random: shard nnz %lld does not fit int32 row pointers", (long long)total); return KHIP_ERR_INVALID; }
  rp32.assign((size_t)m + 1, 0);
  col.assign((size_t)total + kPad, 0);
  val.assign((size_t)total + kPad, 0.0);
  for (int64_t i = 0; i <= m; ++i) rp32[(size_t)i] = (int32_t)rp[(size_t)i];
  parallel_rows(m, [&](int64_t lo, int64_t hi) {
    std::vector<int32_t> c; std::vector<double> v;
    for (int64_t i = lo; i < hi; ++i) {
      build_row(S, row0 + i, c, v, true);
      std::copy(c.begin(), c.end(), col.begin() + rp[(size_t)i]);
      std::copy(v.begin(), v.end(), val.begin() + rp[(size_t)i]);
    }
  });
  *nnz_out = total;
  return KHIP_OK;
}

extern "C" int khip_gen_banded_random(khip_ctx *ctx, int64_t n, int half_band, int links, uint64_t seed, int flags, int dense_rows,
                                      int64_t row0, int64_t m, int32_t **rowptr_dev, int32_t **col_dev, double **val_dev,
                                      int64_t *nnz_out) {
  KHIP_REQUIRE(ctx && rowptr_dev && col_dev && val_dev && nnz_out, "gen_banded_random: null argument");
  std::vector<int32_t> rp32, col;
  std::vector<double> val;
  int64_t total = 0;
  KHIP_TRY(build_banded_random_host(n, half_band, links, seed, flags, dense_rows, row0, m, rp32, col, val, &total));
  KHIP_CHECK_HIP(hipSetDevice(ctx->device));
  int32_t *d_rp = nullptr, *d_cl = nullptr;
  double *d_vl = nullptr;
  KHIP_CHECK_HIP(hipMalloc(&d_rp, sizeof(int32_t) * (size_t)(m + 1)));
  KHIP_CHECK_HIP(hipMalloc(&d_cl, sizeof(int32_t) * (size_t)(total + kPad)));
  KHIP_CHECK_HIP(hipMalloc(&d_vl, sizeof(double) * (size_t)(total + kPad)));
  KHIP_CHECK_HIP(hipMemcpy(d_rp, rp32.data(), sizeof(int32_t) * (size_t)(m + 1), hipMemcpyHostToDevice));
  KHIP_CHECK_HIP(hipMemcpy(d_cl, col.data(), sizeof(int32_t) * (size_t)(total + kPad), hipMemcpyHostToDevice));
  KHIP_CHECK_HIP(hipMemcpy(d_vl, val.data(), sizeof(double) * (size_t)(total + kPad), hipMemcpyHostToDevice));
  *rowptr_dev = d_rp; *col_dev = d_cl; *val_dev = d_vl; *nnz_out = total;
  return KHIP_OK;
}

// test-only, host-only: the same rows into caller-provided host arrays (no device).  Call with col_out = val_out = null to get
// the row pointers (m + 1 entries) and the entry count, then with arrays of that size.
extern "C" int khip_test_gen_banded_random_host(int64_t n, int half_band, int links, uint64_t seed, int flags, int dense_rows, int64_t row0,
                                                int64_t m, int32_t *rowptr_out, int32_t *col_out, double *val_out, int64_t *nnz_out) {
  KHIP_REQUIRE(rowptr_out && nnz_out, "test_gen_banded_random_host: null argument");
  std::vector<int32_t> rp32, col;
  std::vector<double> val;
  int64_t total = 0;
  KHIP_TRY(build_banded_random_host(n, half_band, links, seed, flags, dense_rows, row0, m, rp32, col, val, &total));
  std::copy(rp32.begin(), rp32.end(), rowptr_out);
  if (col_out) std::copy(col.begin(), col.begin() + total, col_out);
  if (val_out) std::copy(val.begin(), val.begin() + total, val_out);
  *nnz_out = total;
  return KHIP_OK;
}
