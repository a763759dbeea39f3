// streamfloor.hip -- what the memory system of this chip gives the ACCESS MIXES of the block-GMRES kernels, free of their
// arithmetic and of their structure.  Two synthetic twins:
//
//   panel   the Gram-Schmidt step of csrc/panel.hip (panel_nn_tn_kernel: read V_i, read Q, write Q, read V_{i+1} = 3R + 1W over
//           n_pad x 16 doubles each) and the two-read form (panel_gemm_tn / axpby: 2R [+ 1W]) as plain streams: NR read streams and
//           one in-place written stream, 8 or 16 bytes per lane and load, U loads in flight per stream and lane, flat tiles.
//           The panel kernels load 8 B per lane (one MFMA operand element); BLAS-1 loads 16 B per lane.
//   spmm    the tile SpMM of csrc/spmm_tile.hip at cfg 5 (27-point 216^3, p = 16): per group of 32 rows (a 4 x 4 x 2 grid tile)
//           9 B per entry (8 B value + 1 B slot) + the group record, the 6 x 6 x 4 = 144 distinct panel rows of 128 B the tile
//           references (same reuse between neighbouring tiles, same pencil order of the groups, same XCD split), 32 rows of Y
//           written.  No LDS, no dependence between loads: every load of a group is issued before anything waits.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/streamfloor tools/streamfloor.hip ;  tools/streamfloor panel | spmm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
typedef double dbl2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <typename F> static float timeit(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int r = 0; r < reps; ++r) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

// ------------------------------------------------------------------------------------------------ panel streams
template <typename T> struct Acc;
template <> struct Acc<double> { static __device__ double mad(double a, double b, double c) { return fma(a, b, c); } };
template <> struct Acc<dbl2> { static __device__ dbl2 mad(dbl2 a, dbl2 b, dbl2 c) { dbl2 o; o.x = fma(a.x, b.x, c.x); o.y = fma(a.y, b.y, c.y); return o; } };

// q <- q + 0.5 v [+ 0.25 u]; NR = 2: reads v, q; NR = 3: reads v, q, u; NR = 1: q <- 1.5 q IN PLACE (one read + one write of the same panel: the
// scalings Q <- Q R^-1 of the panel QR).  WRITE = false: reads only (the TN product's mix).
// NTS = 2: non-temporal LOADS as well as stores (what the BLAS-1 kernels of csrc/blas1.hip do for long vectors)
template <typename T, int NR, int U, bool WRITE, int NTS>
__global__ __launch_bounds__(256) void k_panel(const T *v, T *q, const T *u, long nv, double *sink) {
  const long base = (long)blockIdx.x * (256 * U) + threadIdx.x;
  T a[U], b[U], c[U];
#pragma unroll
  for (int j = 0; j < U; ++j) { const long i = base + j * 256; if (i < nv) {
      if (NTS == 2) { b[j] = __builtin_nontemporal_load(q + i); a[j] = NR == 1 ? b[j] : __builtin_nontemporal_load(v + i); if (NR == 3) c[j] = __builtin_nontemporal_load(u + i); }
      else { b[j] = q[i]; a[j] = NR == 1 ? b[j] : v[i]; if (NR == 3) c[j] = u[i]; } } }
  T half, quarter, acc;
  memset(&acc, 0, sizeof(T));
  if constexpr (sizeof(T) == 8) { half = 0.5; quarter = 0.25; } else { half = dbl2{0.5, 0.5}; quarter = dbl2{0.25, 0.25}; }
#pragma unroll
  for (int j = 0; j < U; ++j) { const long i = base + j * 256; if (i < nv) {
      T o = Acc<T>::mad(half, a[j], b[j]);
      if (NR == 3) o = Acc<T>::mad(quarter, c[j], o);
      if (WRITE) { if (NTS != 0) __builtin_nontemporal_store(o, q + i); else q[i] = o; } else acc = Acc<T>::mad(o, o, acc);
  } }
  if (!WRITE) {
    double s; if constexpr (sizeof(T) == 8) s = acc; else s = acc.x + acc.y;
    if (s == 12345.678) *sink = s;
  }
}

template <typename T, int NR, int U, bool WRITE, int NTS>
static void run_panel(const char *name, double *v, double *q, double *u, long n, double *sink) {
  const long nv = n * 8 / (long)sizeof(T);
  const long G = (nv + 256L * U - 1) / (256L * U);
  const float ms = timeit([&] { hipLaunchKernelGGL((k_panel<T, NR, U, WRITE, NTS>), dim3((unsigned)G), dim3(256), 0, 0, (const T *)v, (T *)q, (const T *)u, nv, sink); }, 20);
  const double bytes = (double)(NR + (WRITE ? 1 : 0)) * 8.0 * (double)n;
  printf("%-10s %2d B/lane  U=%d  nts=%d  %.3f ms  %7.0f GB/s  %.3f of 8 TB/s\n", name, (int)sizeof(T), U, (int)NTS, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0);
  fflush(stdout);
}

// The fused Gram-Schmidt kernel's OWN access pattern without its arithmetic (csrc/panel.hip panel_nn_tn_kernel, p = 16): a wave walks RPW
// consecutive rows in steps of UNT 16-row tiles; per tile two 16-byte loads of the V_i tile as it lies in memory, four 8-byte loads of
// the Q tile and four of the V_{i+1} tile in operand order (row r0 + k + 4g, column i), four stores of the Q tile.  Says whether the
// kernel's distance to the contiguous streams above is its access pattern (this twin is slow too) or its MFMA / LDS chain (this twin is fast).
// BS = threads per workgroup (64 / 128 / 256); XCD = 1: workgroup j takes chunk (j mod 8) * (G / 8) + j / 8, i.e. each XCD walks its own eighth
template <int RPW, int UNT, bool NTL, int BS = 256, int XCD = 0>
__global__ __launch_bounds__(BS) void k_tilewalk(const double *Vi, double *Q, const double *Vn, long n_pad) {
  const int lane = threadIdx.x & 63, i = lane & 15, k = lane >> 4;
  long blk = blockIdx.x;
  if (XCD) { const long G8 = gridDim.x / 8; if (blk < G8 * 8) blk = (blk & 7) * G8 + (blk >> 3); }
  const long wid = blk * (BS / 64) + (threadIdx.x >> 6);
  const long row_begin = wid * RPW;
  if (row_begin >= n_pad) return;
  const long row_end = row_begin + RPW < n_pad ? row_begin + RPW : n_pad;
  for (long r = row_begin; r < row_end; r += 16 * UNT) {
    dbl2 a[UNT][2]; double c[UNT][4], v[UNT][4];
#pragma unroll
    for (int t = 0; t < UNT; ++t) {
      const long r0 = r + 16 * t;
      if (r0 < row_end) {
#pragma unroll
        for (int h = 0; h < 2; ++h) { const int c16 = lane + 64 * h, row = c16 >> 3, cc = c16 & 7;
          const dbl2 *pa = reinterpret_cast<const dbl2 *>(Vi + (r0 + row) * 16 + 2 * cc); a[t][h] = NTL ? __builtin_nontemporal_load(pa) : *pa; }
#pragma unroll
        for (int g = 0; g < 4; ++g) { const double *pq = Q + (r0 + k + 4 * g) * 16 + i; c[t][g] = NTL ? __builtin_nontemporal_load(pq) : *pq; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const double *pv = Vn + (r0 + 4 * u + k) * 16 + i; v[t][u] = NTL ? __builtin_nontemporal_load(pv) : *pv; }
      }
    }
#pragma unroll
    for (int t = 0; t < UNT; ++t) {
      const long r0 = r + 16 * t;
      if (r0 < row_end) {
        const double s = a[t][0].x + a[t][0].y + a[t][1].x + a[t][1].y;
#pragma unroll
        for (int g = 0; g < 4; ++g) { const double o = fma(0.5, c[t][g], fma(0.25, v[t][g], s)); double *pq = Q + (r0 + k + 4 * g) * 16 + i;
          if (NTL) __builtin_nontemporal_store(o, pq); else *pq = o; }
      }
    }
  }
}
template <int RPW, int UNT, bool NTL, int BS = 256, int XCD = 0>
static void run_tilewalk(double *v, double *q, double *u, long n) {
  const long n_pad = n / 16;
  const long per = (long)(BS / 64) * RPW;
  const long G = (n_pad + per - 1) / per;
  const float ms = timeit([&] { hipLaunchKernelGGL((k_tilewalk<RPW, UNT, NTL, BS, XCD>), dim3((unsigned)G), dim3(BS), 0, 0, v, q, u, n_pad); }, 20);
  const double bytes = 4.0 * 8.0 * (double)n;
  printf("3R+1W tile walk  rows/wave=%d  tiles/step=%d  ntl=%d  block=%d  xcd-eighths=%d  %.3f ms  %7.0f GB/s  %.3f of 8 TB/s\n", RPW, UNT, (int)NTL, BS, XCD, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0);
  fflush(stdout);
}

// Cooperative tile loads: the four waves of a workgroup fetch ONE 16 x 16 tile of each stream together (thread t = element t: every wave's
// instruction is 512 contiguous bytes, three loads in flight per lane -- the shape of the fastest contiguous stream above), park it in LDS
// (double-buffered), and wave (tile mod 4) reads it back in MFMA operand order, runs the Gram-Schmidt step's two MFMA chains and stores the
// Q tile.  STORE_ALL: the result goes back through LDS and all four waves store 512 bytes each.  T tiles per workgroup.
typedef double dbl4v __attribute__((ext_vector_type(4)));
template <int T, bool STORE_ALL>
__global__ __launch_bounds__(256) void k_coop(const double *Vi, double *Q, const double *Vn, long tiles, double *sink) {
  __shared__ double s[2][3][256];
  __shared__ double so[2][256];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, k = lane >> 4;
  const long t0 = (long)blockIdx.x * T;
  if (t0 >= tiles) return;
  const long t1 = t0 + T < tiles ? t0 + T : tiles;
  dbl4v tn = {0.0, 0.0, 0.0, 0.0};
  double a = __builtin_nontemporal_load(Vi + t0 * 256 + tid), c = __builtin_nontemporal_load(Q + t0 * 256 + tid), v = __builtin_nontemporal_load(Vn + t0 * 256 + tid);
  for (long t = t0; t < t1; ++t) {
    const int buf = (int)(t - t0) & 1;
    s[buf][0][tid] = a; s[buf][1][tid] = c; s[buf][2][tid] = v;
    if (t + 1 < t1) { a = __builtin_nontemporal_load(Vi + (t + 1) * 256 + tid); c = __builtin_nontemporal_load(Q + (t + 1) * 256 + tid); v = __builtin_nontemporal_load(Vn + (t + 1) * 256 + tid); }
    __syncthreads();
    if (w == (int)((t - t0) & 3)) {
      dbl4v acc = {0.0, 0.0, 0.0, 0.0}, cin, wnew;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(s[buf][0][i * 16 + 4 * kk + k], 0.001 * (double)(kk + i), acc, 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) { cin[g] = s[buf][1][(k + 4 * g) * 16 + i]; wnew[g] = fma(-1.0, acc[g], cin[g]); }
#pragma unroll
      for (int u = 0; u < 4; ++u) tn = __builtin_amdgcn_mfma_f64_16x16x4f64(s[buf][2][(4 * u + k) * 16 + i], wnew[u], tn, 0, 0, 0);
      if (STORE_ALL) {
#pragma unroll
        for (int g = 0; g < 4; ++g) so[buf][(k + 4 * g) * 16 + i] = wnew[g];
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) __builtin_nontemporal_store(wnew[g], Q + t * 256 + (k + 4 * g) * 16 + i);
      }
    }
    if (STORE_ALL) { __syncthreads(); __builtin_nontemporal_store(so[buf][tid], Q + t * 256 + tid); }
  }
  if (tn[0] + tn[1] + tn[2] + tn[3] == 12345.678) *sink = tn[0];
}
template <int T, bool STORE_ALL>
static void run_coop(double *v, double *q, double *u, long n, double *sink) {
  const long tiles = n / 256;
  const long G = (tiles + T - 1) / T;
  const float ms = timeit([&] { hipLaunchKernelGGL((k_coop<T, STORE_ALL>), dim3((unsigned)G), dim3(256), 0, 0, v, q, u, tiles, sink); }, 20);
  const double bytes = 4.0 * 8.0 * (double)n;
  printf("3R+1W cooperative tile loads + MFMA  tiles/workgroup=%d  store_all=%d  %.3f ms  %7.0f GB/s  %.3f of 8 TB/s\n", T, (int)STORE_ALL, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0);
  fflush(stdout);
}

// X <- X + sum_i c_i V_i, K read streams + X read and written in place: the mix of panel_multi_nn (X += sum V_i Y_i, k = 5: 6 reads + 1 write)
struct MultiPtrs { const double *v[8]; };
template <int K, int U>
__global__ __launch_bounds__(256) void k_multi(MultiPtrs P, double *x, long n) {
  const long base = (long)blockIdx.x * (256 * U) + threadIdx.x;
  double a[K][U], b[U];
#pragma unroll
  for (int j = 0; j < U; ++j) { const long i = base + j * 256; if (i < n) { b[j] = x[i];
#pragma unroll
      for (int k = 0; k < K; ++k) a[k][j] = P.v[k][i]; } }
#pragma unroll
  for (int j = 0; j < U; ++j) { const long i = base + j * 256; if (i < n) { double o = b[j];
#pragma unroll
      for (int k = 0; k < K; ++k) o = fma(0.125, a[k][j], o);
      x[i] = o; } }
}
template <int K, int U>
static void run_multi(double **v, double *x, long n) {
  MultiPtrs P; for (int k = 0; k < 8; ++k) P.v[k] = v[k % K];
  const long G = (n + 256L * U - 1) / (256L * U);
  const float ms = timeit([&] { hipLaunchKernelGGL((k_multi<K, U>), dim3((unsigned)G), dim3(256), 0, 0, P, x, n); }, 10);
  const double bytes = (double)(K + 2) * 8.0 * (double)n;
  printf("%dR+1W (X in place + %d panels)  8 B/lane  U=%d  %.3f ms  %7.0f GB/s  %.3f of 8 TB/s\n", K + 1, K, U, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 8000.0);
  fflush(stdout);
}
static void multi_main(long rows, int p) {
  const long n = ((rows + 15) / 16 * 16) * p;
  double *v[5], *x;
  for (auto &q : v) { CK(hipMalloc(&q, n * 8)); CK(hipMemset(q, 0, n * 8)); }
  CK(hipMalloc(&x, n * 8)); CK(hipMemset(x, 0, n * 8));
  printf("multi-panel update streams: %ld x %d doubles per panel (%.2f GB)\n", rows, p, n * 8 / 1e9);
  run_multi<5, 1>(v, x, n); run_multi<5, 2>(v, x, n); run_multi<5, 4>(v, x, n);
  run_multi<2, 2>(v, x, n); run_multi<2, 4>(v, x, n);
  run_multi<1, 4>(v, x, n);
}

static void panel_main(long rows, int p) {
  const long n = ((rows + 15) / 16 * 16) * p;          // doubles of one panel (cfg 5: 10,077,696 x 16 = 1.29 GB)
  double *v, *q, *u, *sink;
  CK(hipMalloc(&v, n * 8)); CK(hipMalloc(&q, n * 8)); CK(hipMalloc(&u, n * 8)); CK(hipMalloc(&sink, 8));
  CK(hipMemset(v, 0, n * 8)); CK(hipMemset(q, 0, n * 8)); CK(hipMemset(u, 0, n * 8));
  printf("panel streams: %ld x %d doubles per panel (%.2f GB)\n", rows, p, n * 8 / 1e9);
#define BOTH(NAME, NR, WRITE, NTS) \
  run_panel<double, NR, 4, WRITE, NTS>(NAME, v, q, u, n, sink); run_panel<double, NR, 8, WRITE, NTS>(NAME, v, q, u, n, sink); \
  run_panel<double, NR, 16, WRITE, NTS>(NAME, v, q, u, n, sink); \
  run_panel<dbl2, NR, 2, WRITE, NTS>(NAME, v, q, u, n, sink); run_panel<dbl2, NR, 4, WRITE, NTS>(NAME, v, q, u, n, sink); \
  run_panel<dbl2, NR, 8, WRITE, NTS>(NAME, v, q, u, n, sink)
  run_coop<4, false>(v, q, u, n, sink); run_coop<16, false>(v, q, u, n, sink); run_coop<64, false>(v, q, u, n, sink);
  run_coop<4, true>(v, q, u, n, sink); run_coop<16, true>(v, q, u, n, sink); run_coop<64, true>(v, q, u, n, sink);
  run_tilewalk<256, 2, false>(v, q, u, n); run_tilewalk<256, 2, true>(v, q, u, n); run_tilewalk<256, 1, true>(v, q, u, n); run_tilewalk<256, 4, true>(v, q, u, n);
  run_tilewalk<32, 2, true>(v, q, u, n); run_tilewalk<16, 1, true>(v, q, u, n); run_tilewalk<1024, 2, true>(v, q, u, n);
  run_tilewalk<16, 1, true, 64>(v, q, u, n); run_tilewalk<16, 1, true, 128>(v, q, u, n); run_tilewalk<32, 2, true, 64>(v, q, u, n); run_tilewalk<256, 2, true, 64>(v, q, u, n);
  run_tilewalk<256, 2, true, 256, 1>(v, q, u, n); run_tilewalk<32, 2, true, 256, 1>(v, q, u, n); run_tilewalk<16, 1, true, 64, 1>(v, q, u, n);
  // the BLAS-1 kernels' shape: one or two 16-byte elements per thread, non-temporal loads and stores
#define NTL(NAME, NR, WRITE) \
  run_panel<dbl2, NR, 1, WRITE, 2>(NAME " ntl", v, q, u, n, sink); run_panel<dbl2, NR, 2, WRITE, 2>(NAME " ntl", v, q, u, n, sink); \
  run_panel<double, NR, 1, WRITE, 2>(NAME " ntl", v, q, u, n, sink); run_panel<double, NR, 4, WRITE, 2>(NAME " ntl", v, q, u, n, sink)
  NTL("1R+1W", 1, true); NTL("2R", 2, false); NTL("2R+1W", 2, true); NTL("3R", 3, false); NTL("3R+1W", 3, true);
  BOTH("1R+1W", 1, true, false);          // in place
  BOTH("1R+1W", 1, true, true);
  BOTH("2R", 2, false, false);
  BOTH("2R+1W", 2, true, false);
  BOTH("2R+1W", 2, true, true);
  BOTH("3R", 3, false, false);
  BOTH("3R+1W", 3, true, false);
  BOTH("3R+1W", 3, true, true);
}

// ------------------------------------------------------------------------------------------------ tile SpMM twin
struct TwinArgs {
  const dbl2 *val;       // 8 B per entry, contiguous per group: epg entries
  const dbl2 *slot;      // 1 B per entry
  const dbl2 *rec;       // record: 32 x 12 B + 144 x 4 B = 960 B per group
  const dbl2 *X;         // n x 16 doubles, row-major
  dbl2 *Y;
  int n1, tiles_x, tiles_y, tiles_z;
  long groups;
  int epg;               // entries per group (27 x 32 = 864 in the interior)
  int waves_total;       // persistent waves
  int pencil;            // tile rows per pencil (4)
  int xrows;             // 1: read the 144 panel rows; 0: skip them (the matrix-stream floor)
  int entries;           // 1: read the entry / record streams
  int sx, sy, sz;        // tile shape in grid points (4 x 4 x 2 = the 32-row groups of csrc/spmm_tile.hip; 4 x 4 x 4: 64 rows per group)
};

__device__ __forceinline__ void tile_of_group(const TwinArgs &a, long g, int &tx, int &ty, int &tz) {
  // pencils of `pencil` tile rows walked through all planes: for P: for tz: for ty in pencil: for tx
  const long per_pencil = (long)a.pencil * a.tiles_x * a.tiles_z;
  const long P = g / per_pencil;
  long r = g - P * per_pencil;
  const int rows_here = (int)((P + 1) * a.pencil <= a.tiles_y ? a.pencil : a.tiles_y - P * a.pencil);
  const long per_plane = (long)rows_here * a.tiles_x;
  tz = (int)(r / per_plane); r -= (long)tz * per_plane;
  ty = (int)(P * a.pencil + r / a.tiles_x);
  tx = (int)(r % a.tiles_x);
}

__global__ __launch_bounds__(256) void k_spmm_twin(TwinArgs a) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= a.waves_total) return;
  // XCD x (= workgroup number mod 8) takes the x-th of eight contiguous runs of groups
  const int xcd = blockIdx.x & 7;
  const long waves_per_xcd = a.waves_total / 8;
  const long w_in_xcd = (long)(blockIdx.x >> 3) * 4 + (threadIdx.x >> 6);
  const long g0 = a.groups * xcd / 8, g1 = a.groups * (xcd + 1) / 8;
  const int n1 = a.n1;
  for (long g = g0 + w_in_xcd; g < g1; g += waves_per_xcd) {
    int tx, ty, tz;
    tile_of_group(a, g, tx, ty, tz);
    dbl2 acc = {0.0, 0.0};
    dbl2 e[16], xr[32];
    const int rows = a.sx * a.sy * a.sz;               // 32 or 64
    const int nvec = a.epg / 2;                        // 16-byte vectors of values
    // entries: epg x 8 B = epg / 2 vectors of 16 B; slots epg / 16 vectors; record 60 (120) vectors
    if (a.entries) {
      const dbl2 *vp = a.val + g * (long)nvec;
#pragma unroll
      for (int j = 0; j < 14; ++j) { const int i = j * 64 + lane; e[j] = i < nvec ? vp[i] : dbl2{0.0, 0.0}; }
      const dbl2 *sp = a.slot + g * (long)(a.epg / 16);
      const dbl2 *rp = a.rec + g * (long)(rows * 60 / 32);
      e[14] = lane < a.epg / 16 ? sp[lane] : dbl2{0.0, 0.0};
      if (lane + 64 < a.epg / 16) { const dbl2 t = sp[lane + 64]; e[14].x += t.x; e[14].y += t.y; }
      e[15] = lane < rows * 60 / 32 ? rp[lane] : dbl2{0.0, 0.0};
      if (lane + 64 < rows * 60 / 32) { const dbl2 t = rp[lane + 64]; e[15].x += t.x; e[15].y += t.y; }
    }
    // the (sx + 2) x (sy + 2) x (sz + 2) panel rows around the tile (clipped at the faces): 8 lanes per 128-byte row
    const int wx = a.sx + 2, wy = a.sy + 2, wz = a.sz + 2, nwin = wx * wy * wz;
    if (a.xrows) {
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        const int r = j * 8 + (lane >> 3);
        xr[j] = dbl2{0.0, 0.0};
        if (r < nwin) {
          const int dx = r % wx, dy = (r / wx) % wy, dz = r / (wx * wy);
          int x = tx * a.sx - 1 + dx, y = ty * a.sy - 1 + dy, z = tz * a.sz - 1 + dz;
          x = x < 0 ? 0 : (x >= n1 ? n1 - 1 : x); y = y < 0 ? 0 : (y >= n1 ? n1 - 1 : y); z = z < 0 ? 0 : (z >= n1 ? n1 - 1 : z);
          const long row = (long)x + (long)n1 * ((long)y + (long)n1 * z);
          xr[j] = a.X[row * 8 + (lane & 7)];
        }
      }
    }
    if (a.entries) {
#pragma unroll
      for (int j = 0; j < 16; ++j) { acc.x += e[j].x; acc.y += e[j].y; }
    }
    if (a.xrows) {
#pragma unroll
      for (int j = 0; j < 32; ++j) { acc.x += xr[j].x; acc.y += xr[j].y; }
    }
    // Y: the tile's rows, 8 rows per instruction
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = j * 8 + (lane >> 3);
      if (r < rows) {
        const int x = tx * a.sx + r % a.sx, y = ty * a.sy + (r / a.sx) % a.sy, z = tz * a.sz + r / (a.sx * a.sy);
        const long row = (long)x + (long)n1 * ((long)y + (long)n1 * z);
        a.Y[row * 8 + (lane & 7)] = acc;
      }
    }
  }
}

static void spmm_shape(int n1, int sx, int sy, int sz);
static void spmm_main(int n1) {
  spmm_shape(n1, 4, 4, 2);          // the groups of csrc/spmm_tile.hip
  spmm_shape(n1, 4, 4, 4);          // 64 rows per group: 216 instead of 2 x 144 panel rows per 64 rows
  spmm_shape(n1, 8, 4, 2);          // 64 rows per group, 10 x 6 x 4 = 240 panel rows
}
static void spmm_shape(int n1, int sx, int sy, int sz) {
  TwinArgs a;
  a.sx = sx; a.sy = sy; a.sz = sz;
  a.n1 = n1; a.tiles_x = n1 / sx; a.tiles_y = n1 / sy; a.tiles_z = n1 / sz;
  a.groups = (long)a.tiles_x * a.tiles_y * a.tiles_z;
  a.epg = 27 * sx * sy * sz; a.pencil = 4;
  const long n = (long)n1 * n1 * n1;
  double *val, *slot, *rec, *X, *Y;
  CK(hipMalloc(&val, a.groups * a.epg * 8)); CK(hipMalloc(&slot, a.groups * a.epg)); CK(hipMalloc(&rec, a.groups * 1920));
  CK(hipMalloc(&X, n * 128)); CK(hipMalloc(&Y, n * 128));
  CK(hipMemset(val, 0, a.groups * a.epg * 8)); CK(hipMemset(slot, 0, a.groups * a.epg)); CK(hipMemset(rec, 0, a.groups * 1920));
  CK(hipMemset(X, 0, n * 128)); CK(hipMemset(Y, 0, n * 128));
  a.val = (const dbl2 *)val; a.slot = (const dbl2 *)slot; a.rec = (const dbl2 *)rec; a.X = (const dbl2 *)X; a.Y = (dbl2 *)Y;
  const double nnz = 27.0 * n - 0.0;                  // the real operator has 7n - ... fewer at the faces; the twin streams 27 per row
  const double stream = (double)a.groups * (a.epg * 9.0 + 30.0 * sx * sy * sz);
  const double alg = 12.0 * nnz + 4.0 * (n + 1) + 2.0 * 128.0 * n;
  printf("tile SpMM twin: %d^3 rows, tiles %d x %d x %d (%d panel rows per %d rows), %ld groups, matrix-side stream %.2f GB (9 B per entry + 30 B per row of records), X + Y %.2f GB, algorithmic (SURVEY 8d) %.3f GB\n",
         n1, sx, sy, sz, (sx + 2) * (sy + 2) * (sz + 2), sx * sy * sz, a.groups, stream / 1e9, 2.0 * 128.0 * n / 1e9, alg / 1e9);
  for (int mode = 0; mode < 3; ++mode) {
    a.entries = mode != 1; a.xrows = mode != 0;
    for (int wpc : {4, 6, 8, 10, 12, 16}) {
      a.waves_total = 256 * wpc;
      const int blocks = a.waves_total / 4;
      const float ms = timeit([&] { hipLaunchKernelGGL(k_spmm_twin, dim3(blocks), dim3(256), 0, 0, a); }, 10);
      printf("%-34s waves/CU=%2d  %.3f ms  (algorithmic bytes / time = %.0f GB/s = %.3f of 8 TB/s)\n",
             mode == 0 ? "entries + records + Y" : (mode == 1 ? "panel rows + Y" : "entries + records + panel rows + Y"), wpc, ms,
             alg / ms / 1e6, alg / ms / 1e6 / 8000.0);
      fflush(stdout);
    }
  }
  CK(hipFree(val)); CK(hipFree(slot)); CK(hipFree(rec)); CK(hipFree(X)); CK(hipFree(Y));
}

// ------------------------------------------------------------------------------------------------ tile SpMM twin, SLIDING windows
// The traffic of the shipped kernel (spmm_tile2_kernel with runs of <= 27 groups along k) and of its look-ahead form (round 6): a
// wave walks a run of `run_len` 4 x 4 x 2 tiles along k for one (tx, ty); the first group of a run reads the whole 6 x 6 x 4 box of
// panel rows, every later one only the two NEW planes (72 rows); entries + records + Y as in k_spmm_twin.  `ahead` = 0: the loads
// of a group are issued, waited for and folded before the next group's are issued (what a single window enforces); 1: the next
// group's loads are issued BEFORE this group's are folded (what the look-ahead slot assignment allows).  Runs are numbered with
// tx fastest, then ty, the pieces of a line slowest; XCD x takes the x-th eighth of the runs.
struct SlideArgs {
  TwinArgs t;
  int run_len, pieces;   // groups per run, runs per line of tiles along z
  long runs;
  int ahead;
  int csr_val;           // 1: the value stream in CSR row order, read as the kernel reads it (strided 64-byte pieces); 0: contiguous per group
};
template <int NX> struct SlideLoads { dbl2 e[9]; dbl2 xr[NX]; };      // 4 x 4 x 2 tiles: 432 value vectors = 7 per lane, + slots + record
typedef double dbl2u8 __attribute__((ext_vector_type(2), aligned(8)));
template <int NX>
__device__ __forceinline__ void slide_issue(const SlideArgs &s, long g, int tx, int ty, int tz, int lane, SlideLoads<NX> &L) {
  const TwinArgs &a = s.t;
  const int nvec = a.epg / 2, n1 = a.n1;
  const dbl2 *vp = a.val + g * (long)nvec;
  if (s.csr_val) {
    // the value stream as the KERNEL reads it out of the CSR arrays: 27 doubles per row in row order; per instruction 16 rows of the tile
    // (4 consecutive rows x 4 lines), 4 lanes per row, 64 bytes of each row (entries 8 f .. 8 f + 7) -- 16 pieces of 64 B at 216-byte
    // and line / plane strides instead of one contiguous KiB (7 of the 8 instructions of a group modelled: f = 3 reads the last 3 entries)
    const double *v0 = reinterpret_cast<const double *>(a.val);
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int q = j >> 2, f = j & 3, sub = lane >> 2, c = lane & 3, row = q * 16 + sub;
      const int di = row & 3, dj = (row >> 2) & 3, dk = row >> 4;
      const long R = (long)(tx * a.sx + di) + (long)n1 * ((long)(ty * a.sy + dj) + (long)n1 * (tz * a.sz + dk));
      L.e[j] = *reinterpret_cast<const dbl2u8 *>(v0 + R * 27 + 2 * c + 8 * f);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 7; ++j) { const int i = j * 64 + lane; L.e[j] = i < nvec ? vp[i] : dbl2{0.0, 0.0}; }
  }
  const dbl2 *sp = a.slot + g * (long)(a.epg / 16);
  const dbl2 *rp = a.rec + g * 60L;
  L.e[7] = lane < a.epg / 16 ? sp[lane] : dbl2{0.0, 0.0};
  L.e[8] = lane < 60 ? rp[lane] : dbl2{0.0, 0.0};
  // panel rows: box (sx + 2) x (sy + 2) x (sz + 2), NX = 18: all of it (first group of a run); NX = 9: only its last sz planes
  const int wx = a.sx + 2, wy = a.sy + 2, wz = a.sz + 2;
  const int z0 = NX == 18 ? 0 : 2, nwin = wx * wy * (wz - z0);
#pragma unroll
  for (int j = 0; j < NX; ++j) {
    const int r = j * 8 + (lane >> 3);
    L.xr[j] = dbl2{0.0, 0.0};
    if (r < nwin) {
      const int dx = r % wx, dy = (r / wx) % wy, dz = z0 + r / (wx * wy);
      int x = tx * a.sx - 1 + dx, y = ty * a.sy - 1 + dy, z = tz * a.sz - 1 + dz;
      x = x < 0 ? 0 : (x >= n1 ? n1 - 1 : x); y = y < 0 ? 0 : (y >= n1 ? n1 - 1 : y); z = z < 0 ? 0 : (z >= n1 ? n1 - 1 : z);
      const long row = (long)x + (long)n1 * ((long)y + (long)n1 * z);
      L.xr[j] = a.X[row * 8 + (lane & 7)];
    }
  }
}
template <int NX>
__device__ __forceinline__ void slide_fold(const SlideArgs &s, int tx, int ty, int tz, int lane, const SlideLoads<NX> &L) {
  const TwinArgs &a = s.t;
  dbl2 acc = {0.0, 0.0};
#pragma unroll
  for (int j = 0; j < 9; ++j) { acc.x += L.e[j].x; acc.y += L.e[j].y; }
#pragma unroll
  for (int j = 0; j < NX; ++j) { acc.x += L.xr[j].x; acc.y += L.xr[j].y; }
  const int rows = a.sx * a.sy * a.sz, n1 = a.n1;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = j * 8 + (lane >> 3);
    if (r < rows) {
      const int x = tx * a.sx + r % a.sx, y = ty * a.sy + (r / a.sx) % a.sy, z = tz * a.sz + r / (a.sx * a.sy);
      const long row = (long)x + (long)n1 * ((long)y + (long)n1 * z);
      a.Y[row * 8 + (lane & 7)] = acc;
    }
  }
}
template <bool AHEAD>
__global__ __launch_bounds__(256) void k_spmm_slide(SlideArgs s) {
  const TwinArgs &a = s.t;
  const int lane = threadIdx.x & 63;
  const int xcd = blockIdx.x & 7;
  const long waves_per_xcd = a.waves_total / 8;
  const long w_in_xcd = (long)(blockIdx.x >> 3) * 4 + (threadIdx.x >> 6);
  if ((long)blockIdx.x * 4 + (threadIdx.x >> 6) >= a.waves_total) return;
  const long r0 = s.runs * xcd / 8, r1 = s.runs * (xcd + 1) / 8;
  const long nlines = (long)a.tiles_x * a.tiles_y;
  for (long run = r0 + w_in_xcd; run < r1; run += waves_per_xcd) {
    const long line = run % nlines, piece = run / nlines;
    const int tx = (int)(line % a.tiles_x), ty = (int)(line / a.tiles_x);
    const int tz0 = (int)piece * s.run_len;
    const int len = tz0 + s.run_len <= a.tiles_z ? s.run_len : a.tiles_z - tz0;
    if (len <= 0) continue;
    const long gbase = ((long)piece * nlines + line) * s.run_len;      // records of a run are consecutive
    SlideLoads<9> A;
    {
      SlideLoads<18> F;                                                // the first group of a run fills the window
      slide_issue<18>(s, gbase, tx, ty, tz0, lane, F);
      if (AHEAD && len > 1) slide_issue<9>(s, gbase + 1, tx, ty, tz0 + 1, lane, A);
      slide_fold<18>(s, tx, ty, tz0, lane, F);
      if (!AHEAD && len > 1) slide_issue<9>(s, gbase + 1, tx, ty, tz0 + 1, lane, A);
    }
    if (AHEAD) {
      SlideLoads<9> B;
      for (int t = 1; t < len; t += 2) {                   // two groups per trip: the buffers swap roles without a copy
        if (t + 1 < len) slide_issue<9>(s, gbase + t + 1, tx, ty, tz0 + t + 1, lane, B);
        slide_fold<9>(s, tx, ty, tz0 + t, lane, A);
        if (t + 1 < len) {
          if (t + 2 < len) slide_issue<9>(s, gbase + t + 2, tx, ty, tz0 + t + 2, lane, A);
          slide_fold<9>(s, tx, ty, tz0 + t + 1, lane, B);
        }
      }
    } else {
      for (int t = 1; t < len; ++t) {
        slide_fold<9>(s, tx, ty, tz0 + t, lane, A);
        if (t + 1 < len) slide_issue<9>(s, gbase + t + 1, tx, ty, tz0 + t + 1, lane, A);
      }
    }
  }
}
static void spmmslide_main(int n1) {
  SlideArgs s;
  TwinArgs &a = s.t;
  a.sx = 4; a.sy = 4; a.sz = 2;
  a.n1 = n1; a.tiles_x = n1 / 4; a.tiles_y = n1 / 4; a.tiles_z = n1 / 2;
  a.groups = (long)a.tiles_x * a.tiles_y * a.tiles_z;
  a.epg = 27 * 32; a.pencil = 4; a.entries = 1; a.xrows = 1;
  const long n = (long)n1 * n1 * n1;
  double *val, *slot, *rec, *X, *Y;
  const long gpad = a.groups + a.groups / 8 + 64;                     // records padded to whole runs (>= n * 27 doubles + slack for the CSR-order reads)
  CK(hipMalloc(&val, gpad * a.epg * 8)); CK(hipMalloc(&slot, gpad * a.epg)); CK(hipMalloc(&rec, gpad * 1920));
  CK(hipMalloc(&X, n * 128)); CK(hipMalloc(&Y, n * 128));
  CK(hipMemset(val, 0, gpad * a.epg * 8)); CK(hipMemset(slot, 0, gpad * a.epg)); CK(hipMemset(rec, 0, gpad * 1920));
  CK(hipMemset(X, 0, n * 128)); CK(hipMemset(Y, 0, n * 128));
  a.val = (const dbl2 *)val; a.slot = (const dbl2 *)slot; a.rec = (const dbl2 *)rec; a.X = (const dbl2 *)X; a.Y = (dbl2 *)Y;
  const double alg = 12.0 * 27.0 * n + 4.0 * (n + 1) + 2.0 * 128.0 * n;
  printf("tile SpMM twin, SLIDING windows: %d^3 rows, 4 x 4 x 2 tiles walked along z in runs; per group 9 B per entry + 960 B of records, 72 new panel rows (144 at the start of a run), 32 rows of Y; algorithmic (SURVEY 8d) %.3f GB\n", n1, alg / 1e9);
  s.csr_val = 0;
  for (int run_len : {27, 54, 108, -27}) {
    if (run_len < 0) { s.csr_val = 1; run_len = -run_len; printf("the same with the value stream in CSR row order, read in the kernel's 64-byte pieces (2.18 GB of values + 27 doubles of slack):\n"); }
    s.pieces = (a.tiles_z + run_len - 1) / run_len;
    s.run_len = (a.tiles_z + s.pieces - 1) / s.pieces;
    s.runs = (long)a.tiles_x * a.tiles_y * s.pieces;
    for (int ahead = 0; ahead < 2; ++ahead) {
      s.ahead = ahead;
      for (int wpc : {4, 8, 10, 12, 16}) {
        a.waves_total = 256 * wpc;
        const int blocks = a.waves_total / 4;
        const float ms = ahead ? timeit([&] { hipLaunchKernelGGL(k_spmm_slide<true>, dim3(blocks), dim3(256), 0, 0, s); }, 10)
                               : timeit([&] { hipLaunchKernelGGL(k_spmm_slide<false>, dim3(blocks), dim3(256), 0, 0, s); }, 10);
        printf("runs of %3d groups  %-26s waves/CU=%2d  %.3f ms  (algorithmic bytes / time = %.0f GB/s = %.3f of 8 TB/s)\n", s.run_len,
               ahead ? "next group issued first" : "one group at a time", wpc, ms, alg / ms / 1e6, alg / ms / 1e6 / 8000.0);
        fflush(stdout);
      }
    }
  }
  CK(hipFree(val)); CK(hipFree(slot)); CK(hipFree(rec)); CK(hipFree(X)); CK(hipFree(Y));
}

// ------------------------------------------------------------------------------------------------ tile SpMM twin, banded + random operator
// The traffic of the tile SpMM on the non-stencil benchmark operator (csrc/gen_irregular.cpp: band of half-width 13, three
// links per row to partners inside its block of 2^20 rows; 10.5 M rows, p = 16): groups of 32 consecutive rows; per group the
// matrix stream (9 B per entry, 27 entries per row here; records), the 32 + 26 consecutive panel rows of the band, 96 panel rows
// at hashed positions inside the group's block of 2^20 rows (one 128-byte line each: a panel row IS a line at p = 16), 32 rows
// of Y.  A wave takes whole groups with every load in flight, as k_spmm_twin.  links = 0 / 1 switches the hashed rows off / on.
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
  z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ull; z ^= z >> 27; z *= 0x94d049bb133111ebull; z ^= z >> 31; return z;
}
__global__ __launch_bounds__(256) void k_spmm_irr(TwinArgs a, long n, int links, int round_robin) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= a.waves_total) return;
  const int xcd = blockIdx.x & 7;
  const long waves_per_xcd = a.waves_total / 8;
  const long w_in_xcd = (long)(blockIdx.x >> 3) * 4 + (threadIdx.x >> 6);
  // round_robin = 0: XCD x takes the x-th eighth of the groups (eight fronts); 1: one front over the matrix, the groups dealt to all waves in turn
  const long g0 = round_robin ? 0 : a.groups * xcd / 8, g1 = round_robin ? a.groups : a.groups * (xcd + 1) / 8;
  const long gstart = round_robin ? wave : g0 + w_in_xcd, gstep = round_robin ? a.waves_total : waves_per_xcd;
  for (long g = gstart; g < g1; g += gstep) {
    dbl2 e[9], xr[20];
    const int nvec = a.epg / 2;
    const dbl2 *vp = a.val + g * (long)nvec;
#pragma unroll
    for (int j = 0; j < 7; ++j) { const int i = j * 64 + lane; e[j] = i < nvec ? vp[i] : dbl2{0.0, 0.0}; }
    e[7] = lane < a.epg / 16 ? (a.slot + g * (long)(a.epg / 16))[lane] : dbl2{0.0, 0.0};
    e[8] = lane < 60 ? (a.rec + g * 60L)[lane] : dbl2{0.0, 0.0};
    const long row0 = g * 32;
#pragma unroll
    for (int j = 0; j < 20; ++j) {
      const int r = j * 8 + (lane >> 3);                 // 0..57: the band; 58..153: the links
      xr[j] = dbl2{0.0, 0.0};
      long row = -1;
      if (r < 58) row = row0 - 13 + r;
      else if (r < 154 && links) row = (row0 & ~((1L << 20) - 1)) | (long)(mix64((unsigned long long)(g * 96 + (r - 58))) & ((1ull << 20) - 1));
      if (row >= 0 && row < n) xr[j] = a.X[row * 8 + (lane & 7)];
    }
    dbl2 acc = {0.0, 0.0};
#pragma unroll
    for (int j = 0; j < 9; ++j) { acc.x += e[j].x; acc.y += e[j].y; }
#pragma unroll
    for (int j = 0; j < 20; ++j) { acc.x += xr[j].x; acc.y += xr[j].y; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long row = row0 + j * 8 + (lane >> 3);
      if (row < n) a.Y[row * 8 + (lane & 7)] = acc;
    }
  }
}
static void spmmirr_main() {
  TwinArgs a;
  const long n = 10L * (1 << 20);
  a.groups = n / 32; a.epg = 27 * 32; a.n1 = 0; a.sx = a.sy = a.sz = 0;
  double *val, *slot, *rec, *X, *Y;
  CK(hipMalloc(&val, a.groups * a.epg * 8)); CK(hipMalloc(&slot, a.groups * a.epg)); CK(hipMalloc(&rec, a.groups * 1920));
  CK(hipMalloc(&X, n * 128)); CK(hipMalloc(&Y, n * 128));
  CK(hipMemset(val, 0, a.groups * a.epg * 8)); CK(hipMemset(slot, 0, a.groups * a.epg)); CK(hipMemset(rec, 0, a.groups * 1920));
  CK(hipMemset(X, 0, n * 128)); CK(hipMemset(Y, 0, n * 128));
  a.val = (const dbl2 *)val; a.slot = (const dbl2 *)slot; a.rec = (const dbl2 *)rec; a.X = (const dbl2 *)X; a.Y = (dbl2 *)Y;
  const double nnz = 27.0 * n, alg = 12.0 * nnz + 4.0 * (n + 1) + 2.0 * 128.0 * n;
  const double moved0 = (double)a.groups * (a.epg * 9.0 + 960.0) + 2.0 * 128.0 * n, moved1 = moved0 + (double)a.groups * 96.0 * 128.0;
  printf("tile SpMM twin, banded + random: %ld rows, p = 16; algorithmic (SURVEY 8d, 27 entries per row) %.3f GB; bytes requested without the links %.3f GB, with the 96 hashed panel rows per group %.3f GB\n",
         n, alg / 1e9, moved0 / 1e9, moved1 / 1e9);
  for (int links = 0; links < 2; ++links)
    for (int rr = 0; rr < 2; ++rr)
      for (int wpc : {4, 8, 16}) {
        a.waves_total = 256 * wpc;
        const int blocks = a.waves_total / 4;
        const float ms = timeit([&] { hipLaunchKernelGGL(k_spmm_irr, dim3(blocks), dim3(256), 0, 0, a, n, links, rr); }, 10);
        printf("%-13s %-28s waves/CU=%2d  %.3f ms  (algorithmic bytes / time = %.0f GB/s = %.3f of 8 TB/s; requested bytes / time = %.0f GB/s)\n",
               links ? "band + links" : "band only", rr ? "one front (round-robin)" : "eight fronts (XCD eighths)", wpc, ms, alg / ms / 1e6, alg / ms / 1e6 / 8000.0,
               (links ? moved1 : moved0) / ms / 1e6);
        fflush(stdout);
      }
  CK(hipFree(val)); CK(hipFree(slot)); CK(hipFree(rec)); CK(hipFree(X)); CK(hipFree(Y));
}

// ------------------------------------------------------------------------------------------------ cg! update folded into the SpMV?
// fold `x += alpha p_old` and `p = r + beta p_old` into the SpMV of the NEXT iteration (the
// product gathers r and p_old instead of p, writes p_new, Ap and x): matrix + 72n instead of matrix + 80n bytes per iteration.
// Twin of both forms on the 7-point 512^3 row walk with the coded operator's 63 B of matrix data per row (streamed as 64):
//   now     : gather 1 vector at the 7 offsets, write y                                    (+ kernel C: 3 reads, 2 writes = 40n)
//   folded  : gather 2 vectors at the 7 offsets, read x, write p_new, Ap, x                 (kernel C gone)
template <int NG, int NEXTRA_R, int NW>
__global__ __launch_bounds__(256) void k_cgfold(const double *__restrict__ g0, const double *__restrict__ g1, const dbl2 *__restrict__ st,
                                                const double *__restrict__ xr, double *__restrict__ w0, double *__restrict__ w1,
                                                double *__restrict__ w2, long n, int n1) {
  const long tile = blockIdx.x;
  const long i = tile * 256 + threadIdx.x;
  const long d[7] = {-(long)n1 * n1, -(long)n1, -1, 0, 1, (long)n1, (long)n1 * n1};
  dbl2 sv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) sv[q] = st[(tile * 4 + q) * 256 + threadIdx.x];
  double a[7], b[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) { const long j = i + d[k]; const bool ok = j >= 0 && j < n; a[k] = ok ? g0[j] : 0.0; b[k] = (NG == 2 && ok) ? g1[j] : 0.0; }
  double acc = NEXTRA_R ? xr[i] : 0.0;
#pragma unroll
  for (int k = 0; k < 7; ++k) acc += fma(0.5, b[k], a[k]);
#pragma unroll
  for (int q = 0; q < 4; ++q) acc += sv[q].x + sv[q].y;
  w0[i] = acc;
  if (NW >= 2) w1[i] = acc + 1.0;
  if (NW >= 3) w2[i] = acc + 2.0;
}

static void cgfold_main(int n1) {
  const long n = (long)n1 * n1 * n1;
  double *v[7]; dbl2 *st;
  for (auto &p : v) { CK(hipMalloc(&p, n * 8)); CK(hipMemset(p, 0, n * 8)); }
  CK(hipMalloc(&st, n * 64)); CK(hipMemset(st, 0, n * 64));
  const unsigned G = (unsigned)(n / 256);
  printf("cg! update folded into the SpMV? twin on the 7-point %d^3 row walk, 64 B of matrix data per row\n", n1);
  for (int rep = 0; rep < 2; ++rep) {
    const float t_now = timeit([&] { hipLaunchKernelGGL((k_cgfold<1, 0, 1>), dim3(G), dim3(256), 0, 0, v[0], v[1], st, v[2], v[3], v[4], v[5], n, n1); }, 10);
    const float t_fold = timeit([&] { hipLaunchKernelGGL((k_cgfold<2, 1, 3>), dim3(G), dim3(256), 0, 0, v[0], v[1], st, v[2], v[3], v[4], v[5], n, n1); }, 10);
    const float t_g2 = timeit([&] { hipLaunchKernelGGL((k_cgfold<2, 0, 1>), dim3(G), dim3(256), 0, 0, v[0], v[1], st, v[2], v[3], v[4], v[5], n, n1); }, 10);
    // kernel C of today: x += a p ; p = r + b p  (3 reads, 2 writes)
    const long nv = n / 2;
    const long GC = (nv + 256L * 4 - 1) / (256L * 4);
    const float t_c = timeit([&] { hipLaunchKernelGGL((k_panel<dbl2, 3, 4, true, false>), dim3((unsigned)GC), dim3(256), 0, 0, (const dbl2 *)v[0], (dbl2 *)v[1], (const dbl2 *)v[2], nv, v[6]);
                                   hipLaunchKernelGGL((k_panel<dbl2, 2, 4, true, false>), dim3((unsigned)GC), dim3(256), 0, 0, (const dbl2 *)v[3], (dbl2 *)v[4], (const dbl2 *)v[2], nv, v[6]); }, 10);
    printf("SpMV twin now (1 gathered vector, 1 store)             %.3f ms\n", t_now);
    printf("SpMV twin, 2 gathered vectors, 1 store                 %.3f ms\n", t_g2);
    printf("SpMV twin folded (2 gathered, + read x, 3 stores)      %.3f ms   (+%.3f ms over now)\n", t_fold, t_fold - t_now);
    printf("streams of today's update kernel as 3R+1W then 2R+1W (56n; the real kernel moves 40n in one pass: 0.875 ms, profiles/r04_rocprofv3_kernel_stats.csv)  %.3f ms\n", t_c);
    fflush(stdout);
  }
}

int main(int argc, char **argv) {
  const char *mode = argc > 1 ? argv[1] : "panel";
  if (strcmp(mode, "panel") == 0) panel_main(argc > 2 ? atol(argv[2]) : 10077696L, 16);
  else if (strcmp(mode, "spmm") == 0) spmm_main(argc > 2 ? atoi(argv[2]) : 216);
  else if (strcmp(mode, "multi") == 0) multi_main(argc > 2 ? atol(argv[2]) : 10077696L, 16);
  else if (strcmp(mode, "spmmirr") == 0) spmmirr_main();
  else if (strcmp(mode, "spmmslide") == 0) spmmslide_main(argc > 2 ? atoi(argv[2]) : 216);
  else if (strcmp(mode, "cgfold") == 0) cgfold_main(argc > 2 ? atoi(argv[2]) : 512);
  else { printf("usage: streamfloor panel [rows] | multi [rows] | spmm [n1] | spmmslide [n1] | spmmirr | cgfold [n1]\n"); return 2; }
  return 0;
}
