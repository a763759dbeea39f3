/*
 * krylov_oracle.c -- CPU restatement of the Krylov.jl hot path (see krylov_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: the checker for the HIP path, never the product.
 *
 * Arithmetic conventions (the reference delegates these to unpinned third-party
 * libraries, so a FIXED order is documented here instead):
 *   - SpMV: per row, entries in stored (ascending column) order, each term is a
 *     rounded product followed by a rounded add (no FMA), accumulator starts at
 *     +0.0.  This is what SparseArrays.mul! does for a symmetric CSC matrix.
 *   - dot / nrm2: sequential accumulation in x87 extended precision (long double,
 *     64-bit mantissa), one rounding to double at the end; nrm2 = sqrt(dot(x,x))
 *     (BLAS nrm2 differs only by overflow-safe scaling).
 *   - axpy / axpby: one fused multiply-add per element, y <- fma(s, x, y) and
 *     y <- fma(s, x, t*y) -- OpenBLAS' Haswell+ kernels use vfmadd as well.
 *   - dense QR: unblocked LAPACK (DGEQR2 / DORG2R / DORM2R); for p <= 32 columns
 *     that is exactly the code path DGEQRF/DORGQR/DORMQR take (block size 32).
 *
 * Build: gcc -O2 -fPIC -shared -fopenmp -ffp-contract=off krylov_oracle.c -lm
 */
#include "krylov_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 1;

void ko_set_threads(int nthreads) {
  g_threads = nthreads < 1 ? 1 : nthreads;
#ifdef _OPENMP
  omp_set_num_threads(g_threads);
#endif
}
int ko_get_threads(void) { return g_threads; }

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ===================================================================== *
 *  CSR generators
 * ===================================================================== */

static int csr_alloc(ko_csr *A, int64_t n, int64_t nnz) {
  A->n = n; A->nnz = nnz;
  A->rowptr = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
  A->col = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1));
  A->val = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
  if (!A->rowptr || !A->col || !A->val) { ko_csr_free(A); return -1; }
  return 0;
}

void ko_csr_free(ko_csr *A) {
  if (!A) return;
  free(A->rowptr); free(A->col); free(A->val);
  A->rowptr = NULL; A->col = NULL; A->val = NULL; A->n = A->nnz = 0;
}

/* Generic 3-D stencil assembler on an n1 x n2 x n3 grid, index
 * i1 + n1*i2 + n1*n2*i3 (i1 fastest), Dirichlet truncation at the faces.
 * coef(d1,d2,d3) gives the entry for offset (d1,d2,d3), 0 => structurally absent.
 * Entries are emitted in ascending column order. */
typedef double (*stencil_fn)(int d1, int d2, int d3);

static int csr_stencil3d(int n1, int n2, int n3, int reach, stencil_fn coef, ko_csr *A) {
  int64_t n = (int64_t)n1 * n2 * n3;
  /* nonzero offsets in ascending column order (d3 slowest) */
  int od1[125], od2[125], od3[125], noff = 0;
  double ov[125];
  for (int d3 = -reach; d3 <= reach; d3++)
    for (int d2 = -reach; d2 <= reach; d2++)
      for (int d1 = -reach; d1 <= reach; d1++) {
        double v = coef(d1, d2, d3);
        if (v == 0.0) continue;
        od1[noff] = d1; od2[noff] = d2; od3[noff] = d3; ov[noff] = v; noff++;
      }
  A->n = n; A->nnz = 0; A->col = NULL; A->val = NULL;
  A->rowptr = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
  if (!A->rowptr) return -1;
  /* pass 1: entries per row (parallel), then a serial prefix sum */
#pragma omp parallel for schedule(static)
  for (int64_t row = 0; row < n; row++) {
    int i1 = (int)(row % n1), i2 = (int)((row / n1) % n2), i3 = (int)(row / ((int64_t)n1 * n2));
    int cnt = 0;
    for (int o = 0; o < noff; o++) {
      int j1 = i1 + od1[o], j2 = i2 + od2[o], j3 = i3 + od3[o];
      cnt += (j1 >= 0 && j1 < n1 && j2 >= 0 && j2 < n2 && j3 >= 0 && j3 < n3);
    }
    A->rowptr[row + 1] = cnt;
  }
  A->rowptr[0] = 0;
  for (int64_t row = 0; row < n; row++) A->rowptr[row + 1] += A->rowptr[row];
  int64_t nnz = A->rowptr[n];
  A->nnz = nnz;
  A->col = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1));
  A->val = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
  if (!A->col || !A->val) { ko_csr_free(A); return -1; }
  /* pass 2: fill (parallel => first touch spreads the pages over the NUMA nodes) */
#pragma omp parallel for schedule(static)
  for (int64_t row = 0; row < n; row++) {
    int i1 = (int)(row % n1), i2 = (int)((row / n1) % n2), i3 = (int)(row / ((int64_t)n1 * n2));
    int64_t k = A->rowptr[row];
    for (int o = 0; o < noff; o++) {
      int j1 = i1 + od1[o], j2 = i2 + od2[o], j3 = i3 + od3[o];
      if (j1 < 0 || j1 >= n1 || j2 < 0 || j2 >= n2 || j3 < 0 || j3 >= n3) continue;
      A->col[k] = (int32_t)((int64_t)j1 + (int64_t)n1 * j2 + (int64_t)n1 * n2 * j3);
      A->val[k] = ov[o];
      k++;
    }
  }
  return 0;
}

/* get_div_grad(n1,n2,n3) = Div*Div' with ddx(n)*ddx(n)' = tridiag(-1,2,-1)
 * (test/get_div_grad.jl:8-25): diagonal 6, the six face neighbours -1. */
static double coef_poisson(int d1, int d2, int d3) {
  int a = abs(d1) + abs(d2) + abs(d3);
  if (a == 0) return 6.0;
  if (a == 1) return -1.0;
  return 0.0;
}
int ko_csr_poisson3d(int n1, int n2, int n3, ko_csr *A) {
  return csr_stencil3d(n1, n2, n3, 1, coef_poisson, A);
}

/* kron_unsymmetric(n) (test/test_utils.jl:160-169):
 *   T = tridiag(-1, 3, -2);  A <- kron(T,I)+kron(I,T) applied TWICE, i.e.
 *   A = T(x)I(x)I + 2 * I(x)T(x)I + I(x)I(x)T      (first Kronecker factor = slowest index)
 * so the middle dimension is counted twice: diagonal 12; offsets
 *   slowest (i3): lower -1, upper -2 ; middle (i2): lower -2, upper -4 ; fastest (i1): lower -1, upper -2. */
static double coef_kron_unsym(int d1, int d2, int d3) {
  int a = abs(d1) + abs(d2) + abs(d3);
  if (a == 0) return 12.0;
  if (a != 1) return 0.0;
  if (d1 == -1) return -1.0;
  if (d1 == 1) return -2.0;
  if (d2 == -1) return -2.0;
  if (d2 == 1) return -4.0;
  if (d3 == -1) return -1.0;
  return -2.0; /* d3 == 1 */
}
int ko_csr_kron_unsymmetric(int n1, ko_csr *A) {
  return csr_stencil3d(n1, n1, n1, 1, coef_kron_unsym, A);
}

/* cfg-5 synthetic "SuiteSparse-shaped" operator (no reference generator exists;
 * SuiteSparse is unavailable offline): 27-point stencil, diagonal 16, neighbour
 * weight w = 1 (faces), 0.5 (edges), 0.25 (corners), scaled by 1.25 when the
 * leading nonzero offset component (slowest index first) is positive and by 0.75
 * otherwise => nonsymmetric, strictly diagonally dominant (sum |offdiag| = 14 < 16). */
static double coef_stencil27(int d1, int d2, int d3) {
  int a = abs(d1) + abs(d2) + abs(d3);
  if (a == 0) return 16.0;
  double w = (a == 1) ? 1.0 : (a == 2 ? 0.5 : 0.25); /* faces, edges, corners */
  int lead = d3 != 0 ? d3 : (d2 != 0 ? d2 : d1);
  return -w * (lead > 0 ? 1.25 : 0.75);
}
int ko_csr_stencil27_unsym(int n1, ko_csr *A) {
  return csr_stencil3d(n1, n1, n1, 1, coef_stencil27, A);
}

/* "banded + random, fixed seed": the non-stencil benchmark operator (stands in for the SuiteSparse
 * matrices of benchmark/cg_bmark.jl:29-54).  Restated from its definition in the header of
 * krylov.jl_amd/csrc/gen_irregular.cpp, independently of that code: candidate lists per row, then an insertion sort.
 * tests/test_gpu_primitives.py compares the two generators array by array. */
static uint64_t br_mix(uint64_t z) {
  z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ull;
  z ^= z >> 27; z *= 0x94d049bb133111ebull;
  z ^= z >> 31;
  return z;
}
typedef struct {
  int64_t n; int hb, K, B, h, unsym, ndense;
  uint64_t seed, mask, a1[16], a2[16], a1i[16], a2i[16];
  int64_t stride;
} br_spec;
static uint64_t br_key(const br_spec *S, int64_t lo, int64_t hi) {
  return br_mix(S->seed ^ ((uint64_t)lo * 0x9E3779B97F4A7C15ull + (uint64_t)hi));
}
static uint64_t br_inverse(uint64_t a, uint64_t mask) {
  /* a odd: bit-by-bit solve of a * x = 1 (mod 2^B) */
  uint64_t x = 0, prod = 0;
  for (uint64_t bit = 1; bit != 0 && bit <= mask; bit <<= 1) {
    if (((prod ^ 1) & bit) != 0) { x |= bit; prod += a * bit; }
  }
  return x & mask;
}
static int64_t br_partner(const br_spec *S, int k, int64_t r) {
  int64_t blk = r >> S->B;
  if (blk >= (S->n >> S->B)) return -1;
  uint64_t s = br_mix(S->seed + 0x51 + 131ull * (uint64_t)k + 977ull * (uint64_t)blk) & S->mask;
  uint64_t x = ((uint64_t)r & S->mask) ^ s;
  x = (x * S->a1[k]) & S->mask; x ^= x >> S->h;
  x = (x * S->a2[k]) & S->mask; x ^= x >> S->h;
  x ^= 1;
  x ^= x >> S->h; x = (x * S->a2i[k]) & S->mask;
  x ^= x >> S->h; x = (x * S->a1i[k]) & S->mask;
  return (blk << S->B) | (int64_t)(x ^ s);
}
/* off-diagonal entries of row r into (c, v), unsorted; returns their number (cap >= 2 hb + K + 3000) */
static int br_row(const br_spec *S, int64_t r, int64_t *c, double *v) {
  int cnt = 0;
  for (int d = 1; d <= S->hb; d++) {
    if (r - d >= 0 && (br_key(S, r - d, r) & 7) != 0) {
      c[cnt] = r - d; v[cnt] = -(1.0 + (double)((br_key(S, r - d, r) >> 8) & 255) / 256.0); cnt++;
    }
    if (r + d < S->n && (br_key(S, r, r + d) & 7) != 0) {
      double m = 1.0 + (double)((br_key(S, r, r + d) >> 8) & 255) / 256.0;
      c[cnt] = r + d; v[cnt] = S->unsym ? -0.5 * m : -m; cnt++;
    }
  }
  int first_link = cnt;
  for (int k = 0; k < S->K; k++) {
    int64_t p = br_partner(S, k, r);
    if (p < 0 || p >= S->n) continue;
    if (llabs((long long)(p - r)) <= S->hb) continue;
    int dup = 0;
    for (int q = first_link; q < cnt; q++) if (c[q] == p) dup = 1;
    if (dup) continue;
    int64_t lo = p < r ? p : r, hi = p < r ? r : p;
    double m = 1.0 + (double)((br_key(S, lo, hi) >> 8) & 255) / 256.0;
    c[cnt] = p; v[cnt] = (S->unsym && p > r) ? -0.5 * m : -m; cnt++;
  }
  if (S->unsym) {
    for (int q = 0; q < S->ndense; q++) {
      if (r != (int64_t)(q + 1) * S->n / (S->ndense + 1)) continue;
      int base = cnt;
      for (int t = 0; t < 3000; t++) {
        int64_t cc = (r + 1 + (int64_t)t * S->stride) % S->n;
        if (cc == r) continue;
        int have = 0;
        for (int u = 0; u < cnt && !have; u++) have = c[u] == cc;      /* band, links and earlier extras */
        if (have) continue;
        int64_t lo = cc < r ? cc : r, hi = cc < r ? r : cc;
        c[cnt] = cc; v[cnt] = -(1.0 + (double)((br_key(S, lo, hi) >> 8) & 255) / 256.0) / 64.0; cnt++;
      }
      (void)base;
    }
  }
  return cnt;
}
int ko_csr_banded_random(int64_t n, int half_band, int links, uint64_t seed, int unsym, int dense_rows, ko_csr *A) {
  if (n < 2 || half_band < 0 || half_band > 64 || links < 0 || links > 16 || dense_rows < 0 || dense_rows > 64) return -2;
  br_spec S;
  S.n = n; S.hb = half_band; S.K = links; S.seed = seed; S.unsym = unsym != 0; S.ndense = dense_rows;
  int B = 0;
  while (((int64_t)2 << B) <= n) B++;
  S.B = B < 20 ? B : 20;
  S.mask = ((uint64_t)1 << S.B) - 1;
  S.h = (S.B + 1) / 2; if (S.h < 1) S.h = 1;
  for (int k = 0; k < links; k++) {
    S.a1[k] = (br_mix(seed * 4 + 4ull * (uint64_t)k + 1) | 1) & S.mask;
    S.a2[k] = (br_mix(seed * 4 + 4ull * (uint64_t)k + 2) | 1) & S.mask;
    S.a1i[k] = br_inverse(S.a1[k], S.mask);
    S.a2i[k] = br_inverse(S.a2[k], S.mask);
  }
  S.stride = (n / 3001) | 1;
  const int cap = 2 * half_band + links + 3000 + 8;
  A->n = n; A->nnz = 0; A->col = NULL; A->val = NULL;
  A->rowptr = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
  if (!A->rowptr) return -1;
  int fail = 0;
#pragma omp parallel
  {
    int64_t *c = (int64_t *)malloc(sizeof(int64_t) * (size_t)cap);
    double *v = (double *)malloc(sizeof(double) * (size_t)cap);
    if (!c || !v) {
#pragma omp atomic write
      fail = 1;
    } else {
#pragma omp for schedule(static)
      for (int64_t r = 0; r < n; r++) A->rowptr[r + 1] = br_row(&S, r, c, v) + 1;
    }
    free(c); free(v);
  }
  if (fail) { ko_csr_free(A); return -1; }
  A->rowptr[0] = 0;
  for (int64_t r = 0; r < n; r++) A->rowptr[r + 1] += A->rowptr[r];
  A->nnz = A->rowptr[n];
  A->col = (int32_t *)malloc(sizeof(int32_t) * (size_t)A->nnz);
  A->val = (double *)malloc(sizeof(double) * (size_t)A->nnz);
  if (!A->col || !A->val) { ko_csr_free(A); return -1; }
#pragma omp parallel
  {
    int64_t *c = (int64_t *)malloc(sizeof(int64_t) * (size_t)cap);
    double *v = (double *)malloc(sizeof(double) * (size_t)cap);
    if (c && v) {
#pragma omp for schedule(static)
      for (int64_t r = 0; r < n; r++) {
        int cnt = br_row(&S, r, c, v);
        double diag = 0.0625;
        for (int q = 0; q < cnt; q++) diag += fabs(v[q]);
        c[cnt] = r; v[cnt] = diag; cnt++;
        for (int q = 1; q < cnt; q++) {                 /* insertion sort by column */
          int64_t cq = c[q]; double vq = v[q]; int u = q - 1;
          while (u >= 0 && c[u] > cq) { c[u + 1] = c[u]; v[u + 1] = v[u]; u--; }
          c[u + 1] = cq; v[u + 1] = vq;
        }
        int64_t k = A->rowptr[r];
        for (int q = 0; q < cnt; q++) { A->col[k + q] = (int32_t)c[q]; A->val[k + q] = v[q]; }
      }
    } else {
#pragma omp atomic write
      fail = 1;
    }
    free(c); free(v);
  }
  if (fail) { ko_csr_free(A); return -1; }
  return 0;
}

int ko_csr_tridiag(int n, double lo, double di, double up, ko_csr *A) {
  int64_t nnz = n <= 0 ? 0 : 3 * (int64_t)n - 2;
  if (csr_alloc(A, n, nnz)) return -1;
  int64_t k = 0;
  for (int i = 0; i < n; i++) {
    A->rowptr[i] = k;
    if (i > 0) { A->col[k] = i - 1; A->val[k] = lo; k++; }
    A->col[k] = i; A->val[k] = di; k++;
    if (i < n - 1) { A->col[k] = i + 1; A->val[k] = up; k++; }
  }
  A->rowptr[n] = k;
  return 0;
}

int ko_csr_row_slice(const ko_csr *A, int64_t r0, int64_t r1, ko_csr *out) {
  if (r0 < 0 || r1 > A->n || r0 > r1) return -1;
  int64_t s = A->rowptr[r0], e = A->rowptr[r1];
  if (csr_alloc(out, r1 - r0, e - s)) return -1;
  for (int64_t i = r0; i <= r1; i++) out->rowptr[i - r0] = A->rowptr[i] - s;
  memcpy(out->col, A->col + s, sizeof(int32_t) * (size_t)(e - s));
  memcpy(out->val, A->val + s, sizeof(double) * (size_t)(e - s));
  return 0;
}

/* ===================================================================== *
 *  SpMV / SpMM
 * ===================================================================== */

void ko_spmv(const ko_csr *A, const double *x, double *y) {
  for (int64_t i = 0; i < A->n; i++) {
    double acc = 0.0;
    for (int64_t k = A->rowptr[i]; k < A->rowptr[i + 1]; k++) {
      double prod = A->val[k] * x[A->col[k]];
      acc = acc + prod;
    }
    y[i] = acc;
  }
}

/* ILU(0), IKJ variant (Saad alg. 10.4) on the pattern of A; see krylov_oracle.h */
int ko_ilu0(const ko_csr *A, double *lu, int64_t *diag) {
  const int64_t n = A->n;
  memcpy(lu, A->val, sizeof(double) * (size_t)A->nnz);
  for (int64_t i = 0; i < n; ++i) {
    diag[i] = -1;
    for (int64_t q = A->rowptr[i]; q < A->rowptr[i + 1]; ++q)
      if (A->col[q] == i) { diag[i] = q; break; }
    if (diag[i] < 0) return -(int)(i + 1);
  }
  for (int64_t i = 0; i < n; ++i) {
    const int64_t rb = A->rowptr[i], re = A->rowptr[i + 1];
    for (int64_t kk = rb; kk < diag[i]; ++kk) {
      const int64_t k = A->col[kk];
      const double piv = lu[diag[k]];
      if (piv == 0.0) return -(int)(k + 1);
      const double lik = lu[kk] / piv;
      lu[kk] = lik;
      int64_t p = kk + 1;
      for (int64_t q = diag[k] + 1; q < A->rowptr[k + 1]; ++q) {
        const int32_t j = A->col[q];
        while (p < re && A->col[p] < j) ++p;
        if (p < re && A->col[p] == j) {
          const double t = lik * lu[q];
          lu[p] = lu[p] - t;
        }
      }
    }
    if (lu[diag[i]] == 0.0) return -(int)(i + 1);
  }
  return 0;
}

void ko_ilu0_solve(const ko_csr *A, const double *lu, const int64_t *diag, const double *x, double *y) {
  const int64_t n = A->n;
  for (int64_t i = 0; i < n; ++i) {            /* L z = x, unit lower; z overwrites y */
    double acc = x[i];
    for (int64_t q = A->rowptr[i]; q < diag[i]; ++q) {
      const double t = lu[q] * y[A->col[q]];
      acc = acc - t;
    }
    y[i] = acc;
  }
  for (int64_t i = n - 1; i >= 0; --i) {       /* U y = z */
    double acc = y[i];
    for (int64_t q = diag[i] + 1; q < A->rowptr[i + 1]; ++q) {
      const double t = lu[q] * y[A->col[q]];
      acc = acc - t;
    }
    y[i] = acc / lu[diag[i]];
  }
}

void ko_spmv_omp(const ko_csr *A, const double *x, double *y) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < A->n; i++) {
    double acc = 0.0;
    for (int64_t k = A->rowptr[i]; k < A->rowptr[i + 1]; k++) {
      double prod = A->val[k] * x[A->col[k]];
      acc = acc + prod;
    }
    y[i] = acc;
  }
}

/* matrix-free get_div_grad (test/get_div_grad.jl:8-25): see krylov_oracle.h.  One row = the loop body of ko_spmv
 * over the entries csr_stencil3d would have stored for it: offsets in ascending column order, value 6 on the
 * diagonal and -1 on the six face neighbours, absent across a Dirichlet face. */
static inline double stencil7_row(const double *x, int64_t row, int i1, int i2, int i3, int n1, int n2, int n3) {
  const int64_t s2 = n1, s3 = (int64_t)n1 * n2;
  double acc = 0.0, prod;
  if (i3 > 0)      { prod = -1.0 * x[row - s3]; acc = acc + prod; }
  if (i2 > 0)      { prod = -1.0 * x[row - s2]; acc = acc + prod; }
  if (i1 > 0)      { prod = -1.0 * x[row - 1];  acc = acc + prod; }
                   { prod =  6.0 * x[row];      acc = acc + prod; }
  if (i1 < n1 - 1) { prod = -1.0 * x[row + 1];  acc = acc + prod; }
  if (i2 < n2 - 1) { prod = -1.0 * x[row + s2]; acc = acc + prod; }
  if (i3 < n3 - 1) { prod = -1.0 * x[row + s3]; acc = acc + prod; }
  return acc;
}
void ko_stencil7_matvec(const double *x, double *y, void *ud) {
  const ko_stencil7 *S = (const ko_stencil7 *)ud;
  const int n1 = S->n1, n2 = S->n2, n3 = S->n3;
  for (int i3 = 0; i3 < n3; i3++)
    for (int i2 = 0; i2 < n2; i2++) {
      const int64_t base = (int64_t)n1 * i2 + (int64_t)n1 * n2 * i3;
      for (int i1 = 0; i1 < n1; i1++) y[base + i1] = stencil7_row(x, base + i1, i1, i2, i3, n1, n2, n3);
    }
}
void ko_stencil7_matvec_omp(const double *x, double *y, void *ud) {
  const ko_stencil7 *S = (const ko_stencil7 *)ud;
  const int n1 = S->n1, n2 = S->n2, n3 = S->n3;
#pragma omp parallel for schedule(static) collapse(2)
  for (int i3 = 0; i3 < n3; i3++)
    for (int i2 = 0; i2 < n2; i2++) {
      const int64_t base = (int64_t)n1 * i2 + (int64_t)n1 * n2 * i3;
      for (int i1 = 0; i1 < n1; i1++) y[base + i1] = stencil7_row(x, base + i1, i1, i2, i3, n1, n2, n3);
    }
}

void ko_spmm(const ko_csr *A, const double *X, double *Y, int p) {
  int64_t n = A->n;
  for (int j = 0; j < p; j++) ko_spmv(A, X + (size_t)j * n, Y + (size_t)j * n);
}

void ko_csr_matvec(const double *x, double *y, void *csr) { ko_spmv((const ko_csr *)csr, x, y); }
void ko_csr_matvec_omp(const double *x, double *y, void *csr) { ko_spmv_omp((const ko_csr *)csr, x, y); }
void ko_csr_block_matvec(const double *X, double *Y, int p, void *csr) {
  ko_spmm((const ko_csr *)csr, X, Y, p);
}
/* same values (rows are independent), rows spread over the OpenMP threads: used for the full-size goldens */
void ko_csr_block_matvec_omp(const double *X, double *Y, int p, void *csr) {
  const ko_csr *A = (const ko_csr *)csr;
  for (int j = 0; j < p; j++) ko_spmv_omp(A, X + (size_t)j * A->n, Y + (size_t)j * A->n);
}

/* ===================================================================== *
 *  BLAS-1 shim
 * ===================================================================== */

/* ko_set_dot_mode(1): every dot / norm of the solvers becomes Dot2 (Ogita, Rump & Oishi 2005: TwoProd by one fma, TwoSum,
 * double-double accumulator; thread partials merged in thread order) -- as accurate as an accumulation in twice the working
 * precision, whatever n.  NOT the documented convention (mode 0, the default: sequential x87 extended precision, whose own
 * rounding reaches ~sqrt(n) 2^-64 -- 2e-15 at n = 2^30): it exists to MEASURE how far that convention is from exact dots
 * at sizes where binary128 (oracle/quad_reference.c) is out of reach (tests/golden/make_scale_golden.py leg 40). */
static int g_dot_mode = 0;
void ko_set_dot_mode(int mode) { g_dot_mode = mode; }
int ko_get_dot_mode(void) { return g_dot_mode; }

static inline void ko_two_sum(double a, double b, double *s, double *e) {
  *s = a + b;
  const double z = *s - a;
  *e = (a - (*s - z)) + (b - z);
}
static double ko_dot2(int64_t n, const double *x, const double *y) {
  int nt = g_threads > 1 && n >= ((int64_t)1 << 16) ? g_threads : 1;
  if (nt > 256) nt = 256;
  double hi[256], lo[256];
#pragma omp parallel for schedule(static) num_threads(nt)
  for (int t = 0; t < nt; t++) {
    const int64_t i0 = n * t / nt, i1 = n * (t + 1) / nt;
    double h = 0.0, l = 0.0;
    for (int64_t i = i0; i < i1; i++) {
      const double p = x[i] * y[i];
      const double pe = fma(x[i], y[i], -p);
      double s, e;
      ko_two_sum(h, p, &s, &e);
      h = s;
      l += e + pe;
    }
    hi[t] = h; lo[t] = l;
  }
  double h = 0.0, l = 0.0;
  for (int t = 0; t < nt; t++) {
    double s, e;
    ko_two_sum(h, hi[t], &s, &e);
    h = s;
    l += e + lo[t];
  }
  return h + l;
}

double ko_dot(int64_t n, const double *x, const double *y) {
  if (g_dot_mode == 1) return ko_dot2(n, x, y);
  long double acc = 0.0L;
  for (int64_t i = 0; i < n; i++) acc += (long double)x[i] * (long double)y[i];
  return (double)acc;
}

double ko_dot_omp(int64_t n, const double *x, const double *y) {
  long double acc = 0.0L;
#pragma omp parallel for schedule(static) reduction(+ : acc)
  for (int64_t i = 0; i < n; i++) acc += (long double)x[i] * (long double)y[i];
  return (double)acc;
}

double ko_nrm2(int64_t n, const double *x) { return sqrt(ko_dot(n, x, x)); }

void ko_scal(int64_t n, double s, double *x) {
  for (int64_t i = 0; i < n; i++) x[i] = s * x[i];
}
void ko_div(int64_t n, double *x, double s) { ko_scal(n, 1.0 / s, x); }
void ko_copy(int64_t n, double *y, const double *x) {
  if (y != x) memmove(y, x, sizeof(double) * (size_t)n);
}
void ko_scalcopy(int64_t n, double *y, double s, const double *x) {
  for (int64_t i = 0; i < n; i++) y[i] = s * x[i];
}
void ko_divcopy(int64_t n, double *y, const double *x, double s) {
  for (int64_t i = 0; i < n; i++) y[i] = x[i] / s;
}
/* The elementwise loops below run on g_threads threads once the vector is long (ko_set_threads; default 1 = the
 * reference's serial BLAS calls): elements are independent, so no value depends on the thread count. */
#define KO_PAR_MIN ((int64_t)1 << 22)
void ko_axpy(int64_t n, double s, const double *x, double *y) {
#pragma omp parallel for schedule(static) if (g_threads > 1 && n >= KO_PAR_MIN)
  for (int64_t i = 0; i < n; i++) y[i] = fma(s, x[i], y[i]);
}
void ko_axpy_omp(int64_t n, double s, const double *x, double *y) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) y[i] = fma(s, x[i], y[i]);
}
void ko_axpby(int64_t n, double s, const double *x, double t, double *y) {
#pragma omp parallel for schedule(static) if (g_threads > 1 && n >= KO_PAR_MIN)
  for (int64_t i = 0; i < n; i++) y[i] = fma(s, x[i], t * y[i]);
}
void ko_axpby_omp(int64_t n, double s, const double *x, double t, double *y) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) y[i] = fma(s, x[i], t * y[i]);
}
void ko_fill(int64_t n, double *x, double val) {
#pragma omp parallel for schedule(static) if (g_threads > 1 && n >= KO_PAR_MIN)
  for (int64_t i = 0; i < n; i++) x[i] = val;
}
/* reflect!(x, y, c, s): x_i <- c x_i + s y_i ; y_i <- s x_i - c y_i  (conj(s) = s for reals) */
void ko_ref(int64_t n, double *x, double *y, double c, double s) {
  for (int64_t i = 0; i < n; i++) {
    double xi = x[i], yi = y[i];
    x[i] = c * xi + s * yi;
    y[i] = s * xi - c * yi;
  }
}

/* ===================================================================== *
 *  Scalar helpers
 * ===================================================================== */

static double jl_sign(double a) { return a > 0 ? 1.0 : (a < 0 ? -1.0 : 0.0); }

/* src/krylov_utils.jl:21-51 */
void ko_sym_givens(double a, double b, double *c, double *s, double *rho) {
  if (b == 0.0) {
    *c = jl_sign(a) + (a == 0.0 ? 1.0 : 0.0);
    *s = 0.0;
    *rho = fabs(a);
  } else if (a == 0.0) {
    *c = 0.0;
    *s = jl_sign(b);
    *rho = fabs(b);
  } else if (fabs(b) > fabs(a)) {
    double t = a / b;
    *s = jl_sign(b) / sqrt(1.0 + t * t);
    *c = *s * t;
    *rho = b / *s;
  } else {
    double t = b / a;
    *c = jl_sign(a) / sqrt(1.0 + t * t);
    *s = *c * t;
    *rho = a / *c;
  }
}

/* src/krylov_utils.jl:110-152 ; returns -1 for "doesn't have real roots" */
int ko_roots_quadratic(double q2, double q1, double q0, int nitref, double *r1, double *r2) {
  double root1, root2;
  if (q2 == 0.0) {
    double root;
    if (q1 == 0.0) {
      if (q0 != 0.0) return -1;
      root = 0.0;
    } else {
      root = -q0 / q1;
    }
    *r1 = root; *r2 = root;
    return 0;
  }
  double rhs = sqrt(2.220446049250313e-16) * q1 * q1;
  if (fabs(q0 * q2) > rhs) {
    double rho = q1 * q1 - 4 * q2 * q0;
    if (rho < 0) return -1;
    double d = -(q1 + copysign(sqrt(rho), q1)) / 2;
    root1 = d / q2;
    root2 = q0 / d;
  } else {
    root1 = -q1 / q2;
    root2 = 0.0;
  }
  for (int it = 0; it < nitref; it++) {
    double q = (q2 * root1 + q1) * root1 + q0;
    double dq = 2 * q2 * root1 + q1;
    if (dq == 0.0) continue;
    root1 = root1 - q / dq;
  }
  for (int it = 0; it < nitref; it++) {
    double q = (q2 * root2 + q1) * root2 + q0;
    double dq = 2 * q2 * root2 + q1;
    if (dq == 0.0) continue;
    root2 = root2 - q / dq;
  }
  *r1 = root1; *r2 = root2;
  return 0;
}

/* src/krylov_utils.jl:375-402 with M === I */
int ko_to_boundary(int64_t n, const double *x, const double *d, double radius, int flip,
                   double xNorm2, double dNorm2, double *s1, double *s2) {
  if (!(radius > 0)) return -1;
  double rxd = ko_dot(n, x, d);
  if (dNorm2 == 0.0) dNorm2 = ko_dot(n, d, d);
  if (xNorm2 == 0.0) xNorm2 = ko_dot(n, x, x);
  if (dNorm2 == 0.0) return -2;
  if (flip) rxd = -rxd;
  double radius2 = radius * radius;
  if (!(xNorm2 <= radius2)) return -3;
  return ko_roots_quadratic(dNorm2, 2 * rxd, xNorm2 - radius2, 1, s1, s2);
}

/* ===================================================================== *
 *  options / stats
 * ===================================================================== */

ko_options ko_default_options(void) {
  ko_options o;
  memset(&o, 0, sizeof(o));
  o.atol = NAN; o.rtol = NAN; o.timemax = NAN;
  return o;
}

void ko_stats_init(ko_stats *s) {
  memset(s, 0, sizeof(*s));
  strcpy(s->status, "unknown");
}
void ko_stats_free(ko_stats *s) {
  free(s->residuals);
  s->residuals = NULL; s->nres = s->cap = 0;
}
static void stats_reset(ko_stats *s) { /* reset!(stats) src/krylov_stats.jl:38-44 */
  s->nres = 0; s->indefinite = 0; s->npcCount = 0; s->error[0] = 0;
}
static void stats_push(ko_stats *s, double v) {
  if (s->nres == s->cap) {
    s->cap = s->cap ? 2 * s->cap : 64;
    s->residuals = (double *)realloc(s->residuals, sizeof(double) * (size_t)s->cap);
  }
  s->residuals[s->nres++] = v;
}
static int fail(ko_stats *s, const char *msg) {
  snprintf(s->error, sizeof(s->error), "%s", msg);
  return -1;
}

static const double EPS = 2.220446049250313e-16;
static double tol_or_default(double t) { return isnan(t) ? sqrt(EPS) : t; }
static double timemax_of(const ko_options *o) {
  return (isnan(o->timemax) || o->timemax <= 0) ? INFINITY : o->timemax;
}
static double *vec_alloc(int64_t n) { return (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double)); }

/* ===================================================================== *
 *  CG   (src/cg.jl:120-291)
 * ===================================================================== */

ko_cg_workspace *ko_cg_workspace_create(int64_t m, int64_t n) { /* src/krylov_workspaces.jl:269-285 */
  ko_cg_workspace *ws = (ko_cg_workspace *)calloc(1, sizeof(*ws));
  ws->m = m; ws->n = n;
  ws->x = vec_alloc(n); ws->r = vec_alloc(n); ws->p = vec_alloc(n); ws->Ap = vec_alloc(n);
  ko_stats_init(&ws->stats);
  return ws;
}
void ko_cg_workspace_free(ko_cg_workspace *ws) {
  if (!ws) return;
  free(ws->dx); free(ws->x); free(ws->r); free(ws->npc_dir); free(ws->p); free(ws->Ap); free(ws->z);
  ko_stats_free(&ws->stats);
  free(ws);
}
void ko_cg_warm_start(ko_cg_workspace *ws, const double *x0) {
  if (!ws->dx) ws->dx = vec_alloc(ws->n);
  ko_copy(ws->n, ws->dx, x0);
  ws->warm_start = 1;
}

int ko_cg(ko_cg_workspace *ws, ko_matvec A, ko_matvec M, void *ud, const double *b,
          const ko_options *opts_in) {
  ko_options o = opts_in ? *opts_in : ko_default_options();
  double t0 = now_s();
  double timemax = timemax_of(&o);
  int64_t n = ws->n;
  ko_stats *st = &ws->stats;
  double atol = tol_or_default(o.atol), rtol = tol_or_default(o.rtol);
  double radius = o.radius;
  int linesearch = o.linesearch;

  if (ws->m != ws->n) return fail(st, "System must be square");
  if (linesearch && radius > 0) return fail(st, "`linesearch` set to `true` but trust-region radius > 0");
  if (ws->warm_start && linesearch) return fail(st, "warm_start and linesearch cannot be used together");
  int MisI = (M == NULL);
  if (!MisI && radius > 0) return fail(st, "oracle: radius > 0 with a preconditioner is not restated");

  if (!MisI && !ws->z) ws->z = vec_alloc(n);                         /* :142 */
  if ((linesearch || radius > 0) && !ws->npc_dir) ws->npc_dir = vec_alloc(n); /* :143 */
  double *dx = ws->dx, *x = ws->x, *r = ws->r, *p = ws->p, *Ap = ws->Ap;
  int warm_start = ws->warm_start;
  stats_reset(st);
  double *z = MisI ? r : ws->z;
  double *npc_dir = ws->npc_dir;

  ko_fill(n, x, 0.0);                                                 /* :153 */
  if (warm_start) {
    A(dx, r, ud);
    ko_axpby(n, 1.0, b, -1.0, r);
  } else {
    ko_copy(n, r, b);
  }
  if (!MisI) M(r, z, ud);
  ko_copy(n, p, z);
  double gamma = ko_dot(n, r, z);
  if (!(gamma >= 0))
    return fail(st, "The linear operator `A` or the preconditioner `M` is not symmetric positive definite.");
  double rNorm = sqrt(gamma);
  if (o.history) stats_push(st, rNorm);
  if (gamma == 0) {                                                   /* :166-174 */
    st->niter = 0; st->solved = 1; st->inconsistent = 0;
    st->timer = now_s() - t0;
    strcpy(st->status, "x is a zero-residual solution");
    if (warm_start) ko_axpy(n, 1.0, dx, x);
    ws->warm_start = 0;
    return 0;
  }

  int64_t iter = 0;
  int64_t itmax = o.itmax == 0 ? 2 * n : o.itmax;
  double pAp = 0.0;
  double pNorm2 = gamma;
  double eps_tol = atol + rtol * rNorm;

  int solved = rNorm <= eps_tol;
  int tired = iter >= itmax;
  int inconsistent = 0, on_boundary = 0, zero_curvature = 0, user_requested_exit = 0, overtimed = 0;
  const char *status = "unknown";

  while (!(solved || tired || zero_curvature || user_requested_exit || overtimed)) {
    A(p, Ap, ud);                                                     /* :196 */
    pAp = ko_dot(n, p, Ap);                                           /* :197 */
    if ((pAp <= EPS * pNorm2) && (radius == 0)) {                     /* :198-211 */
      if (fabs(pAp) <= EPS * pNorm2) {
        zero_curvature = 1;
        inconsistent = !linesearch;
      }
      if (linesearch) {
        if (iter == 0) ko_copy(n, x, p);
        ko_copy(n, npc_dir, p);
        st->npcCount = 1;
        st->indefinite = 1;
        solved = 1;
      }
    }
    if (zero_curvature || solved) continue;

    double alpha = gamma / pAp;                                       /* :213 */
    double sigma;
    if (radius == 0) {
      sigma = alpha;
    } else {
      double s1, s2;
      int rc = ko_to_boundary(n, x, p, radius, 0, 0.0, pNorm2, &s1, &s2);
      if (rc) return fail(st, "to_boundary failed");
      sigma = s1 > s2 ? s1 : s2;
    }
    if ((radius > 0) && ((pAp <= 0) || (alpha > sigma))) {            /* :229-237 */
      alpha = sigma;
      if (pAp <= 0) {
        ko_copy(n, npc_dir, p);
        st->npcCount = 1;
        st->indefinite = 1;
      }
      on_boundary = 1;
    }

    ko_axpy(n, alpha, p, x);                                          /* :239 */
    ko_axpy(n, -alpha, Ap, r);                                        /* :240 */
    if (!MisI) M(r, z, ud);
    double gamma_next = ko_dot(n, r, z);                              /* :242 */
    if (!(gamma_next >= 0))
      return fail(st, "The linear operator `A` or the preconditioner `M` is not symmetric positive definite.");
    rNorm = sqrt(gamma_next);
    if (o.history) stats_push(st, rNorm);

    int resid_decrease_mach = (rNorm + 1.0 <= 1.0);
    int resid_decrease_lim = rNorm <= eps_tol;
    int resid_decrease = resid_decrease_lim || resid_decrease_mach;
    solved = resid_decrease || on_boundary;

    if (!solved) {                                                    /* :255-260 */
      double beta = gamma_next / gamma;
      pNorm2 = gamma_next + beta * beta * pNorm2;
      gamma = gamma_next;
      ko_axpby(n, 1.0, z, beta, p);
    }

    iter = iter + 1;
    tired = iter >= itmax;
    if (o.callback) user_requested_exit = o.callback(ws, o.callback_data) != 0;
    overtimed = (now_s() - t0) > timemax;
  }

  if (solved && on_boundary) status = "on trust-region boundary";
  if (solved && st->indefinite) status = "nonpositive curvature";
  if (solved && strcmp(status, "unknown") == 0) status = "solution good enough given atol and rtol";
  if (zero_curvature) status = "zero curvature detected";
  if (tired) status = "maximum number of iterations exceeded";
  if (user_requested_exit) status = "user-requested exit";
  if (overtimed) status = "time limit exceeded";

  if (warm_start) ko_axpy(n, 1.0, dx, x);
  ws->warm_start = 0;

  st->niter = (int)iter;
  st->solved = solved;
  st->inconsistent = inconsistent;
  st->timer = now_s() - t0;
  snprintf(st->status, sizeof(st->status), "%s", status);
  return 0;
}

/* cfg-4 oracle: ko_cg itself on the matrix-free operator, b = ones (krylov_oracle.h) */
int ko_cg_stencil7(int n1, int n2, int n3, const ko_options *opts, ko_stats *st,
                   int nsample, const int64_t *x_idx, double *x_out) {
  const int64_t n = (int64_t)n1 * n2 * n3;
  ko_stencil7 S = {n1, n2, n3};
  ko_cg_workspace *ws = ko_cg_workspace_create(n, n);
  double *b = (double *)malloc(sizeof(double) * (size_t)n);
  if (!ws || !b) { free(b); if (ws) ko_cg_workspace_free(ws); return -2; }
  ko_fill(n, b, 1.0);
  int rc = ko_cg(ws, g_threads > 1 ? ko_stencil7_matvec_omp : ko_stencil7_matvec, NULL, &S, b, opts);
  for (int i = 0; i < nsample; i++) x_out[i] = ws->x[x_idx[i]];
  if (st) {            /* hand the statistics (history included) to the caller */
    ko_stats_free(st);
    *st = ws->stats;
    ws->stats.residuals = NULL; ws->stats.nres = ws->stats.cap = 0;
  }
  free(b);
  ko_cg_workspace_free(ws);
  return rc;
}

/* ===================================================================== *
 *  GMRES   (src/gmres.jl:121-384)
 * ===================================================================== */

ko_gmres_workspace *ko_gmres_workspace_create(int64_t m, int64_t n, int memory) {
  /* src/krylov_workspaces.jl:2898-2918 */
  ko_gmres_workspace *ws = (ko_gmres_workspace *)calloc(1, sizeof(*ws));
  if (memory <= 0) memory = 20;
  if ((int64_t)memory > m) memory = (int)m;
  ws->m = m; ws->n = n; ws->mem = memory; ws->nV = memory;
  ws->x = vec_alloc(n); ws->w = vec_alloc(n);
  ws->V = (double **)calloc((size_t)(memory > 0 ? memory : 1), sizeof(double *));
  for (int i = 0; i < memory; i++) ws->V[i] = vec_alloc(n);
  ws->capcs = memory; ws->capz = memory; ws->capR = memory * (memory + 1) / 2;
  ws->c = vec_alloc(ws->capcs); ws->s = vec_alloc(ws->capcs);
  ws->z = vec_alloc(ws->capz); ws->R = vec_alloc(ws->capR);
  ko_stats_init(&ws->stats);
  return ws;
}
void ko_gmres_workspace_free(ko_gmres_workspace *ws) {
  if (!ws) return;
  free(ws->dx); free(ws->x); free(ws->w); free(ws->p); free(ws->q);
  for (int i = 0; i < ws->nV; i++) free(ws->V[i]);
  free(ws->V); free(ws->c); free(ws->s); free(ws->z); free(ws->R);
  ko_stats_free(&ws->stats);
  free(ws);
}
void ko_gmres_warm_start(ko_gmres_workspace *ws, const double *x0) {
  if (!ws->dx) ws->dx = vec_alloc(ws->n);
  ko_copy(ws->n, ws->dx, x0);
  ws->warm_start = 1;
}

static void grow(double **a, int *cap, int need) {
  if (need <= *cap) return;
  int nc = *cap;
  while (nc < need) nc = nc ? 2 * nc : 8;
  *a = (double *)realloc(*a, sizeof(double) * (size_t)nc);
  for (int i = *cap; i < nc; i++) (*a)[i] = 0.0;
  *cap = nc;
}

int ko_gmres(ko_gmres_workspace *ws, ko_matvec A, ko_matvec M, ko_matvec N, void *ud,
             const double *b, const ko_options *opts_in) {
  ko_options o = opts_in ? *opts_in : ko_default_options();
  double t0 = now_s();
  double timemax = timemax_of(&o);
  int64_t n = ws->n;
  ko_stats *st = &ws->stats;
  double atol = tol_or_default(o.atol), rtol = tol_or_default(o.rtol);
  int restart = o.restart, reorth = o.reorthogonalization;
  if (ws->m != ws->n) return fail(st, "System must be square");

  int MisI = (M == NULL), NisI = (N == NULL);
  if (!MisI && !ws->q) ws->q = vec_alloc(n);
  if (!NisI && !ws->p) ws->p = vec_alloc(n);
  if (restart && !ws->dx) ws->dx = vec_alloc(n);
  double *dx = ws->dx, *x = ws->x, *w = ws->w;
  int warm_start = ws->warm_start;
  stats_reset(st);
  double *q = MisI ? w : ws->q;
  double *r0 = MisI ? w : ws->q;
  double *xr = restart ? dx : x;

  ko_fill(n, x, 0.0);                                                 /* :155 */
  if (warm_start) {
    A(dx, w, ud);
    ko_axpby(n, 1.0, b, -1.0, w);
    if (restart) ko_axpy(n, 1.0, dx, x);
  } else {
    ko_copy(n, w, b);
  }
  if (!MisI) M(w, r0, ud);
  double beta = ko_nrm2(n, r0);
  double rNorm = beta;
  if (o.history) stats_push(st, beta);
  double eps_tol = atol + rtol * rNorm;

  if (beta == 0) {
    st->niter = 0; st->solved = 1; st->inconsistent = 0;
    st->timer = now_s() - t0;
    strcpy(st->status, "x is a zero-residual solution");
    if (warm_start) ko_axpy(n, 1.0, dx, x);
    ws->warm_start = 0;
    return 0;
  }

  int mem = ws->mem;                                                  /* length(c) */
  int npass = 0;
  int64_t iter = 0;
  int inner_iter = 0;
  int64_t itmax = o.itmax == 0 ? 2 * n : o.itmax;
  int64_t inner_itmax = itmax;
  double btol = pow(EPS, 0.75);

  int breakdown = 0, inconsistent = 0;
  int solved = rNorm <= eps_tol;
  int tired = iter >= itmax;
  int inner_tired = inner_iter >= inner_itmax;
  int user_requested_exit = 0, overtimed = 0;
  const char *status = "unknown";

  while (!(solved || tired || breakdown || user_requested_exit || overtimed)) {
    int nr = 0;
    for (int i = 0; i < mem; i++) ko_fill(n, ws->V[i], 0.0);          /* :211-213 */
    /* kfill!(s,0); kfill!(c,0); kfill!(R,0); kfill!(z,0) over their CURRENT lengths */
    for (int i = 0; i < ws->capcs; i++) { ws->s[i] = 0.0; ws->c[i] = 0.0; }
    for (int i = 0; i < ws->capR; i++) ws->R[i] = 0.0;
    for (int i = 0; i < ws->capz; i++) ws->z[i] = 0.0;

    if (restart) {
      ko_fill(n, xr, 0.0);
      if (npass >= 1) {
        A(x, w, ud);
        ko_axpby(n, 1.0, b, -1.0, w);
        if (!MisI) M(w, r0, ud);
      }
    }

    beta = ko_nrm2(n, r0);                                            /* :229 */
    ws->z[0] = beta;
    ko_divcopy(n, ws->V[0], r0, rNorm);                               /* :231 (divides by rNorm, not beta) */

    npass = npass + 1;
    ws->inner_iter = 0;
    inner_tired = 0;

    while (!(solved || inner_tired || breakdown || user_requested_exit || overtimed)) {
      ws->inner_iter = ws->inner_iter + 1;
      inner_iter = ws->inner_iter;

      if (!restart && (inner_iter > mem)) {                           /* :244-252 push!(R..), push!(s), push!(c) */
        grow(&ws->R, &ws->capR, nr + inner_iter);
        int cap_s = ws->capcs, cap_c = ws->capcs;
        grow(&ws->s, &cap_s, inner_iter);
        grow(&ws->c, &cap_c, inner_iter);
        ws->capcs = cap_s;
        if (inner_iter > ws->mem) ws->mem = inner_iter;               /* length(c) seen by the NEXT solve */
      }

      double *Vk = ws->V[inner_iter - 1];
      double *pp = NisI ? Vk : ws->p;
      if (!NisI) N(Vk, pp, ud);
      A(pp, w, ud);                                                   /* :257 */
      if (!MisI) M(w, q, ud);
      double *R = ws->R, *c = ws->c, *s = ws->s, *z = ws->z;
      for (int i = 0; i < inner_iter; i++) {                          /* :259-262 */
        R[nr + i] = ko_dot(n, ws->V[i], q);
        ko_axpy(n, -R[nr + i], ws->V[i], q);
      }
      if (reorth) {                                                   /* :265-271 */
        for (int i = 0; i < inner_iter; i++) {
          double Htmp = ko_dot(n, ws->V[i], q);
          R[nr + i] += Htmp;
          ko_axpy(n, -Htmp, ws->V[i], q);
        }
      }
      double Hbis = ko_nrm2(n, q);                                    /* :274 */

      for (int i = 0; i < inner_iter - 1; i++) {                      /* :280-284 */
        double Rtmp = c[i] * R[nr + i] + s[i] * R[nr + i + 1];
        R[nr + i + 1] = s[i] * R[nr + i] - c[i] * R[nr + i + 1];
        R[nr + i] = Rtmp;
      }
      int k = inner_iter - 1;
      ko_sym_givens(R[nr + k], Hbis, &c[k], &s[k], &R[nr + k]);       /* :289 */

      double zeta_next = s[k] * z[k];                                 /* :292-293 */
      z[k] = c[k] * z[k];

      rNorm = fabs(zeta_next);
      if (o.history) stats_push(st, rNorm);
      nr = nr + inner_iter;

      int resid_decrease_mach = (rNorm + 1.0 <= 1.0);
      if (o.callback) user_requested_exit = o.callback(ws, o.callback_data) != 0;
      int resid_decrease_lim = rNorm <= eps_tol;
      breakdown = Hbis <= btol;
      solved = resid_decrease_lim || resid_decrease_mach;
      if (restart) {
        int64_t lim = (int64_t)mem < inner_itmax ? (int64_t)mem : inner_itmax;
        inner_tired = inner_iter >= lim;
      } else {
        inner_tired = inner_iter >= inner_itmax;
      }
      overtimed = (now_s() - t0) > timemax;

      if (!(solved || inner_tired || breakdown || user_requested_exit || overtimed)) {
        if (!restart && (inner_iter >= mem)) {                        /* :319-324 */
          if (ws->nV <= inner_iter) {
            ws->V = (double **)realloc(ws->V, sizeof(double *) * (size_t)(inner_iter + 1));
            for (int i = ws->nV; i <= inner_iter; i++) ws->V[i] = vec_alloc(n);
            ws->nV = inner_iter + 1;
          }
          grow(&ws->z, &ws->capz, inner_iter + 1);
          z = ws->z;
        }
        ko_divcopy(n, ws->V[inner_iter], q, Hbis);                    /* :325 */
        z[inner_iter] = zeta_next;
      }
    }

    /* back-substitution R y = z (:331-345), y aliases z */
    double *R = ws->R, *y = ws->z;
    for (int i = inner_iter; i >= 1; i--) {
      int pos = nr + i - inner_iter;                 /* 1-based position of r_{i,k} */
      for (int j = inner_iter; j >= i + 1; j--) {
        y[i - 1] = y[i - 1] - R[pos - 1] * y[j - 1];
        pos = pos - j + 1;
      }
      if (fabs(R[pos - 1]) <= btol) {
        y[i - 1] = 0.0;
        inconsistent = 1;
      } else {
        y[i - 1] = y[i - 1] / R[pos - 1];
      }
    }

    for (int i = 0; i < inner_iter; i++) ko_axpy(n, y[i], ws->V[i], xr);  /* :348-350 */
    if (!NisI) {
      ko_copy(n, ws->p, xr);
      N(ws->p, xr, ud);
    }
    if (restart) ko_axpy(n, 1.0, xr, x);

    inner_itmax = inner_itmax - inner_iter;
    iter = iter + inner_iter;
    tired = iter >= itmax;
    overtimed = (now_s() - t0) > timemax;
  }

  if (tired) status = "maximum number of iterations exceeded";
  if (solved) status = "solution good enough given atol and rtol";
  if (inconsistent) status = "found approximate least-squares solution";
  if (user_requested_exit) status = "user-requested exit";
  if (overtimed) status = "time limit exceeded";

  if (warm_start && !restart) ko_axpy(n, 1.0, dx, x);
  ws->warm_start = 0;

  st->niter = (int)iter;
  st->solved = solved;
  st->inconsistent = inconsistent;
  st->timer = now_s() - t0;
  snprintf(st->status, sizeof(st->status), "%s", status);
  return 0;
}

/* ===================================================================== *
 *  BiCGSTAB   (src/bicgstab.jl:125-277)
 * ===================================================================== */

ko_bicgstab_workspace *ko_bicgstab_workspace_create(int64_t m, int64_t n) {
  /* src/krylov_workspaces.jl:1605-1623 */
  ko_bicgstab_workspace *ws = (ko_bicgstab_workspace *)calloc(1, sizeof(*ws));
  ws->m = m; ws->n = n;
  ws->x = vec_alloc(n); ws->r = vec_alloc(n); ws->p = vec_alloc(n);
  ws->v = vec_alloc(n); ws->s = vec_alloc(n); ws->qd = vec_alloc(n);
  ko_stats_init(&ws->stats);
  return ws;
}
void ko_bicgstab_workspace_free(ko_bicgstab_workspace *ws) {
  if (!ws) return;
  free(ws->dx); free(ws->x); free(ws->r); free(ws->p); free(ws->v); free(ws->s);
  free(ws->qd); free(ws->yz); free(ws->t);
  ko_stats_free(&ws->stats);
  free(ws);
}
void ko_bicgstab_warm_start(ko_bicgstab_workspace *ws, const double *x0) {
  if (!ws->dx) ws->dx = vec_alloc(ws->n);
  ko_copy(ws->n, ws->dx, x0);
  ws->warm_start = 1;
}

int ko_bicgstab(ko_bicgstab_workspace *ws, ko_matvec A, ko_matvec M, ko_matvec N, void *ud,
                const double *b, const double *c, const ko_options *opts_in) {
  ko_options o = opts_in ? *opts_in : ko_default_options();
  double t0 = now_s();
  double timemax = timemax_of(&o);
  int64_t n = ws->n;
  ko_stats *st = &ws->stats;
  double atol = tol_or_default(o.atol), rtol = tol_or_default(o.rtol);
  if (ws->m != ws->n) return fail(st, "System must be square");
  if (!c) c = b;                                                      /* :105 */

  int MisI = (M == NULL), NisI = (N == NULL);
  if (!MisI && !ws->t) ws->t = vec_alloc(n);
  if (!NisI && !ws->yz) ws->yz = vec_alloc(n);
  double *dx = ws->dx, *x = ws->x, *r = ws->r, *p = ws->p, *v = ws->v, *s = ws->s;
  int warm_start = ws->warm_start;
  stats_reset(st);
  double *q = ws->qd, *d = ws->qd;
  double *t = MisI ? d : ws->t;
  double *y = NisI ? p : ws->yz;
  double *z = NisI ? s : ws->yz;
  double *r0 = MisI ? r : ws->qd;

  if (warm_start) {
    A(dx, r0, ud);
    ko_axpby(n, 1.0, b, -1.0, r0);
  } else {
    ko_copy(n, r0, b);
  }
  ko_fill(n, x, 0.0);
  ko_fill(n, s, 0.0);
  ko_fill(n, v, 0.0);
  if (!MisI) M(r0, r, ud);
  ko_copy(n, p, r);

  double alpha = 1.0, omega = 1.0, rho = 1.0;
  double rNorm = ko_nrm2(n, r);
  if (o.history) stats_push(st, rNorm);
  if (rNorm == 0) {
    st->niter = 0; st->solved = 1; st->inconsistent = 0;
    st->timer = now_s() - t0;
    strcpy(st->status, "x is a zero-residual solution");
    if (warm_start) ko_axpy(n, 1.0, dx, x);
    ws->warm_start = 0;
    return 0;
  }

  int64_t iter = 0;
  int64_t itmax = o.itmax == 0 ? 2 * n : o.itmax;
  double eps_tol = atol + rtol * rNorm;

  double next_rho = ko_dot(n, c, r);                                  /* :196 */
  if (next_rho == 0) {
    st->niter = 0; st->solved = 0; st->inconsistent = 0;
    st->timer = now_s() - t0;
    strcpy(st->status, "Breakdown b\xe1\xb4\xb4""c = 0");               /* "Breakdown bᴴc = 0" */
    if (warm_start) ko_axpy(n, 1.0, dx, x);
    ws->warm_start = 0;
    return 0;
  }

  int solved = rNorm <= eps_tol;
  int tired = iter >= itmax;
  int breakdown = 0, user_requested_exit = 0, overtimed = 0;
  const char *status = "unknown";

  while (!(solved || tired || breakdown || user_requested_exit || overtimed)) {
    iter = iter + 1;
    rho = next_rho;

    if (!NisI) N(p, y, ud);
    A(y, q, ud);                                                      /* :221 */
    if (MisI) ko_copy(n, v, q); else M(q, v, ud);                     /* :222 (unguarded) */
    alpha = rho / ko_dot(n, c, v);                                    /* :223 */
    ko_copy(n, s, r);                                                 /* :224 */
    ko_axpy(n, -alpha, v, s);                                         /* :225 */
    ko_axpy(n, alpha, y, x);                                          /* :226 */
    if (!NisI) N(s, z, ud);
    A(z, d, ud);                                                      /* :228 */
    if (!MisI) M(d, t, ud);
    omega = ko_dot(n, t, s) / ko_dot(n, t, t);                        /* :230 */
    ko_axpy(n, omega, z, x);                                          /* :231 */
    ko_copy(n, r, s);                                                 /* :232 */
    ko_axpy(n, -omega, t, r);                                         /* :233 */
    next_rho = ko_dot(n, c, r);                                       /* :234 */
    double beta = (next_rho / rho) * (alpha / omega);                 /* :235 */
    ko_axpy(n, -omega, v, p);                                         /* :236 */
    ko_axpby(n, 1.0, r, beta, p);                                     /* :237 */

    rNorm = ko_nrm2(n, r);                                            /* :240 */
    if (o.history) stats_push(st, rNorm);

    int resid_decrease_mach = (rNorm + 1.0 <= 1.0);
    if (o.callback) user_requested_exit = o.callback(ws, o.callback_data) != 0;
    int resid_decrease_lim = rNorm <= eps_tol;
    solved = resid_decrease_lim || resid_decrease_mach;
    tired = iter >= itmax;
    breakdown = (alpha == 0 || isnan(alpha));
    overtimed = (now_s() - t0) > timemax;
  }

  if (tired) status = "maximum number of iterations exceeded";
  if (breakdown) status = "breakdown \xce\xb1\xe2\x82\x96 == 0";       /* "breakdown αₖ == 0" */
  if (solved) status = "solution good enough given atol and rtol";
  if (user_requested_exit) status = "user-requested exit";
  if (overtimed) status = "time limit exceeded";

  if (warm_start) ko_axpy(n, 1.0, dx, x);
  ws->warm_start = 0;

  st->niter = (int)iter;
  st->solved = solved;
  st->inconsistent = 0;
  st->timer = now_s() - t0;
  snprintf(st->status, sizeof(st->status), "%s", status);
  return 0;
}

/* ===================================================================== *
 *  Dense Householder kernels (LAPACK restatement, column-major)
 *  src/block_krylov_utils.jl:192-301 -> dgeqrf_/dorgqr_/dormqr_
 * ===================================================================== */

static double dlapy2(double x, double y) {
  double xa = fabs(x), ya = fabs(y);
  double w = xa > ya ? xa : ya, z = xa > ya ? ya : xa;
  if (z == 0.0) return w;
  double q = z / w;
  return w * sqrt(1.0 + q * q);
}

/* DLARFG: x has n-1 entries */
static void dlarfg(int n, double *alpha, double *x, double *tau) {
  if (n <= 1) { *tau = 0.0; return; }
  double xnorm = ko_nrm2(n - 1, x);
  if (xnorm == 0.0) { *tau = 0.0; return; }
  double beta = -copysign(dlapy2(*alpha, xnorm), *alpha);
  const double safmin = 2.2250738585072014e-308 / 1.1102230246251565e-16;
  const double rsafmn = 1.0 / safmin;
  int knt = 0;
  if (fabs(beta) < safmin) {
    do {
      knt++;
      for (int i = 0; i < n - 1; i++) x[i] *= rsafmn;
      beta *= rsafmn;
      *alpha *= rsafmn;
    } while (fabs(beta) < safmin && knt < 20);
    xnorm = ko_nrm2(n - 1, x);
    beta = -copysign(dlapy2(*alpha, xnorm), *alpha);
  }
  *tau = (beta - *alpha) / beta;
  double sc = 1.0 / (*alpha - beta);
  for (int i = 0; i < n - 1; i++) x[i] *= sc;
  for (int j = 0; j < knt; j++) beta *= safmin;
  *alpha = beta;
}

/* DLARF side='L': C(m x n) <- (I - tau v v^T) C, v has implicit v[0] = 1 */
static void dlarf_left(int m, int n, const double *v, double tau, double *C, int ldc) {
  if (tau == 0.0) return;
  /* columns of C are independent */
#pragma omp parallel for schedule(dynamic, 1) if (m > 100000)
  for (int j = 0; j < n; j++) {
    double *cj = C + (size_t)j * ldc;
    double w = cj[0];
    for (int i = 1; i < m; i++) w += v[i] * cj[i];
    double tw = tau * w;
    cj[0] -= tw;
    for (int i = 1; i < m; i++) cj[i] -= v[i] * tw;
  }
}

void ko_geqrf(int m, int n, double *A, int lda, double *tau) { /* DGEQR2 */
  int k = m < n ? m : n;
  for (int i = 0; i < k; i++) {
    double *aii = A + (size_t)i * lda + i;
    int rest = m - i;
    dlarfg(rest, aii, aii + (rest > 1 ? 1 : 0), &tau[i]);
    if (i < n - 1) dlarf_left(rest, n - i - 1, aii, tau[i], A + (size_t)(i + 1) * lda + i, lda);
  }
}

void ko_orgqr(int m, int n, int k, double *A, int lda, const double *tau) { /* DORG2R */
  for (int j = k; j < n; j++) {
    for (int l = 0; l < m; l++) A[(size_t)j * lda + l] = 0.0;
    A[(size_t)j * lda + j] = 1.0;
  }
  for (int i = k - 1; i >= 0; i--) {
    double *aii = A + (size_t)i * lda + i;
    if (i < n - 1) dlarf_left(m - i, n - i - 1, aii, tau[i], A + (size_t)(i + 1) * lda + i, lda);
    if (i < m - 1)
      for (int l = 1; l < m - i; l++) aii[l] *= -tau[i];
    aii[0] = 1.0 - tau[i];
    for (int l = 0; l < i; l++) A[(size_t)i * lda + l] = 0.0;
  }
}

void ko_ormqr_LT(int m, int n, int k, const double *A, int lda, const double *tau, double *C,
                 int ldc) { /* DORM2R, side L, trans T: apply H(1)...H(k) in forward order */
  for (int i = 0; i < k; i++)
    dlarf_left(m - i, n, A + (size_t)i * lda + i, tau[i], C + i, ldc);
}

/* householder!(Q, R, tau, buffer; compact) src/block_krylov_utils.jl:201-208 */
void ko_householder(int n, int k, double *Q, double *R, double *tau, int compact) {
  for (int i = 0; i < k * k; i++) R[i] = 0.0;
  ko_geqrf(n, k, Q, n, tau);
  for (int j = 0; j < k; j++)
    for (int i = 0; i <= j && i < n; i++) R[(size_t)j * k + i] = Q[(size_t)j * n + i];
  if (!compact) ko_orgqr(n, k, k, Q, n, tau);
}

/* ===================================================================== *
 *  block-GMRES   (src/block_gmres.jl:110-358)
 * ===================================================================== */

static double *mat_alloc(int64_t r, int64_t c) { return vec_alloc(r * c); }

ko_block_gmres_workspace *ko_block_gmres_workspace_create(int64_t m, int64_t n, int p, int memory) {
  /* src/block_krylov_workspaces.jl:137-163 */
  ko_block_gmres_workspace *ws = (ko_block_gmres_workspace *)calloc(1, sizeof(*ws));
  if (memory <= 0) memory = 5;
  int64_t cap = n / p;
  if ((int64_t)memory > cap) memory = (int)cap;
  ws->m = m; ws->n = n; ws->p = p; ws->mem = memory; ws->nV = memory;
  ws->X = mat_alloc(n, p); ws->W = mat_alloc(n, p);
  ws->C = mat_alloc(p, p); ws->D = mat_alloc(2 * p, p);
  ws->nR = memory * (memory + 1) / 2; ws->nH = memory;
  ws->V = (double **)calloc((size_t)memory + 1, sizeof(double *));
  ws->Z = (double **)calloc((size_t)memory + 1, sizeof(double *));
  ws->H = (double **)calloc((size_t)memory + 1, sizeof(double *));
  ws->tau = (double **)calloc((size_t)memory + 1, sizeof(double *));
  ws->R = (double **)calloc((size_t)ws->nR + 1, sizeof(double *));
  for (int i = 0; i < memory; i++) {
    ws->V[i] = mat_alloc(n, p); ws->Z[i] = mat_alloc(p, p);
    ws->H[i] = mat_alloc(2 * p, p); ws->tau[i] = vec_alloc(p);
  }
  for (int i = 0; i < ws->nR; i++) ws->R[i] = mat_alloc(p, p);
  ko_stats_init(&ws->stats);
  return ws;
}
void ko_block_gmres_workspace_free(ko_block_gmres_workspace *ws) {
  if (!ws) return;
  free(ws->dX); free(ws->X); free(ws->W); free(ws->P); free(ws->Q); free(ws->C); free(ws->D);
  for (int i = 0; i < ws->nV; i++) { free(ws->V[i]); free(ws->Z[i]); }
  for (int i = 0; i < ws->nH; i++) { free(ws->H[i]); free(ws->tau[i]); }
  for (int i = 0; i < ws->nR; i++) free(ws->R[i]);
  free(ws->V); free(ws->Z); free(ws->H); free(ws->tau); free(ws->R);
  ko_stats_free(&ws->stats);
  free(ws);
}
void ko_block_gmres_warm_start(ko_block_gmres_workspace *ws, const double *X0) {
  if (!ws->dX) ws->dX = mat_alloc(ws->n, ws->p);
  memcpy(ws->dX, X0, sizeof(double) * (size_t)ws->n * ws->p);
  ws->warm_start = 1;
}

/* C(p x q) = A(n x p)^T * B(n x q) */
static void gemm_tn(int64_t n, int p, int q, const double *A, const double *B, double *C) {
  /* every (i, j) entry is its own sequential extended-precision dot: the thread count cannot change a value */
#pragma omp parallel for collapse(2) schedule(dynamic, 1) if (n > 100000)
  for (int j = 0; j < q; j++)
    for (int i = 0; i < p; i++) {
      long double acc = 0.0L;
      const double *a = A + (size_t)i * n, *b = B + (size_t)j * n;
      for (int64_t l = 0; l < n; l++) acc += (long double)a[l] * (long double)b[l];
      C[(size_t)j * p + i] = (double)acc;
    }
}
/* C(n x q) = alpha * A(n x p) * B(p x q) + beta * C */
static void gemm_nn(int64_t n, int p, int q, double alpha, const double *A, const double *B,
                    double beta, double *C) {
  /* columns of C are independent */
#pragma omp parallel for schedule(dynamic, 1) if (n > 100000)
  for (int j = 0; j < q; j++) {
    double *c = C + (size_t)j * n;
    if (beta != 1.0) for (int64_t l = 0; l < n; l++) c[l] = beta * c[l];
    for (int i = 0; i < p; i++) {
      double s = alpha * B[(size_t)j * p + i];
      const double *a = A + (size_t)i * n;
      for (int64_t l = 0; l < n; l++) c[l] = fma(s, a[l], c[l]);
    }
  }
}
static double frob(int64_t len, const double *a) { return ko_nrm2(len, a); }

int ko_block_gmres(ko_block_gmres_workspace *ws, ko_block_matvec A, ko_block_matvec M,
                   ko_block_matvec N, void *ud, const double *B, const ko_options *opts_in) {
  ko_options o = opts_in ? *opts_in : ko_default_options();
  double t0 = now_s();
  double timemax = timemax_of(&o);
  int64_t n = ws->n;
  int p = ws->p;
  int64_t np = n * p;
  ko_stats *st = &ws->stats;
  double atol = tol_or_default(o.atol), rtol = tol_or_default(o.rtol);
  int restart = o.restart, reorth = o.reorthogonalization;
  if (ws->m != ws->n) return fail(st, "System must be square");

  int MisI = (M == NULL), NisI = (N == NULL);
  if (!MisI && !ws->Q) ws->Q = mat_alloc(n, p);
  if (!NisI && !ws->P) ws->P = mat_alloc(n, p);
  if (restart && !ws->dX) ws->dX = mat_alloc(n, p);
  double *dX = ws->dX, *X = ws->X, *W = ws->W, *C = ws->C, *D = ws->D;
  double *Psitmp = C;
  int warm_start = ws->warm_start;
  stats_reset(st);
  double *Q = MisI ? W : ws->Q;
  double *R0 = MisI ? W : ws->Q;
  double *Xr = restart ? dX : X;
  int pp = p * p;

  for (int64_t i = 0; i < np; i++) X[i] = 0.0;
  if (warm_start) {
    A(dX, W, p, ud);
    for (int64_t i = 0; i < np; i++) W[i] = B[i] - W[i];
    if (restart) for (int64_t i = 0; i < np; i++) X[i] += dX[i];
  } else {
    memcpy(W, B, sizeof(double) * (size_t)np);
  }
  if (!MisI) M(W, R0, p, ud);
  double RNorm = frob(np, R0);
  if (o.history) stats_push(st, RNorm);
  double eps_tol = atol + rtol * RNorm;

  int mem = ws->mem;
  int npass = 0;
  int64_t iter = 0;
  int inner_iter = 0;
  int64_t itmax = o.itmax == 0 ? 2 * (n / p) : o.itmax;
  int64_t inner_itmax = itmax;

  int solved = RNorm <= eps_tol;
  int tired = iter >= itmax;
  int inner_tired = inner_iter >= inner_itmax;
  int user_requested_exit = 0, overtimed = 0;
  const char *status = "unknown";

  while (!(solved || tired || user_requested_exit || overtimed)) {
    int nr = 0;
    for (int i = 0; i < mem; i++) memset(ws->V[i], 0, sizeof(double) * (size_t)np);
    for (int i = 0; i < ws->nR; i++) memset(ws->R[i], 0, sizeof(double) * (size_t)pp);
    for (int i = 0; i < ws->nV; i++) memset(ws->Z[i], 0, sizeof(double) * (size_t)pp);

    if (restart) {
      for (int64_t i = 0; i < np; i++) Xr[i] = 0.0;
      if (npass >= 1) {
        A(X, W, p, ud);
        for (int64_t i = 0; i < np; i++) W[i] = B[i] - W[i];
        if (!MisI) M(W, R0, p, ud);
      }
    }

    memcpy(ws->V[0], R0, sizeof(double) * (size_t)np);                /* :211 */
    ko_householder((int)n, p, ws->V[0], ws->Z[0], ws->tau[0], 0);     /* :212 */

    npass = npass + 1;
    inner_iter = 0;
    inner_tired = 0;

    while (!(solved || inner_tired || user_requested_exit || overtimed)) {
      inner_iter = inner_iter + 1;

      if (!restart && (inner_iter > mem)) {                           /* :224-232 */
        int newR = ws->nR + inner_iter;
        ws->R = (double **)realloc(ws->R, sizeof(double *) * (size_t)(newR + 1));
        for (int i = ws->nR; i < newR; i++) ws->R[i] = mat_alloc(p, p);
        ws->nR = newR;
        ws->H = (double **)realloc(ws->H, sizeof(double *) * (size_t)(ws->nH + 2));
        ws->tau = (double **)realloc(ws->tau, sizeof(double *) * (size_t)(ws->nH + 2));
        ws->H[ws->nH] = mat_alloc(2 * p, p);
        ws->tau[ws->nH] = vec_alloc(p);
        ws->nH++;
      }

      double *Vk = ws->V[inner_iter - 1];
      double *Pm = NisI ? Vk : ws->P;
      if (!NisI) N(Vk, Pm, p, ud);
      A(Pm, W, p, ud);                                                /* :242 */
      if (!MisI) M(W, Q, p, ud);
      for (int i = 0; i < inner_iter; i++) {                          /* :244-247 */
        gemm_tn(n, p, p, ws->V[i], Q, ws->R[nr + i]);
        gemm_nn(n, p, p, -1.0, ws->V[i], ws->R[nr + i], 1.0, Q);
      }
      if (reorth) {
        for (int i = 0; i < inner_iter; i++) {
          gemm_tn(n, p, p, ws->V[i], Q, Psitmp);
          gemm_nn(n, p, p, -1.0, ws->V[i], Psitmp, 1.0, Q);
          for (int l = 0; l < pp; l++) ws->R[nr + i][l] += Psitmp[l];
        }
      }

      double *tauk = ws->tau[inner_iter - 1];
      ko_householder((int)n, p, Q, C, tauk, 0);                       /* :259 */

      for (int i = 0; i < inner_iter - 1; i++) {                      /* :263-269 */
        for (int j = 0; j < p; j++)
          for (int l = 0; l < p; l++) {
            D[(size_t)j * 2 * p + l] = ws->R[nr + i][(size_t)j * p + l];
            D[(size_t)j * 2 * p + p + l] = ws->R[nr + i + 1][(size_t)j * p + l];
          }
        ko_ormqr_LT(2 * p, p, p, ws->H[i], 2 * p, ws->tau[i], D, 2 * p);
        for (int j = 0; j < p; j++)
          for (int l = 0; l < p; l++) {
            ws->R[nr + i][(size_t)j * p + l] = D[(size_t)j * 2 * p + l];
            ws->R[nr + i + 1][(size_t)j * p + l] = D[(size_t)j * 2 * p + p + l];
          }
      }

      double *Hk = ws->H[inner_iter - 1];                             /* :272-274 */
      for (int j = 0; j < p; j++)
        for (int l = 0; l < p; l++) {
          Hk[(size_t)j * 2 * p + l] = ws->R[nr + inner_iter - 1][(size_t)j * p + l];
          Hk[(size_t)j * 2 * p + p + l] = C[(size_t)j * p + l];
        }
      ko_householder(2 * p, p, Hk, ws->R[nr + inner_iter - 1], tauk, 1);

      double *Zk = ws->Z[inner_iter - 1];                             /* :277-280 */
      for (int j = 0; j < p; j++)
        for (int l = 0; l < p; l++) {
          D[(size_t)j * 2 * p + l] = Zk[(size_t)j * p + l];
          D[(size_t)j * 2 * p + p + l] = 0.0;
        }
      ko_ormqr_LT(2 * p, p, p, Hk, 2 * p, tauk, D, 2 * p);
      for (int j = 0; j < p; j++)
        for (int l = 0; l < p; l++) {
          Zk[(size_t)j * p + l] = D[(size_t)j * 2 * p + l];
          C[(size_t)j * p + l] = D[(size_t)j * 2 * p + p + l];        /* C .= D2 (:284) */
        }
      RNorm = frob(pp, C);
      if (o.history) stats_push(st, RNorm);
      nr = nr + inner_iter;

      if (o.callback) user_requested_exit = o.callback(ws, o.callback_data) != 0;
      solved = RNorm <= eps_tol;
      if (restart) {
        int64_t lim = (int64_t)mem < inner_itmax ? (int64_t)mem : inner_itmax;
        inner_tired = inner_iter >= lim;
      } else {
        inner_tired = inner_iter >= inner_itmax;
      }
      overtimed = (now_s() - t0) > timemax;

      if (!(solved || inner_tired || user_requested_exit || overtimed)) {
        if (!restart && (inner_iter >= mem)) {                        /* :300-305 */
          if (ws->nV <= inner_iter) {
            ws->V = (double **)realloc(ws->V, sizeof(double *) * (size_t)(inner_iter + 2));
            ws->Z = (double **)realloc(ws->Z, sizeof(double *) * (size_t)(inner_iter + 2));
            for (int i = ws->nV; i <= inner_iter; i++) {
              ws->V[i] = mat_alloc(n, p);
              ws->Z[i] = mat_alloc(p, p);
            }
            ws->nV = inner_iter + 1;
          }
        }
        memcpy(ws->V[inner_iter], Q, sizeof(double) * (size_t)np);    /* :307 */
        for (int j = 0; j < p; j++)
          for (int l = 0; l < p; l++)
            ws->Z[inner_iter][(size_t)j * p + l] = D[(size_t)j * 2 * p + p + l];
      }
    }

    /* block back-substitution (:313-321), Y aliases Z */
    double **Y = ws->Z;
    double *tmp = (double *)malloc(sizeof(double) * (size_t)pp);
    for (int i = inner_iter; i >= 1; i--) {
      int pos = nr + i - inner_iter;
      for (int j = inner_iter; j >= i + 1; j--) {
        /* Y_i <- Y_i - R[pos] * Y_j */
        const double *Rm = ws->R[pos - 1];
        for (int cc = 0; cc < p; cc++)
          for (int rr = 0; rr < p; rr++) {
            double acc = 0.0;
            for (int l = 0; l < p; l++) acc += Rm[(size_t)l * p + rr] * Y[j - 1][(size_t)cc * p + l];
            tmp[(size_t)cc * p + rr] = acc;
          }
        for (int l = 0; l < pp; l++) Y[i - 1][l] -= tmp[l];
        pos = pos - j + 1;
      }
      /* ldiv!(UpperTriangular(R[pos]), Y_i) */
      const double *U = ws->R[pos - 1];
      for (int cc = 0; cc < p; cc++) {
        double *ycol = Y[i - 1] + (size_t)cc * p;
        for (int rr = p - 1; rr >= 0; rr--) {
          double acc = ycol[rr];
          for (int l = rr + 1; l < p; l++) acc -= U[(size_t)l * p + rr] * ycol[l];
          ycol[rr] = acc / U[(size_t)rr * p + rr];
        }
      }
    }
    free(tmp);

    for (int i = 0; i < inner_iter; i++) gemm_nn(n, p, p, 1.0, ws->V[i], Y[i], 1.0, Xr); /* :324-326 */
    if (!NisI) {
      memcpy(ws->P, Xr, sizeof(double) * (size_t)np);
      N(ws->P, Xr, p, ud);
    }
    if (restart) for (int64_t i = 0; i < np; i++) X[i] += Xr[i];

    inner_itmax = inner_itmax - inner_iter;
    iter = iter + inner_iter;
    tired = iter >= itmax;
    overtimed = (now_s() - t0) > timemax;
  }

  if (tired) status = "maximum number of iterations exceeded";
  if (solved) status = "solution good enough given atol and rtol";
  if (overtimed) status = "time limit exceeded";
  if (user_requested_exit) status = "user-requested exit";

  if (warm_start && !restart) for (int64_t i = 0; i < np; i++) X[i] += dX[i];
  ws->warm_start = 0;

  st->niter = (int)iter;
  st->solved = solved;
  st->timer = now_s() - t0;
  snprintf(st->status, sizeof(st->status), "%s", status);
  return 0;
}

/* ===================================================================== *
 *  CPU baseline loop (bench.py cpu_baseline leg): `iters` iterations of the CG recurrence of
 *  src/cg.jl:195-268 (M = I, no stopping test) with the OpenMP kernels; threads = 1 is the
 *  faithful single-thread mode (SparseArrays.mul! is not threaded, docs/src/tips.md:38),
 *  threads = nproc the all-core mode (threaded_mul! recipe + OPENBLAS_NUM_THREADS, tips.md:8-55).
 *  Returns seconds per iteration (setup and allocation excluded, like stats.timer vs
 *  stats.allocation_timer); *rnorm_out = final ||r||.
 * ===================================================================== */
double ko_cg_bench(const ko_csr *A, int iters, int threads, double *rnorm_out) {
  int64_t n = A->n;
  ko_set_threads(threads);
  double *x = (double *)malloc(sizeof(double) * (size_t)n), *r = (double *)malloc(sizeof(double) * (size_t)n);
  double *p = (double *)malloc(sizeof(double) * (size_t)n), *Ap = (double *)malloc(sizeof(double) * (size_t)n);
  if (!x || !r || !p || !Ap) { free(x); free(r); free(p); free(Ap); return -1.0; }
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) { x[i] = 0.0; r[i] = 1.0; p[i] = 1.0; Ap[i] = 0.0; }   /* b = ones */
  double gamma = ko_dot_omp(n, r, r);
  /* one untimed warm-up ITERATION (it = -1: pages touched by every loop, threads spun up), then `iters` timed ones */
  double t0 = 0.0;
  for (int it = -1; it < iters; it++) {
    if (it == 0) t0 = now_s();
    ko_spmv_omp(A, p, Ap);
    double pAp = ko_dot_omp(n, p, Ap);
    double alpha = gamma / pAp;
    ko_axpy_omp(n, alpha, p, x);
    ko_axpy_omp(n, -alpha, Ap, r);
    double gamma_next = ko_dot_omp(n, r, r);
    double beta = gamma_next / gamma;
    gamma = gamma_next;
    ko_axpby_omp(n, 1.0, r, beta, p);
  }
  double dt = now_s() - t0;
  if (rnorm_out) *rnorm_out = sqrt(gamma);
  free(x); free(r); free(p); free(Ap);
  return dt / (iters > 0 ? iters : 1);
}
