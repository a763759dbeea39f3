/*
 * quad_reference.c -- the oracle's recurrences in IEEE binary128 (__float128), as the yardstick for "which side
 * carries the rounding": TEST INFRASTRUCTURE ONLY (see krylov_oracle.h).
 *
 * The same source as the oracle (oracle/krylov_oracle.c: src/cg.jl:120-291, src/gmres.jl:121-384,
 * src/bicgstab.jl:125-277, src/block_gmres.jl:110-358 restated) is compiled with every `double` / `long double`
 * replaced by __float128 (sed, see the Makefile target `quadref`): same operations in the same order, 113-bit
 * significands, so its residual histories are exact to ~1e-30 relative to the recurrences themselves.  Tolerances,
 * eps-based thresholds and the inputs (matrix values, right-hand sides) are the double-precision ones.
 * A double-precision implementation's distance to THIS history is its own rounding error; two double implementations
 * (the oracle and the HIP path) cannot be expected to agree better than the larger of their two distances.
 *
 * Usage: quad_reference case.bin  ->  one JSON object on stdout.   case.bin (native endian):
 *   int32[8]  solver (0 cg, 1 gmres, 2 bicgstab, 3 block_gmres), matrix (0 get_div_grad, 1 kron_unsymmetric,
 *             2 27-point), n1, p, memory, restart, reorthogonalization, itmax
 *   double[2] atol, rtol (NaN = solver default)
 *   double[n*p] right-hand side(s), column-major
 */
#include <math.h>
#include <quadmath.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define KO_REAL __float128
#define KO_ACC __float128
#define sqrt sqrtq
#define fabs fabsq
#define fma fmaq
#define copysign copysignq
#define pow powq
#undef isnan
#define isnan isnanq

#include "_ref/krylov_oracle_q.c"

static void die(const char *m) { fprintf(stderr, "quad_reference: %s\n", m); exit(2); }

int main(int argc, char **argv) {
  if (argc < 2) die("usage: quad_reference case.bin");
  FILE *f = fopen(argv[1], "rb");
  if (!f) die("cannot open the case file");
  int32_t h[8];
  double tol[2];
  if (fread(h, sizeof(int32_t), 8, f) != 8 || fread(tol, sizeof(double), 2, f) != 2) die("short header");
  const int solver = h[0], kind = h[1], n1 = h[2], p = h[3] > 0 ? h[3] : 1, memory = h[4];
  ko_csr A;
  int rc = kind == 0 ? ko_csr_poisson3d(n1, n1, n1, &A) : (kind == 1 ? ko_csr_kron_unsymmetric(n1, &A) : ko_csr_stencil27_unsym(n1, &A));
  if (rc) die("generator failed");
  const int64_t n = A.n;
  double *bd = (double *)malloc(sizeof(double) * (size_t)n * p);
  if (fread(bd, sizeof(double), (size_t)n * p, f) != (size_t)n * p) die("short right-hand side");
  fclose(f);
  KO_REAL *b = (KO_REAL *)malloc(sizeof(KO_REAL) * (size_t)n * p);
  for (int64_t i = 0; i < n * p; i++) b[i] = bd[i];
  ko_options o = ko_default_options();
  if (!(tol[0] != tol[0])) o.atol = tol[0];
  if (!(tol[1] != tol[1])) o.rtol = tol[1];
  o.restart = h[5]; o.reorthogonalization = h[6]; o.itmax = h[7]; o.history = 1;
  ko_stats *st = NULL;
  if (solver == 0) {
    ko_cg_workspace *ws = ko_cg_workspace_create(n, n);
    rc = ko_cg(ws, ko_csr_matvec, NULL, &A, b, &o); st = &ws->stats;
  } else if (solver == 1) {
    ko_gmres_workspace *ws = ko_gmres_workspace_create(n, n, memory);
    rc = ko_gmres(ws, ko_csr_matvec, NULL, NULL, &A, b, &o); st = &ws->stats;
  } else if (solver == 2) {
    ko_bicgstab_workspace *ws = ko_bicgstab_workspace_create(n, n);
    rc = ko_bicgstab(ws, ko_csr_matvec, NULL, NULL, &A, b, NULL, &o); st = &ws->stats;
  } else {
    ko_block_gmres_workspace *ws = ko_block_gmres_workspace_create(n, n, p, memory);
    rc = ko_block_gmres(ws, ko_csr_block_matvec, NULL, NULL, &A, b, &o); st = &ws->stats;
  }
  if (rc) die("solver returned an error");
  printf("{\"niter\": %d, \"solved\": %d, \"status\": \"%s\", \"residuals\": [", st->niter, st->solved, st->status);
  for (int i = 0; i < st->nres; i++) printf("%s%.17g", i ? ", " : "", (double)st->residuals[i]);
  printf("]}\n");
  return 0;
}
