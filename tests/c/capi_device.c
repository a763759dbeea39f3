/* capi_device.c -- the KRYLOV_HIP device mode of libkrylov_hip_capi.so (include/krylov_hip_ext.h), driven through
 * the reference's own interface (krylov.h).  Built by `make -C oracle refhip` (needs the reference header), run
 * on a GPU box by tests/test_gpu_refclients.py.  Prints PASS/FAIL lines; exit code = number of failures.
 *
 *  1. cg   : built-in CSR operator (krylov_hip_set_csr), b on the device, matvec_A = NULL     -> niter as in host mode
 *  2. gmres: device CALLBACK (calls khip_spmv on the device pointers it is handed), right-hand side on the device
 *  3. block_gmres: built-in operator, column-major device block
 *  4. the same cg problem in KRYLOV_CPU mode with a host callback gives the same iteration count and solution
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "krylov.h"
#include "krylov_hip_ext.h"
#include "krylov_hip.h"

static int fails = 0;
#define CHECK(cond, msg) do { if (cond) printf("  PASS  %s\n", msg); else { printf("  FAIL  %s\n", msg); ++fails; } } while (0)

/* tridiag(-1, 4, -1.5), n rows, CSR, 0-based */
static void build(int n, long long *rowptr, int *col, double *val, long long *nnz) {
  long long k = 0;
  for (int i = 0; i < n; ++i) {
    rowptr[i] = k;
    if (i > 0) { col[k] = i - 1; val[k++] = -1.0; }
    col[k] = i; val[k++] = 4.0;
    if (i + 1 < n) { col[k] = i + 1; val[k++] = -1.5; }
  }
  rowptr[n] = k;
  *nnz = k;
}

typedef struct { int n; const long long *rowptr; const int *col; const double *val; khip_csr *dev; } Op;

static void host_matvec(const void *xv, void *yv, void *ud) {
  const Op *op = (const Op *)ud;
  const double *x = (const double *)xv; double *y = (double *)yv;
  for (int i = 0; i < op->n; ++i) {
    double acc = 0.0;
    for (long long q = op->rowptr[i]; q < op->rowptr[i + 1]; ++q) acc = acc + op->val[q] * x[op->col[q]];
    y[i] = acc;
  }
}
static void dev_matvec(const void *x, void *y, void *ud) {      /* x, y are DEVICE pointers in KRYLOV_HIP mode */
  const Op *op = (const Op *)ud;
  khip_spmv((khip_ctx *)krylov_hip_context(), op->dev, (const double *)x, (double *)y);
}

int main(void) {
  enum { N = 2000, P = 4 };
  static long long rowptr[N + 1];
  static int col[3 * N];
  static double val[3 * N], b[N], x_host[N], x_dev[N], B[N * P], X[N * P];
  long long nnz;
  build(N, rowptr, col, val, &nnz);
  Op op = {N, rowptr, col, val, NULL};
  double ones[N];
  for (int i = 0; i < N; ++i) ones[i] = 1.0;
  host_matvec(ones, b, &op);                                   /* b = A * ones -> x = ones */

  khip_ctx *ctx = (khip_ctx *)krylov_hip_context();
  CHECK(ctx != NULL, "context");
  if (!ctx) return 1;
  double *d_b = NULL, *d_x = NULL;
  khip_malloc(ctx, sizeof(double) * N, (void **)&d_b);
  khip_malloc(ctx, sizeof(double) * N, (void **)&d_x);
  khip_memcpy_h2d(ctx, d_b, b, sizeof(double) * N);

  KrylovOptions o = krylov_default_options();
  o.rtol = 1e-10; o.atol = 0.0;

  /* 4 first: reference behaviour in host mode */
  void *ws = NULL;
  printf("bicgstab, KRYLOV_CPU, host callback ...\n");
  CHECK(krylov_workspace_create(KRYLOV_BICGSTAB, N, N, KRYLOV_FLOAT64, KRYLOV_CPU, NULL, &ws) == 0, "workspace (host)");
  CHECK(krylov_solve(ws, host_matvec, NULL, NULL, NULL, b, NULL, &op, &o) == 0, "solve (host)");
  const int it_host = krylov_niter(ws);
  CHECK(krylov_is_solved(ws) == 1 && it_host > 0, "solved (host)");
  krylov_get_x(ws, x_host, N);
  krylov_workspace_free(ws);

  /* 1: device mode, built-in operator */
  printf("bicgstab, KRYLOV_HIP, built-in CSR operator ...\n");
  CHECK(krylov_workspace_create(KRYLOV_BICGSTAB, N, N, KRYLOV_FLOAT64, KRYLOV_HIP, NULL, &ws) == 0, "workspace (device)");
  CHECK(krylov_hip_set_csr(ws, nnz, rowptr, 64, col, val, 0, 0) == 0, "set_csr");
  CHECK(krylov_solve(ws, NULL, NULL, NULL, NULL, d_b, NULL, NULL, &o) == 0, "solve (device, matvec_A = NULL)");
  CHECK(krylov_is_solved(ws) == 1, "solved (device)");
  CHECK(krylov_niter(ws) == it_host, "same iteration count as the host-callback run");
  krylov_get_x(ws, d_x, N);                                    /* device buffer */
  khip_memcpy_d2h(ctx, x_dev, d_x, sizeof(double) * N);
  double err = 0, diff = 0;
  for (int i = 0; i < N; ++i) { err = fmax(err, fabs(x_dev[i] - 1.0)); diff = fmax(diff, fabs(x_dev[i] - x_host[i])); }
  CHECK(err < 1e-7, "x = ones");
  CHECK(diff < 1e-9, "same solution as the host-callback run");
  krylov_workspace_free(ws);

  /* 2: device callback */
  printf("gmres, KRYLOV_HIP, device callback ...\n");
  khip_csr_create(ctx, N, N, nnz, rowptr, 64, col, val, 0, 0, &op.dev);
  KrylovWorkspaceOptions wo = krylov_default_workspace_options();
  wo.memory = 30;
  CHECK(krylov_workspace_create(KRYLOV_GMRES, N, N, KRYLOV_FLOAT64, KRYLOV_HIP, &wo, &ws) == 0, "workspace");
  o.restart = 1;
  CHECK(krylov_solve(ws, dev_matvec, NULL, NULL, NULL, d_b, NULL, &op, &o) == 0, "solve (device callback)");
  CHECK(krylov_is_solved(ws) == 1, "solved");
  krylov_get_x(ws, d_x, N);
  khip_memcpy_d2h(ctx, x_dev, d_x, sizeof(double) * N);
  err = 0;
  for (int i = 0; i < N; ++i) err = fmax(err, fabs(x_dev[i] - 1.0));
  CHECK(err < 1e-6, "x = ones");
  krylov_workspace_free(ws);
  o.restart = 0;

  /* 3: block interface, device mode, built-in operator */
  printf("block_gmres, KRYLOV_HIP, built-in CSR operator ...\n");
  for (int j = 0; j < P; ++j) {                                /* B(:, j) = A * (j + 1) * ones  (column-major) */
    for (int i = 0; i < N; ++i) B[j * N + i] = (j + 1) * b[i] + ((i % (j + 2)) == 0 ? 0.0 : 0.0);
  }
  /* make the block full rank: add A * e-like perturbations */
  { double v[N], w[N];
    for (int j = 0; j < P; ++j) { for (int i = 0; i < N; ++i) v[i] = cos(0.001 * (j + 1) * i); host_matvec(v, w, &op);
                                  for (int i = 0; i < N; ++i) B[j * N + i] += w[i]; } }
  double *d_B = NULL, *d_X = NULL;
  khip_malloc(ctx, sizeof(double) * N * P, (void **)&d_B);
  khip_malloc(ctx, sizeof(double) * N * P, (void **)&d_X);
  khip_memcpy_h2d(ctx, d_B, B, sizeof(double) * N * P);
  CHECK(krylov_block_workspace_create(KRYLOV_BLOCK_GMRES, N, N, P, KRYLOV_FLOAT64, KRYLOV_HIP, NULL, &ws) == 0, "block workspace");
  CHECK(krylov_hip_set_csr(ws, nnz, rowptr, 64, col, val, 0, 0) == 0, "set_csr");
  CHECK(krylov_block_solve(ws, NULL, NULL, NULL, d_B, NULL, &o) == 0, "block solve");
  CHECK(krylov_block_is_solved(ws) == 1, "block solved");
  CHECK(krylov_block_get_X(ws, d_X, N, P) == 0, "block_get_X (device buffer)");
  khip_memcpy_d2h(ctx, X, d_X, sizeof(double) * N * P);
  err = 0;
  for (int j = 0; j < P; ++j)
    for (int i = 0; i < N; ++i) err = fmax(err, fabs(X[j * N + i] - ((j + 1) + cos(0.001 * (j + 1) * i))));
  CHECK(err < 1e-6, "X = the known solution");
  krylov_block_workspace_free(ws);

  /* error handling of the extension */
  CHECK(krylov_workspace_create(KRYLOV_CG, N, N, KRYLOV_FLOAT64, (KrylovDeviceType)7, NULL, &ws) == -1, "unknown device -> -1");
  CHECK(krylov_workspace_create(KRYLOV_CG, N, N, KRYLOV_FLOAT64, KRYLOV_HIP, NULL, &ws) == 0, "cg workspace");
  CHECK(krylov_solve(ws, NULL, NULL, NULL, NULL, d_b, NULL, NULL, &o) == -1, "no callback and no operator -> -1");
  krylov_workspace_free(ws);

  khip_csr_destroy(op.dev);
  khip_free(ctx, d_b); khip_free(ctx, d_x); khip_free(ctx, d_B); khip_free(ctx, d_X);
  printf("%d failure(s)\n", fails);
  return fails;
}
