/* cg_poisson.c -- plain C against include/krylov_hip.h: CG on the 3-D Poisson operator, everything resident in HBM.
 *
 *   cc -O2 -Iinclude examples/cg_poisson.c -Lkrylov.jl_amd -lkrylov_hip -Wl,-rpath,$PWD/krylov.jl_amd -lm -o cg_poisson
 *   ./cg_poisson 64
 *
 * Mirrors the reference's own C example (interfaces/examples/C/basic_cg.c) with its single device enumerator replaced by
 * device pointers: workspace create -> solve -> stats -> solution.  Prints niter / status / true residual. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "krylov_hip.h"

#define CK(call)                                                              \
  do {                                                                        \
    int rc_ = (call);                                                         \
    if (rc_ != KHIP_OK) {                                                     \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, khip_last_error());       \
      return 1;                                                               \
    }                                                                         \
  } while (0)

int main(int argc, char **argv) {
  const int n1 = argc > 1 ? atoi(argv[1]) : 32;
  const int64_t n = (int64_t)n1 * n1 * n1;
  khip_ctx *ctx = NULL;
  CK(khip_ctx_create(0, NULL, &ctx));

  /* operator: get_div_grad(n1, n1, n1) generated on the device, then wrapped as a CSR handle without a copy to the host */
  int32_t *rowptr = NULL, *col = NULL;
  double *val = NULL;
  int64_t nnz = 0;
  CK(khip_gen_stencil(ctx, 0 /* get_div_grad: 7-point Poisson */, n1, n1, n1, 0, n, &rowptr, &col, &val, &nnz));
  khip_csr *A = NULL;
  CK(khip_csr_create(ctx, n, n, nnz, rowptr, 32, col, val, 0, /*on_device=*/1, &A));
  CK(khip_free(ctx, rowptr)); CK(khip_free(ctx, col)); CK(khip_free(ctx, val));

  double *b = NULL;
  CK(khip_malloc(ctx, sizeof(double) * (size_t)n, (void **)&b));
  CK(khip_fill(ctx, n, b, 1.0));

  khip_cg_workspace *ws = NULL;
  CK(khip_cg_workspace_create(ctx, n, n, &ws));
  khip_operator opA = {A, NULL, NULL};
  khip_options o = khip_default_options();
  o.rtol = 1e-8; o.atol = 0.0; o.history = 1; o.fused = 2;
  CK(khip_cg_solve(ws, &opA, NULL, b, &o));
  const khip_stats *st = khip_cg_stats(ws);

  /* true residual ||b - A x|| with the same primitives */
  double *r = NULL, rn = 0.0, bn = 0.0;
  CK(khip_malloc(ctx, sizeof(double) * (size_t)n, (void **)&r));
  CK(khip_spmv(ctx, A, khip_cg_solution(ws), r));
  CK(khip_axpby(ctx, n, 1.0, b, -1.0, r));
  CK(khip_nrm2(ctx, n, r, &rn));
  CK(khip_nrm2(ctx, n, b, &bn));
  printf("n = %lld, nnz = %lld\nSolved: %s\nniter: %d\nstatus: %s\nrelative residual: %.3e\nfirst/last history: %.6e %.6e\n",
         (long long)n, (long long)nnz, st->solved ? "yes" : "no", st->niter, st->status, rn / bn, st->residuals[0],
         st->residuals[st->nres - 1]);

  khip_free(ctx, r); khip_free(ctx, b);
  khip_cg_workspace_destroy(ws);
  khip_csr_destroy(A);
  khip_ctx_destroy(ctx);
  return (st->solved && rn / bn < 1e-6) ? 0 : 2;
}
