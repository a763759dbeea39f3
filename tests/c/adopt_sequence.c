/* adopt_sequence.c -- the solvers on CALLER-OWNED work vectors (khip_*_workspace_adopt), as a binding that specialises
 * `cg!(ws::CgWorkspace{Float64,Float64,HIPVector}, A, b)` would drive them (julia/KrylovHIP/src/KrylovHIP.jl):
 *
 *   x, r, p, Ap = khip_malloc(8 n) each           S(undef, n), src/krylov_workspaces.jl:269-285
 *   khip_cg_workspace_adopt(ctx, n, n, x, r, p, Ap, &ws)
 *   khip_cg_solve(ws, A, NULL, b, &opts)          the fused, device-resident loop, on the caller's vectors
 *   solution(ws) === x                            test/test_interface.jl:260
 *
 * Checked against the library-owned workspace of khip_cg_workspace_create on the same problem: same iteration count, status
 * and residual history BIT FOR BIT, same solution bits, the solution in the caller's buffer, nothing of the caller's freed.
 * Also: a Jacobi M with the caller's z handed over late (allocate_if, src/cg.jl:142), a warm start through the caller's
 * Δx, gmres! with restart = false growing the caller's basis through the grow callback (src/gmres.jl:319-324), bicgstab!,
 * and block_gmres! on the caller's panels with B read in place.
 *
 *   usage: adopt_sequence [n1 ...]     (grid sizes of get_div_grad(n1, n1, n1); default 64; the GPU test adds 512)
 *   cc -O2 -Iinclude tests/c/adopt_sequence.c -Lkrylov.jl_amd -lkrylov_hip -Wl,-rpath,... -lm
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "krylov_hip.h"

#define CK(call)                                                         \
  do {                                                                   \
    int rc_ = (call);                                                    \
    if (rc_ != KHIP_OK) {                                                \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, khip_last_error());  \
      exit(1);                                                           \
    }                                                                    \
  } while (0)

static int failures = 0;
#define EXPECT(cond, ...)                                  \
  do {                                                     \
    if (!(cond)) {                                         \
      printf("FAIL %s:%d: ", __FILE__, __LINE__);          \
      printf(__VA_ARGS__);                                 \
      printf("\n");                                        \
      failures++;                                          \
    }                                                      \
  } while (0)

static double *dvec(khip_ctx *ctx, int64_t n) {           /* S(undef, n): exactly 8 n bytes, as the Julia glue allocates */
  void *p = NULL;
  CK(khip_malloc(ctx, sizeof(double) * (size_t)(n > 0 ? n : 1), &p));
  return (double *)p;
}

/* khip_options.callback as the Julia trampoline uses it (KrylovHIP.jl callback_trampoline): the history so far is readable through
 * khip_cg_stats() INSIDE the callback (published before every call, ABI 0.4); stop after `stop_at` calls */
typedef struct { khip_cg_workspace *ws; int calls, stop_at, nres_seen[8]; double last; } cb_state;
static int count_callback(void *workspace, void *userdata) {
  cb_state *c = (cb_state *)userdata;
  const khip_stats *st = khip_cg_stats((khip_cg_workspace *)workspace);
  if (c->calls < 8) c->nres_seen[c->calls] = st->nres;
  c->last = st->nres > 0 ? st->residuals[st->nres - 1] : -1.0;
  c->calls++;
  return c->calls >= c->stop_at;
}

/* history of a finished solve, copied (the stats pointer is only valid until the workspace's next solve) */
typedef struct { int niter, solved, nres; char status[96]; double *res; } snapshot;
static snapshot snap(const khip_stats *st) {
  snapshot s;
  s.niter = st->niter; s.solved = st->solved; s.nres = st->nres;
  memcpy(s.status, st->status, sizeof(s.status));
  s.res = (double *)malloc(sizeof(double) * (size_t)(st->nres > 0 ? st->nres : 1));
  if (st->nres > 0) memcpy(s.res, st->residuals, sizeof(double) * (size_t)st->nres);
  return s;
}
static int same_history(const snapshot *a, const snapshot *b) {
  return a->niter == b->niter && a->solved == b->solved && a->nres == b->nres && strcmp(a->status, b->status) == 0 &&
         memcmp(a->res, b->res, sizeof(double) * (size_t)a->nres) == 0;
}
/* x == y bit for bit (no NaN expected): || x - y ||_2 == 0 exactly, computed on the device (t is scratch) */
static int same_vector(khip_ctx *ctx, int64_t n, const double *x, const double *y, double *t) {
  double nrm = -1.0;
  CK(khip_copy(ctx, n, t, x));
  CK(khip_axpy(ctx, n, -1.0, y, t));
  CK(khip_nrm2(ctx, n, t, &nrm));
  return nrm == 0.0;
}

typedef struct { khip_ctx *ctx; int64_t n; double **list; int count, cap; } grow_state;
static double *grow_cb(void *ud) {                        /* push!(V, similar(x)) */
  grow_state *g = (grow_state *)ud;
  if (g->count == g->cap) return NULL;
  g->list[g->count] = dvec(g->ctx, g->n);
  return g->list[g->count++];
}

static void run_size(khip_ctx *ctx, int n1, int quick) {
  const int64_t n = (int64_t)n1 * n1 * n1;
  int32_t *rp = NULL, *col = NULL; double *val = NULL; int64_t nnz = 0;
  CK(khip_gen_stencil(ctx, 0, n1, n1, n1, 0, n, &rp, &col, &val, &nnz));
  khip_csr *A = NULL;
  CK(khip_csr_create(ctx, n, n, nnz, rp, 32, col, val, 0, 1, &A));
  CK(khip_free(ctx, rp)); CK(khip_free(ctx, col)); CK(khip_free(ctx, val));
  khip_operator opA = {A, NULL, NULL};
  double *b = dvec(ctx, n), *scratch = dvec(ctx, n);
  CK(khip_fill(ctx, n, b, 1.0));

  khip_options o = khip_default_options();
  o.atol = 0.0; o.rtol = 1e-8; o.itmax = (int)(n < 2000000000 ? n : 2000000000); o.history = 1;   /* benchmark/benchmarks.jl:14-21 */

  /* ---------------- cg!: library-owned workspace vs the caller's four vectors ---------------- */
  khip_cg_workspace *wo = NULL, *wa = NULL;
  CK(khip_cg_workspace_create(ctx, n, n, &wo));
  CK(khip_cg_solve(wo, &opA, NULL, b, &o));
  snapshot so = snap(khip_cg_stats(wo));

  double *x = dvec(ctx, n), *r = dvec(ctx, n), *p = dvec(ctx, n), *Ap = dvec(ctx, n);
  CK(khip_cg_workspace_adopt(ctx, n, n, x, r, p, Ap, &wa));
  EXPECT(khip_cg_solution(wa) == x, "solution(ws) is not the caller's x");
  EXPECT(khip_cg_workspace_bytes(wa) == 4 * sizeof(double) * (size_t)n, "adopted CgWorkspace is not 4 n");
  CK(khip_cg_solve(wa, &opA, NULL, b, &o));
  snapshot sa = snap(khip_cg_stats(wa));
  EXPECT(same_history(&so, &sa), "cg n1=%d: adopted history differs (niter %d vs %d, nres %d vs %d)", n1, sa.niter, so.niter, sa.nres, so.nres);
  EXPECT(so.solved && so.niter > 0, "cg n1=%d did not converge", n1);
  EXPECT(same_vector(ctx, n, khip_cg_solution(wo), x, scratch), "cg n1=%d: solution bits differ", n1);
  printf("cg   n1=%d  niter=%d  %s  adopted == owned: history %s\n", n1, sa.niter, sa.status, same_history(&so, &sa) ? "bit-identical" : "DIFFERS");
  EXPECT(khip_cg_last_path(wa) == 2 && khip_cg_last_path(wo) == 2, "cg n1=%d: the default options did not run the device-resident loop", n1);
  /* a second solve on the same adopted workspace (in-place API, test/test_allocations.jl:53-56) */
  CK(khip_cg_solve(wa, &opA, NULL, b, &o));
  snapshot sa2 = snap(khip_cg_stats(wa));
  EXPECT(same_history(&so, &sa2), "cg n1=%d: second solve on the adopted workspace differs", n1);
  free(sa2.res);

  if (!quick) {
    /* a user callback: the host-driven loop on the fused kernels (last_path 1), same bits, history visible inside the callback */
    cb_state cs; memset(&cs, 0, sizeof(cs)); cs.ws = wa; cs.stop_at = 5;
    khip_options oc = o; oc.callback = count_callback; oc.callback_data = &cs;
    CK(khip_cg_solve(wa, &opA, NULL, b, &oc));
    snapshot sc = snap(khip_cg_stats(wa));
    EXPECT(khip_cg_last_path(wa) == 1, "cg n1=%d callback: last_path %d", n1, khip_cg_last_path(wa));
    EXPECT(cs.calls == 5 && sc.niter == 5 && strcmp(sc.status, "user-requested exit") == 0, "cg n1=%d callback: calls %d niter %d status %s", n1, cs.calls, sc.niter, sc.status);
    for (int i = 0; i < 5; i++) EXPECT(cs.nres_seen[i] == i + 2, "cg n1=%d callback %d saw %d history entries", n1, i, cs.nres_seen[i]);
    EXPECT(sc.nres == 6 && memcmp(sc.res, so.res, sizeof(double) * 6) == 0 && cs.last == so.res[5], "cg n1=%d callback: history prefix differs from the device loop's", n1);
    free(sc.res);
    /* fused = 0 (the reference's primitive sequence) on the adopted vectors: same bits as on the owned ones */
    khip_options o0 = o; o0.fused = 0; o0.itmax = 60;
    CK(khip_cg_solve(wo, &opA, NULL, b, &o0)); snapshot s0o = snap(khip_cg_stats(wo));
    CK(khip_cg_solve(wa, &opA, NULL, b, &o0)); snapshot s0a = snap(khip_cg_stats(wa));
    EXPECT(same_history(&s0o, &s0a), "cg n1=%d fused=0: adopted history differs", n1);
    free(s0o.res); free(s0a.res);

    /* Jacobi M: z is allocated by the CALLER when first needed (allocate_if, src/cg.jl:142) and handed over */
    khip_operator M;
    CK(khip_jacobi_create(ctx, A, &M));
    CK(khip_cg_solve(wo, &opA, &M, b, &o)); snapshot smo = snap(khip_cg_stats(wo));
    double *z = dvec(ctx, n);
    CK(khip_cg_workspace_adopt_vector(wa, "z", z));
    EXPECT(khip_cg_vector(wa, "z") == z, "adopted z is not what khip_cg_vector returns");
    CK(khip_cg_solve(wa, &opA, &M, b, &o)); snapshot sma = snap(khip_cg_stats(wa));
    EXPECT(same_history(&smo, &sma), "cg n1=%d Jacobi: adopted history differs", n1);
    EXPECT(same_vector(ctx, n, khip_cg_solution(wo), x, scratch), "cg n1=%d Jacobi: solution bits differ", n1);
    free(smo.res); free(sma.res);
    CK(khip_jacobi_destroy(&M));

    /* warm start through the caller's Δx: kcopy!(n, ws.Δx, x0) on the caller's side, then only the flag */
    double *x0 = dvec(ctx, n), *dx = dvec(ctx, n);
    CK(khip_fill(ctx, n, x0, 0.125));
    CK(khip_cg_warm_start(wo, x0));
    CK(khip_cg_solve(wo, &opA, NULL, b, &o)); snapshot swo = snap(khip_cg_stats(wo));
    CK(khip_cg_workspace_adopt_vector(wa, "dx", dx));
    CK(khip_copy(ctx, n, dx, x0));
    CK(khip_cg_warm_start(wa, dx));
    CK(khip_cg_solve(wa, &opA, NULL, b, &o)); snapshot swa = snap(khip_cg_stats(wa));
    EXPECT(same_history(&swo, &swa), "cg n1=%d warm start: adopted history differs", n1);
    EXPECT(same_vector(ctx, n, khip_cg_solution(wo), x, scratch), "cg n1=%d warm start: solution bits differ", n1);
    free(swo.res); free(swa.res);
    CK(khip_cg_workspace_adopt_vector(wa, "dx", NULL));       /* empty the slots again: the caller frees its own vectors */
    CK(khip_cg_workspace_adopt_vector(wa, "z", NULL));
    CK(khip_free(ctx, x0)); CK(khip_free(ctx, dx)); CK(khip_free(ctx, z));
    EXPECT(khip_cg_workspace_adopt_vector(wa, "x", NULL) == KHIP_ERR_INVALID, "emptying x must be refused");
    EXPECT(khip_cg_workspace_adopt_vector(wa, "nope", x) == KHIP_ERR_INVALID, "unknown vector name must be refused");
  }
  CK(khip_cg_workspace_destroy(wa));
  CK(khip_fill(ctx, n, x, 3.0));                                /* the caller's vectors outlive the workspace */
  CK(khip_fill(ctx, n, Ap, 3.0));
  CK(khip_free(ctx, x)); CK(khip_free(ctx, r)); CK(khip_free(ctx, p)); CK(khip_free(ctx, Ap));
  CK(khip_cg_workspace_destroy(wo));
  free(so.res); free(sa.res);

  if (!quick) {
    /* ---------------- bicgstab! on the caller's six vectors ---------------- */
    khip_bicgstab_workspace *bo = NULL, *ba = NULL;
    CK(khip_bicgstab_workspace_create(ctx, n, n, &bo));
    CK(khip_bicgstab_solve(bo, &opA, NULL, NULL, b, NULL, &o)); snapshot sbo = snap(khip_bicgstab_stats(bo));
    double *six[6];
    for (int i = 0; i < 6; i++) six[i] = dvec(ctx, n);
    CK(khip_bicgstab_workspace_adopt(ctx, n, n, six[0], six[1], six[2], six[3], six[4], six[5], &ba));
    EXPECT(khip_bicgstab_solution(ba) == six[0], "bicgstab: solution(ws) is not the caller's x");
    CK(khip_bicgstab_solve(ba, &opA, NULL, NULL, b, NULL, &o)); snapshot sba = snap(khip_bicgstab_stats(ba));
    EXPECT(same_history(&sbo, &sba), "bicgstab n1=%d: adopted history differs", n1);
    EXPECT(same_vector(ctx, n, khip_bicgstab_solution(bo), six[0], scratch), "bicgstab n1=%d: solution bits differ", n1);
    printf("bicgstab n1=%d niter=%d %s\n", n1, sba.niter, sba.status);
    CK(khip_bicgstab_workspace_destroy(ba)); CK(khip_bicgstab_workspace_destroy(bo));
    for (int i = 0; i < 6; i++) CK(khip_free(ctx, six[i]));
    free(sbo.res); free(sba.res);

    /* ---------------- gmres!: restart = true on the caller's basis; restart = false growing it ---------------- */
    const int mem = 10;
    for (int restart = 1; restart >= 0; restart--) {
      khip_options og = o; og.restart = restart; og.itmax = restart ? 400 : 35;   /* 35 > mem: the basis must grow */
      khip_gmres_workspace *go = NULL, *ga = NULL;
      CK(khip_gmres_workspace_create(ctx, n, n, mem, &go));
      CK(khip_gmres_solve(go, &opA, NULL, NULL, b, &og)); snapshot sgo = snap(khip_gmres_stats(go));
      double *gx = dvec(ctx, n), *gw = dvec(ctx, n), *V[64];
      for (int i = 0; i < mem; i++) V[i] = dvec(ctx, n);
      CK(khip_gmres_workspace_adopt(ctx, n, n, mem, gx, gw, V, &ga));
      grow_state g = {ctx, n, V, mem, 64};
      CK(khip_gmres_workspace_set_grow(ga, grow_cb, &g));
      double *gdx = NULL;
      if (restart) { gdx = dvec(ctx, n); CK(khip_gmres_workspace_adopt_vector(ga, "dx", gdx)); }   /* allocate_if(restart, ws, :Δx, ...) */
      CK(khip_gmres_solve(ga, &opA, NULL, NULL, b, &og)); snapshot sga = snap(khip_gmres_stats(ga));
      EXPECT(same_history(&sgo, &sga), "gmres n1=%d restart=%d: adopted history differs (niter %d vs %d)", n1, restart, sga.niter, sgo.niter);
      EXPECT(same_vector(ctx, n, khip_gmres_solution(go), gx, scratch), "gmres n1=%d restart=%d: solution bits differ", n1, restart);
      EXPECT(khip_gmres_solution(ga) == gx, "gmres: solution(ws) is not the caller's x");
      if (!restart) EXPECT(g.count > mem, "gmres restart=false: the grow callback was never asked (basis %d)", g.count);
      else EXPECT(g.count == mem, "gmres restart=true must not grow the basis");
      int len = 0, inner = 0;
      double cs[64], ss[64];
      CK(khip_gmres_host_state(ga, 64, cs, ss, NULL, NULL, &len, &inner));
      EXPECT(len >= mem && inner >= 1, "gmres host state: len %d inner_iter %d", len, inner);
      for (int i = 0; i < inner && i < 64; i++) EXPECT(fabs(cs[i] * cs[i] + ss[i] * ss[i] - 1.0) < 1e-14, "Givens pair %d is not a rotation", i);
      /* second solve: hand the (possibly grown) basis over again, as the binding does before every solve */
      CK(khip_gmres_workspace_adopt_basis(ga, g.count, V));
      CK(khip_gmres_solve(ga, &opA, NULL, NULL, b, &og)); snapshot sga2 = snap(khip_gmres_stats(ga));
      EXPECT(same_history(&sgo, &sga2), "gmres n1=%d restart=%d: second adopted solve differs", n1, restart);
      printf("gmres n1=%d restart=%d niter=%d basis=%d %s\n", n1, restart, sga.niter, g.count, sga.status);
      CK(khip_gmres_workspace_destroy(ga)); CK(khip_gmres_workspace_destroy(go));
      for (int i = 0; i < g.count; i++) CK(khip_free(ctx, V[i]));
      CK(khip_free(ctx, gx)); CK(khip_free(ctx, gw));
      if (gdx) CK(khip_free(ctx, gdx));
      free(sgo.res); free(sga.res); free(sga2.res);
    }

    /* ---------------- block_gmres! on the caller's panels, B read in place ---------------- */
    const int pb = 8, bmem = 3;
    int64_t np = 0;
    CK(khip_panel_rows(n, &np));
    double *Bh = (double *)malloc(sizeof(double) * (size_t)n * pb);
    for (int j = 0; j < pb; j++)
      for (int64_t i = 0; i < n; i++) Bh[(size_t)j * n + i] = 1.0 + sin(0.37 * (double)(j + 1) * (double)(i % 1013));   /* independent columns */
    double *Bc = dvec(ctx, n * pb), *Bp = dvec(ctx, np * pb), *Xc = dvec(ctx, n * pb), *Xp = dvec(ctx, np * pb), *pan[16];
    CK(khip_memcpy_h2d(ctx, Bc, Bh, sizeof(double) * (size_t)n * pb));
    CK(khip_fill(ctx, np * pb, Bp, 0.0));
    CK(khip_panel_from_colmajor(ctx, n, pb, Bc, Bp));
    for (int restart = 0; restart <= 1; restart++) {
      khip_options ob = khip_default_options();
      ob.history = 1; ob.restart = restart; ob.itmax = restart ? 30 : 7;          /* 7 > bmem: the basis grows */
      khip_block_gmres_workspace *ko = NULL, *ka = NULL;
      CK(khip_block_gmres_workspace_create(ctx, n, n, pb, bmem, &ko));
      CK(khip_block_gmres_solve(ko, &opA, NULL, NULL, Bc, &ob)); snapshot sko = snap(khip_block_gmres_stats(ko));
      CK(khip_block_gmres_get_X(ko, Xc));
      for (int i = 0; i < 2 + bmem; i++) { pan[i] = dvec(ctx, np * pb); CK(khip_fill(ctx, np * pb, pan[i], 0.0)); }
      CK(khip_block_gmres_workspace_adopt(ctx, n, n, pb, bmem, pan[0], pan[1], pan + 2, &ka));
      grow_state g = {ctx, np * pb, pan + 2, bmem, 14};
      CK(khip_block_gmres_workspace_set_grow(ka, grow_cb, &g));
      double *dXp = NULL;
      if (restart) { dXp = dvec(ctx, np * pb); CK(khip_fill(ctx, np * pb, dXp, 0.0)); CK(khip_block_gmres_workspace_adopt_panel(ka, "dX", dXp)); }
      CK(khip_block_gmres_solve_panel(ka, &opA, NULL, NULL, Bp, &ob)); snapshot ska = snap(khip_block_gmres_stats(ka));
      EXPECT(same_history(&sko, &ska), "block_gmres n1=%d restart=%d: adopted history differs (niter %d vs %d)", n1, restart, ska.niter, sko.niter);
      CK(khip_fill(ctx, np * pb, Xp, 0.0));
      CK(khip_panel_from_colmajor(ctx, n, pb, Xc, Xp));
      double *t2 = dvec(ctx, np * pb);
      EXPECT(same_vector(ctx, np * pb, Xp, pan[0], t2), "block_gmres n1=%d restart=%d: X panel bits differ", n1, restart);
      CK(khip_free(ctx, t2));
      if (!restart) EXPECT(g.count > bmem, "block_gmres restart=false: the grow callback was never asked");
      printf("block_gmres n1=%d restart=%d niter=%d panels=%d %s\n", n1, restart, ska.niter, g.count, ska.status);
      CK(khip_block_gmres_workspace_destroy(ka)); CK(khip_block_gmres_workspace_destroy(ko));
      for (int i = 0; i < 2 + g.count; i++) CK(khip_free(ctx, pan[i]));
      if (dXp) CK(khip_free(ctx, dXp));
      free(sko.res); free(ska.res);
    }
    CK(khip_free(ctx, Bc)); CK(khip_free(ctx, Bp)); CK(khip_free(ctx, Xc)); CK(khip_free(ctx, Xp));
    free(Bh);
  }
  CK(khip_free(ctx, b)); CK(khip_free(ctx, scratch));
  CK(khip_csr_destroy(A));
}

int main(int argc, char **argv) {
  khip_ctx *ctx = NULL;
  CK(khip_ctx_create(0, NULL, &ctx));
  int maj = 0, min = 0;
  khip_version(&maj, &min);
  if (maj != KHIP_VERSION_MAJOR || min < 4) { fprintf(stderr, "library %d.%d lacks the adopt / last_path entry points\n", maj, min); return 1; }
  if (argc <= 1) run_size(ctx, 64, 0);
  for (int i = 1; i < argc; i++) {
    const int n1 = atoi(argv[i]);
    run_size(ctx, n1, n1 > 128);            /* the large grids run the cg! comparison only (what bench.py times) */
  }
  CK(khip_ctx_destroy(ctx));
  printf("%d failure(s)\n", failures);
  if (failures == 0) printf("PASS\n");
  return failures != 0;
}
