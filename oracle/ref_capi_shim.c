/*
 * ref_capi_shim.c -- exposes the oracle through the reference's own C ABI
 * (interfaces/include/krylov.h) so that the reference's Julia-free C clients
 * (interfaces/test/C/test_api.c, test_all_solvers.c, test_block.c and
 * interfaces/examples/C/basic_cg.c) can be compiled FROM WHERE THEY LIE under
 * /root/reference and run against the oracle.  That is how the oracle is pinned
 * to the reference's own known answers.
 *
 * TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile into oracle/_ref/ and only
 * when /root/reference exists (the header is included from there, never copied).
 *
 * Behaviour restated from interfaces/src/LibKrylov.jl:100-260 and
 * interfaces/src/c_stores.jl:186-300,377-396: opaque handles, NaN/0 sentinels
 * (c_stores.jl:255-260), return codes 0 / -1 / -2, free returns 1 when the
 * handle is unknown, get_y returns -2 for one-solution solvers.
 * Only Float64 CG / GMRES / BiCGSTAB / block-GMRES are in scope; every other
 * (solver, dtype) pair answers -2 exactly like an unknown pair.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "krylov.h"          /* from /root/reference/interfaces/include (-I) */
#include "krylov_oracle.h"

typedef enum { K_CG, K_GMRES, K_BICGSTAB, K_BLOCK_GMRES } kind_t;

typedef struct handle {
  kind_t kind;
  int m, n, p;
  void *ws;
  struct handle *next;
} handle;

static handle *g_handles = NULL;

static handle *find(void *h) {
  for (handle *it = g_handles; it; it = it->next)
    if (it == (handle *)h) return it;
  return NULL;
}

static ko_stats *stats_of(handle *h) {
  switch (h->kind) {
    case K_CG: return &((ko_cg_workspace *)h->ws)->stats;
    case K_GMRES: return &((ko_gmres_workspace *)h->ws)->stats;
    case K_BICGSTAB: return &((ko_bicgstab_workspace *)h->ws)->stats;
    default: return &((ko_block_gmres_workspace *)h->ws)->stats;
  }
}

KrylovWorkspaceOptions krylov_default_workspace_options(void) {
  KrylovWorkspaceOptions w;
  memset(&w, 0, sizeof(w));
  return w;
}

KrylovOptions krylov_default_options(void) {
  KrylovOptions o;
  memset(&o, 0, sizeof(o));
  o.atol = NAN; o.rtol = NAN; o.tau = NAN; o.nu = NAN; o.timemax = NAN;
  return o;
}

void krylov_get_version(int *major, int *minor, int *patch) {
  *major = KRYLOV_VERSION_MAJOR; *minor = KRYLOV_VERSION_MINOR; *patch = KRYLOV_VERSION_PATCH;
}

int krylov_workspace_create(KrylovSolverType solver, int m, int n, KrylovDataType dtype,
                            KrylovDeviceType device, const KrylovWorkspaceOptions *wopts,
                            void **ws_out) {
  (void)device;
  if (dtype != KRYLOV_FLOAT64) return -2;
  int memory = (wopts && wopts->memory > 0) ? wopts->memory : 20;
  handle *h = (handle *)calloc(1, sizeof(*h));
  h->m = m; h->n = n;
  switch ((int)solver) {
    case KRYLOV_CG: h->kind = K_CG; h->ws = ko_cg_workspace_create(m, n); break;
    case KRYLOV_GMRES: h->kind = K_GMRES; h->ws = ko_gmres_workspace_create(m, n, memory); break;
    case KRYLOV_BICGSTAB: h->kind = K_BICGSTAB; h->ws = ko_bicgstab_workspace_create(m, n); break;
    default: free(h); return -2;
  }
  h->next = g_handles; g_handles = h;
  *ws_out = h;
  return 0;
}

/* adaptors: the reference callback type has the same shape as ko_matvec */
typedef struct { KrylovMatvec A, M, N; void *ud; } cb_t;
static void call_A(const double *x, double *y, void *c) { ((cb_t *)c)->A(x, y, ((cb_t *)c)->ud); }
static void call_M(const double *x, double *y, void *c) { ((cb_t *)c)->M(x, y, ((cb_t *)c)->ud); }
static void call_N(const double *x, double *y, void *c) { ((cb_t *)c)->N(x, y, ((cb_t *)c)->ud); }

static ko_options map_opts(const KrylovOptions *opts) {
  ko_options o = ko_default_options();
  if (!opts) return o;
  o.atol = opts->atol; o.rtol = opts->rtol; o.itmax = opts->itmax;
  o.timemax = opts->timemax; o.radius = opts->radius; o.linesearch = opts->linesearch;
  o.restart = opts->restart; o.reorthogonalization = opts->reorthogonalization;
  return o;
}

int krylov_solve(void *ws, KrylovMatvec matvec_A, KrylovMatvec matvec_At, KrylovMatvec matvec_M,
                 KrylovMatvec matvec_N, const void *b, const void *c, void *userdata,
                 const KrylovOptions *opts) {
  (void)matvec_At;
  handle *h = find(ws);
  if (!h || !matvec_A) return -1;
  cb_t cb = {matvec_A, matvec_M, matvec_N, userdata};
  ko_options o = map_opts(opts);
  switch (h->kind) {
    case K_CG:
      return ko_cg((ko_cg_workspace *)h->ws, call_A, matvec_M ? call_M : NULL, &cb,
                   (const double *)b, &o);
    case K_GMRES:
      return ko_gmres((ko_gmres_workspace *)h->ws, call_A, matvec_M ? call_M : NULL,
                      matvec_N ? call_N : NULL, &cb, (const double *)b, &o);
    case K_BICGSTAB:
      return ko_bicgstab((ko_bicgstab_workspace *)h->ws, call_A, matvec_M ? call_M : NULL,
                         matvec_N ? call_N : NULL, &cb, (const double *)b, (const double *)c, &o);
    default: return -1;
  }
}

int krylov_get_x(void *ws, void *x, int n) {
  handle *h = find(ws);
  if (!h) return -1;
  const double *src = NULL;
  switch (h->kind) {
    case K_CG: src = ((ko_cg_workspace *)h->ws)->x; break;
    case K_GMRES: src = ((ko_gmres_workspace *)h->ws)->x; break;
    case K_BICGSTAB: src = ((ko_bicgstab_workspace *)h->ws)->x; break;
    default: return -1;
  }
  memcpy(x, src, sizeof(double) * (size_t)n);
  return 0;
}

int krylov_get_y(void *ws, void *y, int m) {
  (void)y; (void)m;
  return find(ws) ? -2 : -1;
}

int krylov_is_solved(void *ws) { handle *h = find(ws); return h ? stats_of(h)->solved : -1; }
int krylov_niter(void *ws) { handle *h = find(ws); return h ? stats_of(h)->niter : -1; }
double krylov_elapsed_time(void *ws) { handle *h = find(ws); return h ? stats_of(h)->timer : -1.0; }

int krylov_warm_start(void *ws, const void *x0, int n) {
  handle *h = find(ws);
  if (!h || n != h->n) return -1;
  switch (h->kind) {
    case K_CG: ko_cg_warm_start((ko_cg_workspace *)h->ws, (const double *)x0); return 0;
    case K_GMRES: ko_gmres_warm_start((ko_gmres_workspace *)h->ws, (const double *)x0); return 0;
    case K_BICGSTAB: ko_bicgstab_warm_start((ko_bicgstab_workspace *)h->ws, (const double *)x0); return 0;
    default: return -1;
  }
}

int krylov_warm_start2(void *ws, const void *x0, const void *y0, int nx, int ny) {
  (void)x0; (void)y0; (void)nx; (void)ny;
  return find(ws) ? -2 : -1;
}

static int release(void *ws, int want_block) {
  handle **pp = &g_handles;
  while (*pp) {
    handle *h = *pp;
    if (h == (handle *)ws && ((h->kind == K_BLOCK_GMRES) == want_block)) {
      *pp = h->next;
      switch (h->kind) {
        case K_CG: ko_cg_workspace_free((ko_cg_workspace *)h->ws); break;
        case K_GMRES: ko_gmres_workspace_free((ko_gmres_workspace *)h->ws); break;
        case K_BICGSTAB: ko_bicgstab_workspace_free((ko_bicgstab_workspace *)h->ws); break;
        default: ko_block_gmres_workspace_free((ko_block_gmres_workspace *)h->ws); break;
      }
      free(h);
      return 0;
    }
    pp = &h->next;
  }
  return 1;
}

int krylov_workspace_free(void *ws) { return release(ws, 0); }

/* ---- block interface ----------------------------------------------------- */

int krylov_block_workspace_create(KrylovBlockSolverType solver, int m, int n, int p,
                                  KrylovDataType dtype, KrylovDeviceType device,
                                  const KrylovWorkspaceOptions *wopts, void **ws_out) {
  (void)device;
  if (dtype != KRYLOV_FLOAT64 || (int)solver != KRYLOV_BLOCK_GMRES) return -2;
  int memory = (wopts && wopts->memory > 0) ? wopts->memory : 5;
  handle *h = (handle *)calloc(1, sizeof(*h));
  h->kind = K_BLOCK_GMRES; h->m = m; h->n = n; h->p = p;
  h->ws = ko_block_gmres_workspace_create(m, n, p, memory);
  h->next = g_handles; g_handles = h;
  *ws_out = h;
  return 0;
}

typedef struct { KrylovBlockMatvec A, M, N; void *ud; } bcb_t;
static void bcall_A(const double *X, double *Y, int p, void *c) { ((bcb_t *)c)->A(X, Y, p, ((bcb_t *)c)->ud); }
static void bcall_M(const double *X, double *Y, int p, void *c) { ((bcb_t *)c)->M(X, Y, p, ((bcb_t *)c)->ud); }
static void bcall_N(const double *X, double *Y, int p, void *c) { ((bcb_t *)c)->N(X, Y, p, ((bcb_t *)c)->ud); }

int krylov_block_solve(void *ws, KrylovBlockMatvec matvec_A, KrylovBlockMatvec matvec_M,
                       KrylovBlockMatvec matvec_N, const void *B, void *userdata,
                       const KrylovOptions *opts) {
  handle *h = find(ws);
  if (!h || h->kind != K_BLOCK_GMRES || !matvec_A) return -1;
  bcb_t cb = {matvec_A, matvec_M, matvec_N, userdata};
  ko_options o = map_opts(opts);
  return ko_block_gmres((ko_block_gmres_workspace *)h->ws, bcall_A, matvec_M ? bcall_M : NULL,
                        matvec_N ? bcall_N : NULL, &cb, (const double *)B, &o);
}

int krylov_block_get_X(void *ws, void *X, int n, int p) {
  handle *h = find(ws);
  if (!h || h->kind != K_BLOCK_GMRES) return -1;
  memcpy(X, ((ko_block_gmres_workspace *)h->ws)->X, sizeof(double) * (size_t)n * (size_t)p);
  return 0;
}

int krylov_block_is_solved(void *ws) {
  handle *h = find(ws);
  return (h && h->kind == K_BLOCK_GMRES) ? stats_of(h)->solved : -1;
}
int krylov_block_niter(void *ws) {
  handle *h = find(ws);
  return (h && h->kind == K_BLOCK_GMRES) ? stats_of(h)->niter : -1;
}
double krylov_block_elapsed_time(void *ws) {
  handle *h = find(ws);
  return (h && h->kind == K_BLOCK_GMRES) ? stats_of(h)->timer : -1.0;
}
int krylov_block_warm_start(void *ws, const void *x0, int n, int p) {
  handle *h = find(ws);
  if (!h || h->kind != K_BLOCK_GMRES || n != h->n || p != h->p) return -1;
  ko_block_gmres_warm_start((ko_block_gmres_workspace *)h->ws, (const double *)x0);
  return 0;
}
int krylov_block_workspace_free(void *ws) { return release(ws, 1); }
