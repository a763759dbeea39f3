// ref_capi_hip_shim.cpp -- the reference's C ABI (interfaces/include/krylov.h) in front of libkrylov_hip.so.
//
// TEST INFRASTRUCTURE (built by `make ref` into oracle/_ref/, only where /root/reference exists; the header
// is included from there, never copied).  It lets the reference's own C clients -- interfaces/test/C/
// {test_api,test_all_solvers,test_block}.c and interfaces/examples/C/{basic_cg,block_gmres}.c, compiled from
// where they lie -- drive the HIP path on a GPU: the strongest conformance evidence available without Julia
// (SURVEY.md section 8c / 8f N3).  The clients pass HOST arrays and HOST callbacks (KRYLOV_CPU is the only
// device enumerator, krylov.h:44-46), so every operator application here copies x to the host, calls the
// client's callback and copies y back; that is a conformance harness, not a performance path.
#include <cmath>
#include <cstring>
#include <vector>

#include "krylov.h"       // /root/reference/interfaces/include (-I)
#include "krylov_hip.h"   // include/ of this repo (-I)

namespace {

enum Kind { K_CG, K_GMRES, K_BICGSTAB, K_BLOCK_GMRES };

khip_ctx *g_ctx = nullptr;
khip_ctx *ctx() {
  if (!g_ctx && khip_ctx_create(0, nullptr, &g_ctx) != 0) g_ctx = nullptr;
  return g_ctx;
}

struct Handle {
  Kind kind;
  int m, n, p;
  void *ws;
  Handle *next;
};
Handle *g_handles = nullptr;
Handle *find(void *h) {
  for (Handle *it = g_handles; it; it = it->next)
    if (it == h) return it;
  return nullptr;
}

struct VecCb {               // host callback y = Op(x) wrapped as a device operator
  KrylovMatvec f;
  void *ud;
  int n;
  std::vector<double> hx, hy;
};
int apply_vec(void *self, const double *x, double *y) {
  VecCb *c = static_cast<VecCb *>(self);
  if (khip_memcpy_d2h(ctx(), c->hx.data(), x, sizeof(double) * c->n)) return 1;
  c->f(c->hx.data(), c->hy.data(), c->ud);
  return khip_memcpy_h2d(ctx(), y, c->hy.data(), sizeof(double) * c->n);
}

struct BlockCb {             // host callback on column-major n x p blocks wrapped for row-major device panels
  KrylovBlockMatvec f;
  void *ud;
  int n, p;
  double *dcol;              // device scratch, n * p column-major
  std::vector<double> hx, hy;
};
int apply_block(void *self, const double *X, double *Y) {
  BlockCb *c = static_cast<BlockCb *>(self);
  const size_t bytes = sizeof(double) * (size_t)c->n * c->p;
  if (khip_panel_to_colmajor(ctx(), c->n, c->p, X, c->dcol)) return 1;
  if (khip_memcpy_d2h(ctx(), c->hx.data(), c->dcol, bytes)) return 1;
  c->f(c->hx.data(), c->hy.data(), c->p, c->ud);
  if (khip_memcpy_h2d(ctx(), c->dcol, c->hy.data(), bytes)) return 1;
  return khip_panel_from_colmajor(ctx(), c->n, c->p, c->dcol, Y);
}

khip_options map_opts(const KrylovOptions *o) {
  khip_options k = khip_default_options();
  if (!o) return k;
  k.atol = o->atol; k.rtol = o->rtol; k.itmax = o->itmax; k.timemax = o->timemax;
  k.radius = o->radius; k.linesearch = o->linesearch; k.restart = o->restart;
  k.reorthogonalization = o->reorthogonalization;
  return k;
}

const khip_stats *stats_of(Handle *h) {
  switch (h->kind) {
    case K_CG: return khip_cg_stats(static_cast<khip_cg_workspace *>(h->ws));
    case K_GMRES: return khip_gmres_stats(static_cast<khip_gmres_workspace *>(h->ws));
    case K_BICGSTAB: return khip_bicgstab_stats(static_cast<khip_bicgstab_workspace *>(h->ws));
    default: return khip_block_gmres_stats(static_cast<khip_block_gmres_workspace *>(h->ws));
  }
}

int release(void *ws, bool want_block) {
  Handle **pp = &g_handles;
  while (*pp) {
    Handle *h = *pp;
    if (h == ws && ((h->kind == K_BLOCK_GMRES) == want_block)) {
      *pp = h->next;
      switch (h->kind) {
        case K_CG: khip_cg_workspace_destroy(static_cast<khip_cg_workspace *>(h->ws)); break;
        case K_GMRES: khip_gmres_workspace_destroy(static_cast<khip_gmres_workspace *>(h->ws)); break;
        case K_BICGSTAB: khip_bicgstab_workspace_destroy(static_cast<khip_bicgstab_workspace *>(h->ws)); break;
        default: khip_block_gmres_workspace_destroy(static_cast<khip_block_gmres_workspace *>(h->ws)); break;
      }
      delete h;
      return 0;
    }
    pp = &h->next;
  }
  return 1;
}

}  // namespace

extern "C" {

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

int krylov_workspace_create(KrylovSolverType solver, int m, int n, KrylovDataType dtype, KrylovDeviceType,
                            const KrylovWorkspaceOptions *wopts, void **ws_out) {
  if (dtype != KRYLOV_FLOAT64) return -2;
  if ((int)solver != KRYLOV_CG && (int)solver != KRYLOV_GMRES && (int)solver != KRYLOV_BICGSTAB) return -2;
  if (!ctx()) return -1;
  const int memory = (wopts && wopts->memory > 0) ? wopts->memory : 20;
  Handle *h = new Handle{K_CG, m, n, 0, nullptr, nullptr};
  int rc = 0;
  if ((int)solver == KRYLOV_CG) {
    khip_cg_workspace *w = nullptr; rc = khip_cg_workspace_create(ctx(), m, n, &w); h->kind = K_CG; h->ws = w;
  } else if ((int)solver == KRYLOV_GMRES) {
    khip_gmres_workspace *w = nullptr; rc = khip_gmres_workspace_create(ctx(), m, n, memory, &w); h->kind = K_GMRES; h->ws = w;
  } else {
    khip_bicgstab_workspace *w = nullptr; rc = khip_bicgstab_workspace_create(ctx(), m, n, &w); h->kind = K_BICGSTAB; h->ws = w;
  }
  if (rc) { delete h; return -1; }
  h->next = g_handles; g_handles = h;
  *ws_out = h;
  return 0;
}

int krylov_solve(void *ws, KrylovMatvec matvec_A, KrylovMatvec, KrylovMatvec matvec_M, KrylovMatvec matvec_N,
                 const void *b, const void *c, void *userdata, const KrylovOptions *opts) {
  Handle *h = find(ws);
  if (!h || !matvec_A || h->kind == K_BLOCK_GMRES) return -1;
  const int n = h->n;
  VecCb cbA{matvec_A, userdata, n, std::vector<double>(n), std::vector<double>(n)};
  VecCb cbM{matvec_M, userdata, n, std::vector<double>(n), std::vector<double>(n)};
  VecCb cbN{matvec_N, userdata, n, std::vector<double>(n), std::vector<double>(n)};
  khip_operator A{nullptr, apply_vec, &cbA}, M{nullptr, apply_vec, &cbM}, N{nullptr, apply_vec, &cbN};
  double *db = nullptr, *dc = nullptr;
  if (khip_malloc(ctx(), sizeof(double) * (n + 2), reinterpret_cast<void **>(&db))) return -1;
  khip_memcpy_h2d(ctx(), db, b, sizeof(double) * n);
  if (c) {
    khip_malloc(ctx(), sizeof(double) * (n + 2), reinterpret_cast<void **>(&dc));
    khip_memcpy_h2d(ctx(), dc, c, sizeof(double) * n);
  }
  khip_options o = map_opts(opts);
  int rc;
  switch (h->kind) {
    case K_CG: rc = khip_cg_solve(static_cast<khip_cg_workspace *>(h->ws), &A, matvec_M ? &M : nullptr, db, &o); break;
    case K_GMRES:
      rc = khip_gmres_solve(static_cast<khip_gmres_workspace *>(h->ws), &A, matvec_M ? &M : nullptr, matvec_N ? &N : nullptr, db, &o);
      break;
    default:
      rc = khip_bicgstab_solve(static_cast<khip_bicgstab_workspace *>(h->ws), &A, matvec_M ? &M : nullptr,
                               matvec_N ? &N : nullptr, db, dc, &o);
      break;
  }
  khip_free(ctx(), db);
  khip_free(ctx(), dc);
  return rc == 0 ? 0 : -1;
}

int krylov_get_x(void *ws, void *x, int n) {
  Handle *h = find(ws);
  if (!h) return -1;
  double *src = nullptr;
  switch (h->kind) {
    case K_CG: src = khip_cg_solution(static_cast<khip_cg_workspace *>(h->ws)); break;
    case K_GMRES: src = khip_gmres_solution(static_cast<khip_gmres_workspace *>(h->ws)); break;
    case K_BICGSTAB: src = khip_bicgstab_solution(static_cast<khip_bicgstab_workspace *>(h->ws)); break;
    default: return -1;
  }
  return khip_memcpy_d2h(ctx(), x, src, sizeof(double) * n) ? -1 : 0;
}
int krylov_get_y(void *ws, void *, int) { return find(ws) ? -2 : -1; }
int krylov_is_solved(void *ws) { Handle *h = find(ws); return h ? stats_of(h)->solved : -1; }
int krylov_niter(void *ws) { Handle *h = find(ws); return h ? stats_of(h)->niter : -1; }
double krylov_elapsed_time(void *ws) { Handle *h = find(ws); return h ? stats_of(h)->timer : -1.0; }

int krylov_warm_start(void *ws, const void *x0, int n) {
  Handle *h = find(ws);
  if (!h || n != h->n || h->kind == K_BLOCK_GMRES) return -1;
  double *d = nullptr;
  if (khip_malloc(ctx(), sizeof(double) * (n + 2), reinterpret_cast<void **>(&d))) return -1;
  khip_memcpy_h2d(ctx(), d, x0, sizeof(double) * n);
  int rc;
  switch (h->kind) {
    case K_CG: rc = khip_cg_warm_start(static_cast<khip_cg_workspace *>(h->ws), d); break;
    case K_GMRES: rc = khip_gmres_warm_start(static_cast<khip_gmres_workspace *>(h->ws), d); break;
    default: rc = khip_bicgstab_warm_start(static_cast<khip_bicgstab_workspace *>(h->ws), d); break;
  }
  khip_ctx_sync(ctx());
  khip_free(ctx(), d);
  return rc ? -1 : 0;
}
int krylov_warm_start2(void *ws, const void *, const void *, int, int) { return find(ws) ? -2 : -1; }
int krylov_workspace_free(void *ws) { return release(ws, false); }

// ---- block interface ---------------------------------------------------------------------------
int krylov_block_workspace_create(KrylovBlockSolverType solver, int m, int n, int p, KrylovDataType dtype,
                                  KrylovDeviceType, const KrylovWorkspaceOptions *wopts, void **ws_out) {
  if (dtype != KRYLOV_FLOAT64 || (int)solver != KRYLOV_BLOCK_GMRES) return -2;
  if (!ctx()) return -1;
  const int memory = (wopts && wopts->memory > 0) ? wopts->memory : 5;
  khip_block_gmres_workspace *w = nullptr;
  if (khip_block_gmres_workspace_create(ctx(), m, n, p, memory, &w)) return -1;
  Handle *h = new Handle{K_BLOCK_GMRES, m, n, p, w, g_handles};
  g_handles = h;
  *ws_out = h;
  return 0;
}

int krylov_block_solve(void *ws, KrylovBlockMatvec matvec_A, KrylovBlockMatvec matvec_M, KrylovBlockMatvec matvec_N,
                       const void *B, void *userdata, const KrylovOptions *opts) {
  Handle *h = find(ws);
  if (!h || h->kind != K_BLOCK_GMRES || !matvec_A) return -1;
  const int n = h->n, p = h->p;
  const size_t cnt = (size_t)n * p;
  double *dcol[3] = {nullptr, nullptr, nullptr}, *dB = nullptr;
  for (int i = 0; i < 3; ++i) khip_malloc(ctx(), sizeof(double) * (cnt + 2), reinterpret_cast<void **>(&dcol[i]));
  khip_malloc(ctx(), sizeof(double) * (cnt + 2), reinterpret_cast<void **>(&dB));
  khip_memcpy_h2d(ctx(), dB, B, sizeof(double) * cnt);
  BlockCb cbA{matvec_A, userdata, n, p, dcol[0], std::vector<double>(cnt), std::vector<double>(cnt)};
  BlockCb cbM{matvec_M, userdata, n, p, dcol[1], std::vector<double>(cnt), std::vector<double>(cnt)};
  BlockCb cbN{matvec_N, userdata, n, p, dcol[2], std::vector<double>(cnt), std::vector<double>(cnt)};
  khip_operator A{nullptr, apply_block, &cbA}, M{nullptr, apply_block, &cbM}, N{nullptr, apply_block, &cbN};
  khip_options o = map_opts(opts);
  const int rc = khip_block_gmres_solve(static_cast<khip_block_gmres_workspace *>(h->ws), &A, matvec_M ? &M : nullptr,
                                        matvec_N ? &N : nullptr, dB, &o);
  for (int i = 0; i < 3; ++i) khip_free(ctx(), dcol[i]);
  khip_free(ctx(), dB);
  return rc == 0 ? 0 : -1;
}

int krylov_block_get_X(void *ws, void *X, int n, int p) {
  Handle *h = find(ws);
  if (!h || h->kind != K_BLOCK_GMRES) return -1;
  double *d = nullptr;
  const size_t cnt = (size_t)n * p;
  if (khip_malloc(ctx(), sizeof(double) * (cnt + 2), reinterpret_cast<void **>(&d))) return -1;
  int rc = khip_block_gmres_get_X(static_cast<khip_block_gmres_workspace *>(h->ws), d);
  if (!rc) rc = khip_memcpy_d2h(ctx(), X, d, sizeof(double) * cnt);
  khip_free(ctx(), d);
  return rc ? -1 : 0;
}
int krylov_block_is_solved(void *ws) { Handle *h = find(ws); return (h && h->kind == K_BLOCK_GMRES) ? stats_of(h)->solved : -1; }
int krylov_block_niter(void *ws) { Handle *h = find(ws); return (h && h->kind == K_BLOCK_GMRES) ? stats_of(h)->niter : -1; }
double krylov_block_elapsed_time(void *ws) { Handle *h = find(ws); return (h && h->kind == K_BLOCK_GMRES) ? stats_of(h)->timer : -1.0; }
int krylov_block_warm_start(void *ws, const void *x0, int n, int p) {
  Handle *h = find(ws);
  if (!h || h->kind != K_BLOCK_GMRES || n != h->n || p != h->p) return -1;
  double *d = nullptr;
  const size_t cnt = (size_t)n * p;
  if (khip_malloc(ctx(), sizeof(double) * (cnt + 2), reinterpret_cast<void **>(&d))) return -1;
  khip_memcpy_h2d(ctx(), d, x0, sizeof(double) * cnt);
  const int rc = khip_block_gmres_warm_start(static_cast<khip_block_gmres_workspace *>(h->ws), d);
  khip_ctx_sync(ctx());
  khip_free(ctx(), d);
  return rc ? -1 : 0;
}
int krylov_block_workspace_free(void *ws) { return release(ws, true); }

}  // extern "C"
