// capi_compat.cpp -- libkrylov_hip_capi.so: the reference's C / Fortran interface (interfaces/include/krylov.h,
// krylov.f90) served by libkrylov_hip.so, plus a device enumerator.
//
// The reference's `KrylovDeviceType` has the single value KRYLOV_CPU (krylov.h:44-46; "GPU devices" are listed as
// not yet exposed, docs/src/interfaces/overview.md:4).  This library keeps every entry point of that header and
//   * device == KRYLOV_CPU : the client's arrays and callbacks live on the HOST, exactly as with libkrylov.so --
//     every operator application copies x down, calls the callback and copies y up.  That is what lets the
//     reference's own C and Fortran clients (interfaces/test/C/*.c, interfaces/test/Fortran/*.f90,
//     interfaces/examples/C/*.c), compiled from where they lie, run on the GPU (tests/test_gpu_refclients.py):
//     conformance, not speed;
//   * device == KRYLOV_HIP (include/krylov_hip_ext.h) : b, c, x0, the vectors handed to the callbacks and the
//     buffer of krylov_get_x are DEVICE pointers, nothing is copied; krylov_hip_set_csr attaches a CSR operator
//     resident in HBM, after which matvec_A may be NULL and the solve runs the fused / device-resident loops.
// krylov.h itself is not redistributed here: it is taken from the reference tree (or from an installed libkrylov)
// at build time (-I), so this component is built only where that header is available.
#include <cmath>
#include <cstring>
#include <vector>

#include "krylov.h"           // the reference's header (-I <reference>/interfaces/include)
#include "krylov_hip.h"
#include "krylov_hip_ext.h"

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
  bool device = false;        // KRYLOV_HIP: client pointers are device pointers
  khip_csr *csr = nullptr;    // built-in operator (krylov_hip_set_csr), owned
  int fused = 2;
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

int apply_vec_dev(void *self, const double *x, double *y) {      // KRYLOV_HIP: the callback works on device pointers
  VecCb *c = static_cast<VecCb *>(self);
  c->f(x, y, c->ud);
  return 0;
}

struct BlockCb {             // host callback on column-major n x p blocks wrapped for row-major device panels
  KrylovBlockMatvec f;
  void *ud;
  int n, p;
  double *dcol;              // device scratch, n * p column-major
  std::vector<double> hx, hy;
  double *dcol2 = nullptr;   // KRYLOV_HIP: output of the callback (column-major, device)
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

int apply_block_dev(void *self, const double *X, double *Y);   // below

int apply_block_dev(void *self, const double *X, double *Y) {   // KRYLOV_HIP: column-major DEVICE blocks to the callback
  BlockCb *c = static_cast<BlockCb *>(self);
  if (khip_panel_to_colmajor(ctx(), c->n, c->p, X, c->dcol)) return 1;
  c->f(c->dcol, c->dcol2, c->p, c->ud);
  return khip_panel_from_colmajor(ctx(), c->n, c->p, c->dcol2, Y);
}

khip_options map_opts(const KrylovOptions *o) {
  khip_options k = khip_default_options();
  if (!o) return k;
  k.atol = o->atol; k.rtol = o->rtol; k.itmax = o->itmax; k.timemax = o->timemax;
  k.radius = o->radius; k.linesearch = o->linesearch; k.restart = o->restart;
  k.reorthogonalization = o->reorthogonalization;
  k.verbose = o->verbose;                       // interfaces/include/krylov.h:145: the reference's per-iteration log
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
      if (h->csr) khip_csr_destroy(h->csr);
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

int krylov_workspace_create(KrylovSolverType solver, int m, int n, KrylovDataType dtype, KrylovDeviceType device,
                            const KrylovWorkspaceOptions *wopts, void **ws_out) {
  if (dtype != KRYLOV_FLOAT64) return -2;
  if ((int)device != (int)KRYLOV_CPU && (int)device != (int)KRYLOV_HIP) return -1;
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
  h->device = (int)device == (int)KRYLOV_HIP;
  h->next = g_handles; g_handles = h;
  *ws_out = h;
  return 0;
}

int krylov_solve(void *ws, KrylovMatvec matvec_A, KrylovMatvec, KrylovMatvec matvec_M, KrylovMatvec matvec_N,
                 const void *b, const void *c, void *userdata, const KrylovOptions *opts) {
  Handle *h = find(ws);
  if (!h || h->kind == K_BLOCK_GMRES || !b) return -1;
  if (!matvec_A && !h->csr) return -1;                       // no callback and no built-in operator
  const int n = h->n;
  const size_t hn = h->device ? 0 : (size_t)n;               // host staging only in KRYLOV_CPU mode
  VecCb cbA{matvec_A, userdata, n, std::vector<double>(hn), std::vector<double>(hn)};
  VecCb cbM{matvec_M, userdata, n, std::vector<double>(hn), std::vector<double>(hn)};
  VecCb cbN{matvec_N, userdata, n, std::vector<double>(hn), std::vector<double>(hn)};
  khip_apply_fn fn = h->device ? apply_vec_dev : apply_vec;
  khip_operator A{nullptr, fn, &cbA}, M{nullptr, fn, &cbM}, N{nullptr, fn, &cbN};
  if (!matvec_A) { A.csr = h->csr; A.apply = nullptr; A.self = nullptr; }
  double *db = nullptr, *dc = nullptr;
  if (h->device) {
    db = const_cast<double *>(static_cast<const double *>(b));
    dc = const_cast<double *>(static_cast<const double *>(c));
  } else {
    if (khip_malloc(ctx(), sizeof(double) * (n + 2), reinterpret_cast<void **>(&db))) return -1;
    khip_memcpy_h2d(ctx(), db, b, sizeof(double) * n);
    if (c) {
      khip_malloc(ctx(), sizeof(double) * (n + 2), reinterpret_cast<void **>(&dc));
      khip_memcpy_h2d(ctx(), dc, c, sizeof(double) * n);
    }
  }
  khip_options o = map_opts(opts);
  o.fused = h->fused;
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
  if (!h->device) {
    khip_free(ctx(), db);
    khip_free(ctx(), dc);
  }
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
  if (h->device) return khip_memcpy_d2d(ctx(), x, src, sizeof(double) * n) ? -1 : 0;
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
  if (h->device) {
    d = const_cast<double *>(static_cast<const double *>(x0));
  } else {
    if (khip_malloc(ctx(), sizeof(double) * (n + 2), reinterpret_cast<void **>(&d))) return -1;
    khip_memcpy_h2d(ctx(), d, x0, sizeof(double) * n);
  }
  int rc;
  switch (h->kind) {
    case K_CG: rc = khip_cg_warm_start(static_cast<khip_cg_workspace *>(h->ws), d); break;
    case K_GMRES: rc = khip_gmres_warm_start(static_cast<khip_gmres_workspace *>(h->ws), d); break;
    default: rc = khip_bicgstab_warm_start(static_cast<khip_bicgstab_workspace *>(h->ws), d); break;
  }
  khip_ctx_sync(ctx());
  if (!h->device) khip_free(ctx(), d);
  return rc ? -1 : 0;
}
int krylov_warm_start2(void *ws, const void *, const void *, int, int) { return find(ws) ? -2 : -1; }
int krylov_workspace_free(void *ws) { return release(ws, false); }

// ---- block interface ---------------------------------------------------------------------------
int krylov_block_workspace_create(KrylovBlockSolverType solver, int m, int n, int p, KrylovDataType dtype,
                                  KrylovDeviceType device, const KrylovWorkspaceOptions *wopts, void **ws_out) {
  if (dtype != KRYLOV_FLOAT64 || (int)solver != KRYLOV_BLOCK_GMRES) return -2;
  if ((int)device != (int)KRYLOV_CPU && (int)device != (int)KRYLOV_HIP) return -1;
  if (!ctx()) return -1;
  const int memory = (wopts && wopts->memory > 0) ? wopts->memory : 5;
  khip_block_gmres_workspace *w = nullptr;
  if (khip_block_gmres_workspace_create(ctx(), m, n, p, memory, &w)) return -1;
  Handle *h = new Handle{K_BLOCK_GMRES, m, n, p, w, g_handles};
  h->device = (int)device == (int)KRYLOV_HIP;
  g_handles = h;
  *ws_out = h;
  return 0;
}

int krylov_block_solve(void *ws, KrylovBlockMatvec matvec_A, KrylovBlockMatvec matvec_M, KrylovBlockMatvec matvec_N,
                       const void *B, void *userdata, const KrylovOptions *opts) {
  Handle *h = find(ws);
  if (!h || h->kind != K_BLOCK_GMRES || !B) return -1;
  if (!matvec_A && !h->csr) return -1;
  const int n = h->n, p = h->p;
  const size_t cnt = (size_t)n * p;
  const size_t hcnt = h->device ? 0 : cnt;
  double *dcol[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, *dB = nullptr;
  // column-major device scratch: one buffer per callback (two in device mode: input and output of the callback)
  for (int i = 0; i < (h->device ? 6 : 3); ++i) khip_malloc(ctx(), sizeof(double) * (cnt + 2), reinterpret_cast<void **>(&dcol[i]));
  if (h->device) {
    dB = const_cast<double *>(static_cast<const double *>(B));
  } else {
    khip_malloc(ctx(), sizeof(double) * (cnt + 2), reinterpret_cast<void **>(&dB));
    khip_memcpy_h2d(ctx(), dB, B, sizeof(double) * cnt);
  }
  BlockCb cbA{matvec_A, userdata, n, p, dcol[0], std::vector<double>(hcnt), std::vector<double>(hcnt), dcol[3]};
  BlockCb cbM{matvec_M, userdata, n, p, dcol[1], std::vector<double>(hcnt), std::vector<double>(hcnt), dcol[4]};
  BlockCb cbN{matvec_N, userdata, n, p, dcol[2], std::vector<double>(hcnt), std::vector<double>(hcnt), dcol[5]};
  khip_apply_fn fn = h->device ? apply_block_dev : apply_block;
  khip_operator A{nullptr, fn, &cbA}, M{nullptr, fn, &cbM}, N{nullptr, fn, &cbN};
  if (!matvec_A) { A.csr = h->csr; A.apply = nullptr; A.self = nullptr; }
  khip_options o = map_opts(opts);
  const int rc = khip_block_gmres_solve(static_cast<khip_block_gmres_workspace *>(h->ws), &A, matvec_M ? &M : nullptr,
                                        matvec_N ? &N : nullptr, dB, &o);
  for (int i = 0; i < 6; ++i) khip_free(ctx(), dcol[i]);
  if (!h->device) khip_free(ctx(), dB);
  return rc == 0 ? 0 : -1;
}

int krylov_block_get_X(void *ws, void *X, int n, int p) {
  Handle *h = find(ws);
  if (!h || h->kind != K_BLOCK_GMRES) return -1;
  double *d = nullptr;
  const size_t cnt = (size_t)n * p;
  if (khip_malloc(ctx(), sizeof(double) * (cnt + 2), reinterpret_cast<void **>(&d))) return -1;
  int rc = khip_block_gmres_get_X(static_cast<khip_block_gmres_workspace *>(h->ws), d);
  if (!rc) rc = h->device ? khip_memcpy_d2d(ctx(), X, d, sizeof(double) * cnt) : khip_memcpy_d2h(ctx(), X, d, sizeof(double) * cnt);
  if (!rc) rc = khip_ctx_sync(ctx());
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
  if (h->device) {
    d = const_cast<double *>(static_cast<const double *>(x0));
  } else {
    if (khip_malloc(ctx(), sizeof(double) * (cnt + 2), reinterpret_cast<void **>(&d))) return -1;
    khip_memcpy_h2d(ctx(), d, x0, sizeof(double) * cnt);
  }
  const int rc = khip_block_gmres_warm_start(static_cast<khip_block_gmres_workspace *>(h->ws), d);
  khip_ctx_sync(ctx());
  if (!h->device) khip_free(ctx(), d);
  return rc ? -1 : 0;
}
int krylov_block_workspace_free(void *ws) { return release(ws, true); }

// ---- extension (include/krylov_hip_ext.h) ------------------------------------------------------
void *krylov_hip_context(void) { return ctx(); }
void *krylov_hip_stream(void) { return ctx() ? khip_ctx_stream(ctx()) : nullptr; }

int krylov_hip_set_csr(void *ws, long long nnz, const void *rowptr, int rowptr_bits, const int *col, const double *val,
                       int index_base, int on_device) {
  Handle *h = find(ws);
  if (!h || !ctx()) return -1;
  if (h->csr) { khip_csr_destroy(h->csr); h->csr = nullptr; }
  return khip_csr_create(ctx(), h->m, h->n, nnz, rowptr, rowptr_bits, col, val, index_base, on_device, &h->csr) ? -1 : 0;
}

int krylov_hip_set_fused(void *ws, int level) {
  Handle *h = find(ws);
  if (!h || level < 0 || level > 2) return -1;
  h->fused = level;
  return 0;
}

const char *krylov_hip_last_error(void) { return khip_last_error(); }

}  // extern "C"
