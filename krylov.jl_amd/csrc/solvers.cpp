// solvers.cpp -- host control flow of cg!, gmres!, bicgstab! above the device primitives.
//
// With Julia present the UNMODIFIED reference solvers (src/cg.jl, src/gmres.jl, src/bicgstab.jl)
// drive the k* entry points through the glue of INTEGRATION.md.  Julia is absent in this build
// environment, so the same control flow is restated here in C++ -- scalar recurrences, stopping
// tests, status strings and workspace aliasing follow the cited lines -- and only ever touches
// device memory through the khip_* primitives of include/krylov_hip.h.
//
// opts.fused = 0 issues exactly the primitive sequence of the reference (one launch per k* call,
// one host sync per kdot/knorm).  opts.fused = 1 replaces legal groups by fused kernels
// (SpMV+dot, axpy+axpy+dot, copy+axpy, MGS cascade with device-resident coefficients, multi-axpy);
// elementwise results are bit-identical, reductions agree to one ulp.  opts.fused = 2 (cg!
// only so far) additionally keeps the scalar recurrences and stopping tests on the device
// (solver_device.hpp): no host round trip inside the loop, iterations are enqueued ahead.
#include <chrono>
#include <cmath>
#include <limits>

#include "khip_internal.hpp"
#include "solver_device.hpp"

using namespace khip;

namespace {

constexpr double kEps = std::numeric_limits<double>::epsilon();

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
double tol_or_default(double t) { return std::isnan(t) ? std::sqrt(kEps) : t; }
double timemax_of(const khip_options &o) {
  return (std::isnan(o.timemax) || o.timemax <= 0) ? std::numeric_limits<double>::infinity() : o.timemax;
}

struct StatsBox {
  khip_stats st;
  std::vector<double> residuals;
  int path = -1;       // which loop the last solve ran (khip_*_last_path): 2 device-resident / look-ahead, 1 host-driven fused, 0 primitive sequence
  StatsBox() {
    memset(&st, 0, sizeof(st));
    snprintf(st.status, sizeof(st.status), "unknown");
  }
  void reset() {   // reset!(stats), src/krylov_stats.jl:38-44
    residuals.clear();
    st.residuals = nullptr;
    st.nres = 0;
    st.indefinite = 0;
    st.npcCount = 0;
    st.error[0] = 0;
  }
  void push(double v) { residuals.push_back(v); }
  void publish() {
    st.residuals = residuals.empty() ? nullptr : residuals.data();
    st.nres = (int)residuals.size();
  }
  int fail(int code, const char *msg) {
    snprintf(st.error, sizeof(st.error), "%s", msg);
    set_error("%s", msg);
    publish();
    return code;
  }
  int fail_rc(int rc) {   // a primitive failed: message already in khip_last_error()
    snprintf(st.error, sizeof(st.error), "%s", khip_last_error());
    publish();
    return rc;
  }
};

int apply_op(khip_ctx *ctx, const khip_operator *op, const double *x, double *y) {
  if (op->apply) {
    int rc = op->apply(op->self, x, y);
    if (rc != 0) { set_error("user operator returned %d", rc); return KHIP_ERR_INVALID; }
    return KHIP_OK;
  }
  if (!op->csr) { set_error("operator has neither a CSR handle nor an apply callback"); return KHIP_ERR_INVALID; }
  return khip_spmv(ctx, op->csr, x, y);
}

int64_t padded(int64_t n) { return (n + 31) & ~(int64_t)31; }   // 256-byte multiples keep every slice 16-B aligned

// stats.allocation_timer (src/krylov_utils.jl:281-288, allocate_if): every vector allocation of a workspace -- at its
// creation and the lazy ones inside later solves -- adds its wall time to this accumulator; the creation / solve entry
// points move it into the workspace's stats (take_alloc_seconds).
static thread_local double g_alloc_seconds = 0.0;
int alloc_vec(khip_ctx *ctx, int64_t n, double **out) {
  const double t = now_s();
  const int rc = khip_malloc(ctx, sizeof(double) * (size_t)padded(n > 0 ? n : 1), reinterpret_cast<void **>(out));
  g_alloc_seconds += now_s() - t;
  return rc;
}
double take_alloc_seconds() { const double v = g_alloc_seconds; g_alloc_seconds = 0.0; return v; }

// Caller-owned work vectors (khip_*_workspace_adopt, khip_*_workspace_adopt_vector): the workspace of the reference owns
// its vectors on the Julia side (src/krylov_workspaces.jl:236-291), so a binding hands their device pointers over and the
// library must neither free nor replace them.  One list per workspace; everything not in it was allocated here.
struct Borrowed {
  std::vector<const double *> v;
  bool has(const double *p) const {
    for (const double *q : v) if (q == p) return true;
    return false;
  }
  void add(const double *p) { if (p && !has(p)) v.push_back(p); }
  void drop(const double *p) {
    for (size_t i = 0; i < v.size(); ++i) if (v[i] == p) { v.erase(v.begin() + (long)i); return; }
  }
};
void free_unless_borrowed(khip_ctx *ctx, const Borrowed &b, double *p) {
  if (p && !b.has(p)) khip_free(ctx, p);
}
// slot <- ptr as a caller-owned vector (ptr == nullptr empties the slot); what the library had allocated there is freed
// (`Borrowed` is a set of pointers, so a pointer may sit in ONE slot only -- the callers below refuse a pointer that
// already is another vector of the workspace; handing a slot the pointer it already holds changes nothing, in particular not
// who owns it.)
void adopt_into(khip_ctx *ctx, Borrowed &b, double **slot, double *ptr) {
  if (*slot == ptr) return;
  if (*slot) { if (b.has(*slot)) b.drop(*slot); else khip_free(ctx, *slot); }
  *slot = ptr;
  b.add(ptr);
}
// ptr (non-null) already is a vector of the workspace other than `self`: the name of that slot, else nullptr
template <class Tab>
const char *held_elsewhere(const Tab &tab, const double *const *self, const double *ptr, const std::vector<double *> *basis = nullptr) {
  if (!ptr) return nullptr;
  for (const auto &e : tab) if (e.slot != self && *e.slot == ptr) return e.k;
  if (basis) for (const double *v : *basis) if (v == ptr) return "V";
  return nullptr;
}

// ---- options.verbose: the reference's per-iteration log (kdisplay, src/krylov_utils.jl:301) on stdout.  Column headers are
// padded by hand: the labels are UTF-8 and printf pads bytes, Julia pads characters.
inline bool kdisplay(int64_t iter, int verbose) { return verbose > 0 && iter % verbose == 0; }

#define K(expr)                                   \
  do {                                            \
    int rc_k = (expr);                            \
    if (rc_k != KHIP_OK) return ws->box.fail_rc(rc_k); \
  } while (0)

}  // namespace

// ================================================================== CG ==========
struct khip_cg_workspace {
  khip_ctx *ctx;
  int64_t m, n;
  double *dx = nullptr, *x = nullptr, *r = nullptr, *npc_dir = nullptr, *p = nullptr, *Ap = nullptr, *z = nullptr;
  bool warm_start = false;
  Borrowed borrowed;                   // vectors of a caller's CgWorkspace (khip_cg_workspace_adopt)
  StatsBox box;
  // device-resident loop state (fused = 2), allocated on first use
  CgDevState *dev_state = nullptr;
  CgDevState *snap = nullptr;          // pinned host snapshots [2]
  CgcgDevState *cgcg_state = nullptr, *cgcg_snap = nullptr;   // single-reduction and pipelined variants
  double *pz = nullptr, *pq = nullptr;                        // pipelined variant: z = A s, q = A w (allocated on first use)
  double *hist_dev = nullptr;
  hipEvent_t snap_ev[2] = {nullptr, nullptr};
};

namespace {

constexpr int kDevChunk = 4;             // iterations enqueued between two snapshots of the device state
constexpr long long kHistWindowMax = 1 << 14;   // device history window (entries); ctx option "hist_window" shrinks it (tests)

// The loop of src/cg.jl:195-268 with scalars and stopping tests on the device.  Preconditions (checked by
// the caller): CSR operator, M = I, radius = 0, no linesearch, no callback.  On return the vectors are in
// the state the reference's loop leaves them in and `out` holds the final scalar state.
int cg_device_loop(khip_cg_workspace *ws, const khip_csr *A, double gamma, double eps_tol, int64_t itmax, bool history,
                   double t0, double timemax, CgDevState *out, bool *overtimed) {
  khip_ctx *ctx = ws->ctx;
  const int64_t n = ws->n;
  // every resource is guarded on its own: hist_dev and the events are shared with the single-reduction loop below
  if (!ws->dev_state) KHIP_CHECK_HIP(hipMalloc(&ws->dev_state, sizeof(CgDevState)));
  if (!ws->snap) KHIP_CHECK_HIP(hipHostMalloc(reinterpret_cast<void **>(&ws->snap), 2 * sizeof(CgDevState), hipHostMallocDefault));
  if (!ws->hist_dev) KHIP_CHECK_HIP(hipMalloc(&ws->hist_dev, sizeof(double) * (size_t)kHistWindowMax));
  for (auto &e : ws->snap_ev) if (!e) KHIP_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  long long kHistWindow = ctx->tune.hist_window;
  if (kHistWindow < kDevChunk) kHistWindow = kDevChunk;
  if (kHistWindow > kHistWindowMax) kHistWindow = kHistWindowMax;
  CgDevState *dev = ws->dev_state;
  CgDevState h;
  memset(&h, 0, sizeof(h));
  h.gamma = gamma; h.pNorm2 = gamma; h.eps_tol = eps_tol; h.keps = kEps; h.rNorm = std::sqrt(gamma);
  h.stop_seq = kSeqNever;
  h.hist = history ? ws->hist_dev : nullptr;
  h.hist_cap = kHistWindow;
  KHIP_CHECK_HIP(hipMemcpyAsync(dev, &h, sizeof(h), hipMemcpyHostToDevice, ctx->stream));
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));          // h is a stack object

  int64_t enq = 0;            // iterations enqueued
  long long hist_base = 0;
  std::vector<double> win;
  auto drain_history = [&](long long upto_iter) -> int {      // entries for iterations (hist_base, upto_iter]
    const long long cnt = upto_iter - hist_base;
    if (!history || cnt <= 0) return KHIP_OK;
    win.resize((size_t)cnt);
    KHIP_CHECK_HIP(hipMemcpy(win.data(), ws->hist_dev, sizeof(double) * (size_t)cnt, hipMemcpyDeviceToHost));
    for (double v : win) ws->box.push(v);
    return KHIP_OK;
  };
  int rc = KHIP_OK;
  bool stopped = false;
  for (int chunk = 0; !stopped; ++chunk) {
    const int64_t c = std::min<int64_t>(kDevChunk, itmax - enq);
    if (history && enq + c - hist_base > kHistWindow) {        // window full: empty it (rare: every 16384 iterations)
      KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
      CgDevState cur;
      KHIP_CHECK_HIP(hipMemcpy(&cur, dev, sizeof(cur), hipMemcpyDeviceToHost));
      if (cur.stop_seq != kSeqNever) break;
      if ((rc = drain_history(cur.iter)) != KHIP_OK) break;
      hist_base = cur.iter;
      KHIP_CHECK_HIP(hipMemcpy(&dev->hist_base, &hist_base, sizeof(hist_base), hipMemcpyHostToDevice));
    }
    for (int64_t i = 0; i < c && rc == KHIP_OK; ++i) {
      const long long j = (long long)(enq + i);
      ctx->ctl = SeqCtl{&dev->stop_seq, 3 * j, EPI_CG_STEP1, dev};
      const int s1 = take_slots(ctx, 1);
      rc = spmv_any(ctx, A, ws->p, ws->Ap, s1);                                     // :196-197, epilogue :198-213
      if (rc == KHIP_OK && ctx->comm) rc = comm_allreduce_dd_device(ctx, s1, 1);
      if (rc != KHIP_OK) break;
      ctx->ctl = SeqCtl{&dev->stop_seq, 3 * j + 1, EPI_CG_STEP2, dev};
      const int s2 = take_slots(ctx, 1);
      rc = launch_axpy_dev_dot(ctx, n, &dev->alpha, ws->Ap, ws->r, ws->r, s2);      // :240, :242, epilogue :243-262
      if (rc == KHIP_OK && ctx->comm) rc = comm_allreduce_dd_device(ctx, s2, 1);
      ctx->ctl = SeqCtl{};
      if (rc != KHIP_OK) break;
      rc = launch_cg_update_dev(ctx, n, dev, 3 * j + 2, ws->r, ws->p, ws->x);       // :239 and :259
    }
    ctx->ctl = SeqCtl{};
    if (rc != KHIP_OK) break;
    enq += c;
    const int b = chunk & 1;
    KHIP_CHECK_HIP(hipMemcpyAsync(&ws->snap[b], dev, sizeof(CgDevState), hipMemcpyDeviceToHost, ctx->stream));
    KHIP_CHECK_HIP(hipEventRecord(ws->snap_ev[b], ctx->stream));
    if (chunk >= 1) {                                        // look at the PREVIOUS chunk: the queue never runs dry
      KHIP_CHECK_HIP(hipEventSynchronize(ws->snap_ev[b ^ 1]));
      if (ws->snap[b ^ 1].stop_seq != kSeqNever) stopped = true;
    }
    if (enq >= itmax) stopped = true;
    if (!stopped && time_limit_reached(ctx, now_s() - t0, timemax)) { *overtimed = true; stopped = true; }
  }
  ctx->ctl = SeqCtl{};
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  if (rc != KHIP_OK) return rc;
  KHIP_CHECK_HIP(hipMemcpy(out, dev, sizeof(CgDevState), hipMemcpyDeviceToHost));
  return drain_history(out->iter);
}

// Single-reduction CG (options.variant = 1; Chronopoulos & Gear 1989): two passes per iteration --
//   update:  p = r + beta p ; s = w + beta s ; x += alpha p ; r -= alpha s            (72n bytes)
//   product: w = A r fused with BOTH dots (r.w, r.r) in one reduction; its epilogue forms beta, alpha, the tests
// -- i.e. ONE all-reduce per iteration on N GPUs instead of two, and 2 + 1 kernels instead of 3 + 2.  Not the
// reference's recurrence: same Krylov space, different rounding (tests/test_gpu_solvers.py holds its parity budget).
int cg_single_reduction_loop(khip_cg_workspace *ws, const khip_csr *A, double gamma0, double delta0, double eps_tol,
                             int64_t itmax, bool history, double t0, double timemax, CgcgDevState *out, bool *overtimed) {
  khip_ctx *ctx = ws->ctx;
  const int64_t n = ws->n;
  if (!ws->cgcg_state) {
    KHIP_CHECK_HIP(hipMalloc(&ws->cgcg_state, sizeof(CgcgDevState)));
    KHIP_CHECK_HIP(hipHostMalloc(reinterpret_cast<void **>(&ws->cgcg_snap), 2 * sizeof(CgcgDevState), hipHostMallocDefault));
  }
  if (!ws->hist_dev) KHIP_CHECK_HIP(hipMalloc(&ws->hist_dev, sizeof(double) * (size_t)kHistWindowMax));
  for (auto &e : ws->snap_ev) if (!e) KHIP_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  long long window = ctx->tune.hist_window;
  if (window < kDevChunk) window = kDevChunk;
  if (window > kHistWindowMax) window = kHistWindowMax;
  CgcgDevState *dev = ws->cgcg_state;
  CgcgDevState h;
  memset(&h, 0, sizeof(h));
  h.gamma = gamma0; h.alpha = gamma0 / delta0; h.beta = 0.0; h.rNorm = std::sqrt(gamma0); h.eps_tol = eps_tol;
  h.stop_seq = kSeqNever;
  h.hist = history ? ws->hist_dev : nullptr;
  h.hist_cap = window;
  KHIP_CHECK_HIP(hipMemcpyAsync(dev, &h, sizeof(h), hipMemcpyHostToDevice, ctx->stream));
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  double *x = ws->x, *r = ws->r, *p = ws->p, *s = ws->Ap, *w = ws->z;
  int64_t enq = 0;
  long long hist_base = 0;
  std::vector<double> win;
  auto drain_history = [&](long long upto_iter) -> int {
    const long long cnt = upto_iter - hist_base;
    if (!history || cnt <= 0) return KHIP_OK;
    win.resize((size_t)cnt);
    KHIP_CHECK_HIP(hipMemcpy(win.data(), ws->hist_dev, sizeof(double) * (size_t)cnt, hipMemcpyDeviceToHost));
    for (double val : win) ws->box.push(val);
    return KHIP_OK;
  };
  int rc = KHIP_OK;
  bool stopped = false;
  for (int chunk = 0; !stopped; ++chunk) {
    const int64_t cnt = std::min<int64_t>(kDevChunk, itmax - enq);
    if (history && enq + cnt - hist_base > window) {
      KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
      CgcgDevState cur;
      KHIP_CHECK_HIP(hipMemcpy(&cur, dev, sizeof(cur), hipMemcpyDeviceToHost));
      if (cur.stop_seq != kSeqNever) break;
      if ((rc = drain_history(cur.iter)) != KHIP_OK) break;
      hist_base = cur.iter;
      KHIP_CHECK_HIP(hipMemcpy(&dev->hist_base, &hist_base, sizeof(hist_base), hipMemcpyHostToDevice));
    }
    for (int64_t i = 0; i < cnt && rc == KHIP_OK; ++i) {
      const long long j = (long long)(enq + i);
      rc = launch_cgcg_update(ctx, n, dev, 2 * j, w, r, p, s, x);
      if (rc != KHIP_OK) break;
      ctx->ctl = SeqCtl{&dev->stop_seq, 2 * j + 1, EPI_CGCG, dev};
      const int slot = take_slots(ctx, 2);
      rc = spmv_any(ctx, A, r, w, slot, nullptr, 2);                              // w = A r ; (r.w, r.r)
      if (rc == KHIP_OK && ctx->comm) rc = comm_allreduce_dd_device(ctx, slot, 2);
      ctx->ctl = SeqCtl{};
    }
    ctx->ctl = SeqCtl{};
    if (rc != KHIP_OK) break;
    enq += cnt;
    const int b = chunk & 1;
    KHIP_CHECK_HIP(hipMemcpyAsync(&ws->cgcg_snap[b], dev, sizeof(CgcgDevState), hipMemcpyDeviceToHost, ctx->stream));
    KHIP_CHECK_HIP(hipEventRecord(ws->snap_ev[b], ctx->stream));
    if (chunk >= 1) {
      KHIP_CHECK_HIP(hipEventSynchronize(ws->snap_ev[b ^ 1]));
      if (ws->cgcg_snap[b ^ 1].stop_seq != kSeqNever) stopped = true;
    }
    if (enq >= itmax) stopped = true;
    if (!stopped && time_limit_reached(ctx, now_s() - t0, timemax)) { *overtimed = true; stopped = true; }
  }
  ctx->ctl = SeqCtl{};
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  if (rc != KHIP_OK) return rc;
  KHIP_CHECK_HIP(hipMemcpy(out, dev, sizeof(CgcgDevState), hipMemcpyDeviceToHost));
  return drain_history(out->iter);
}

// Pipelined CG (Ghysels & Vanroose 2014), opt-in as options.variant = 2: per iteration ONE reduction (r.w, r.r) and one
// product q = A w that does not depend on it, then all recurrences in one pass (launch_pcg_update).  On N GPUs the
// all-gather of the reduction runs on the communication stream WHILE the product runs on the main stream (RCCL backend with a
// separate halo communicator; otherwise in program order).  Scalars: the same epilogue as the single-reduction variant
// (beta = gamma' / gamma, alpha = gamma' / (delta - beta gamma' / alpha)).  Not the reference's recurrence: own parity budget.
// On entry r = b - A x0, w = A r, p = s = z = 0-initialised by the first update (beta_0 = 0), state = (gamma0, alpha0, 0).
int cg_pipelined_loop(khip_cg_workspace *ws, const khip_csr *A, double gamma0, double delta0, double eps_tol, int64_t itmax,
                      bool history, double t0, double timemax, CgcgDevState *out, bool *overtimed) {
  khip_ctx *ctx = ws->ctx;
  const int64_t n = ws->n;
  if (!ws->cgcg_state) {
    KHIP_CHECK_HIP(hipMalloc(&ws->cgcg_state, sizeof(CgcgDevState)));
    KHIP_CHECK_HIP(hipHostMalloc(reinterpret_cast<void **>(&ws->cgcg_snap), 2 * sizeof(CgcgDevState), hipHostMallocDefault));
  }
  if (!ws->hist_dev) KHIP_CHECK_HIP(hipMalloc(&ws->hist_dev, sizeof(double) * (size_t)kHistWindowMax));
  for (auto &e : ws->snap_ev) if (!e) KHIP_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  long long window = ctx->tune.hist_window;
  if (window < kDevChunk) window = kDevChunk;
  if (window > kHistWindowMax) window = kHistWindowMax;
  CgcgDevState *dev = ws->cgcg_state;
  CgcgDevState h;
  memset(&h, 0, sizeof(h));
  h.gamma = gamma0; h.alpha = gamma0 / delta0; h.beta = 0.0; h.rNorm = std::sqrt(gamma0); h.eps_tol = eps_tol;
  h.stop_seq = kSeqNever;
  h.hist = history ? ws->hist_dev : nullptr;
  h.hist_cap = window;
  KHIP_CHECK_HIP(hipMemcpyAsync(dev, &h, sizeof(h), hipMemcpyHostToDevice, ctx->stream));
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  double *x = ws->x, *r = ws->r, *p = ws->p, *s = ws->Ap, *w = ws->z, *z = ws->pz, *q = ws->pq;
  int64_t enq = 0;
  long long hist_base = 0;
  std::vector<double> win;
  auto drain_history = [&](long long upto_iter) -> int {
    const long long cnt = upto_iter - hist_base;
    if (!history || cnt <= 0) return KHIP_OK;
    win.resize((size_t)cnt);
    KHIP_CHECK_HIP(hipMemcpy(win.data(), ws->hist_dev, sizeof(double) * (size_t)cnt, hipMemcpyDeviceToHost));
    for (double val : win) ws->box.push(val);
    return KHIP_OK;
  };
  int rc = KHIP_OK;
  bool stopped = false;
  for (int chunk = 0; !stopped; ++chunk) {
    const int64_t cnt = std::min<int64_t>(kDevChunk, itmax - enq);
    if (history && enq + cnt - hist_base > window) {
      KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
      CgcgDevState cur;
      KHIP_CHECK_HIP(hipMemcpy(&cur, dev, sizeof(cur), hipMemcpyDeviceToHost));
      if (cur.stop_seq != kSeqNever) break;
      if ((rc = drain_history(cur.iter)) != KHIP_OK) break;
      hist_base = cur.iter;
      KHIP_CHECK_HIP(hipMemcpy(&dev->hist_base, &hist_base, sizeof(hist_base), hipMemcpyHostToDevice));
    }
    for (int64_t i = 0; i < cnt && rc == KHIP_OK; ++i) {
      const long long j = (long long)(enq + i);
      // (a) q = A w: needs nothing of the reduction still in flight on the communication stream
      ctx->ctl = SeqCtl{&dev->stop_seq, 3 * j, EPI_NONE, nullptr};
      rc = spmv_any(ctx, A, w, q, -1);
      if (rc != KHIP_OK) break;
      // (b) the recurrences with the alpha, beta the PREVIOUS reduction's epilogue left in the state: wait for it now
      if (ctx->comm) rc = comm_allreduce_dd_device_end(ctx);
      if (rc == KHIP_OK) rc = launch_pcg_update(ctx, n, dev, 3 * j + 1, q, z, s, p, x, r, w);
      if (rc != KHIP_OK) break;
      // (c) (r.w, r.r) of the new iterate and the scalars of the next iteration; on several ranks its all-gather, the
      //     cross-rank combine and the scalar epilogue run on the communication stream while the NEXT product (a) runs here
      ctx->ctl = SeqCtl{&dev->stop_seq, 3 * j + 2, EPI_CGCG, dev};
      const int slot = take_slots(ctx, 2);
      rc = launch_dot2(ctx, n, r, w, slot);
      if (rc == KHIP_OK && ctx->comm) rc = comm_allreduce_dd_device_begin(ctx, slot, 2);
      ctx->ctl = SeqCtl{};
    }
    if (rc == KHIP_OK && ctx->comm) rc = comm_allreduce_dd_device_end(ctx);      // the snapshot below reads the state the epilogue writes
    ctx->ctl = SeqCtl{};
    if (rc != KHIP_OK) break;
    enq += cnt;
    const int b = chunk & 1;
    KHIP_CHECK_HIP(hipMemcpyAsync(&ws->cgcg_snap[b], dev, sizeof(CgcgDevState), hipMemcpyDeviceToHost, ctx->stream));
    KHIP_CHECK_HIP(hipEventRecord(ws->snap_ev[b], ctx->stream));
    if (chunk >= 1) {
      KHIP_CHECK_HIP(hipEventSynchronize(ws->snap_ev[b ^ 1]));
      if (ws->cgcg_snap[b ^ 1].stop_seq != kSeqNever) stopped = true;
    }
    if (enq >= itmax) stopped = true;
    if (!stopped && time_limit_reached(ctx, now_s() - t0, timemax)) { *overtimed = true; stopped = true; }
  }
  ctx->ctl = SeqCtl{};
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  if (rc != KHIP_OK) return rc;
  KHIP_CHECK_HIP(hipMemcpy(out, dev, sizeof(CgcgDevState), hipMemcpyDeviceToHost));
  return drain_history(out->iter);
}

}  // namespace

extern "C" {

khip_options khip_default_options(void) {
  khip_options o;
  memset(&o, 0, sizeof(o));
  o.atol = NAN; o.rtol = NAN; o.timemax = NAN;
  o.fused = 2;
  return o;
}

int khip_cg_workspace_create(khip_ctx *ctx, int64_t m, int64_t n, khip_cg_workspace **out) {
  KHIP_REQUIRE(ctx && out && m >= 0 && n >= 0, "cg_workspace_create: bad argument");
  khip_cg_workspace *ws = new khip_cg_workspace();
  ws->ctx = ctx; ws->m = m; ws->n = n;
  (void)take_alloc_seconds();
  // x, r, p, Ap allocated; dx, npc_dir, z stay empty until needed (src/krylov_workspaces.jl:269-285)
  int rc = alloc_vec(ctx, n, &ws->x);
  if (!rc) rc = alloc_vec(ctx, n, &ws->r);
  if (!rc) rc = alloc_vec(ctx, n, &ws->p);
  if (!rc) rc = alloc_vec(ctx, n, &ws->Ap);
  if (rc) { khip_cg_workspace_destroy(ws); return rc; }
  ws->box.st.allocation_timer = take_alloc_seconds();           // workspace.stats.allocation_timer, src/krylov_workspaces.jl:288-289
  *out = ws;
  return KHIP_OK;
}

// CgWorkspace whose x, r, p, Ap are the CALLER's device vectors of n entries (a Julia CgWorkspace{Float64,Float64,HIPVector}:
// src/krylov_workspaces.jl:236-248).  Nothing is allocated; solution(ws) === x.  The lazily allocated fields (z, dx = Δx,
// npc_dir, src/cg.jl:142-143) are handed over with khip_cg_workspace_adopt_vector when the caller has them; a solve that
// needs one the caller did not hand over allocates it here (and owns it).
int khip_cg_workspace_adopt(khip_ctx *ctx, int64_t m, int64_t n, double *x, double *r, double *p, double *Ap,
                            khip_cg_workspace **out) {
  KHIP_REQUIRE(ctx && out && m >= 0 && n >= 0, "cg_workspace_adopt: bad argument");
  KHIP_REQUIRE(n == 0 || (x && r && p && Ap), "cg_workspace_adopt: x, r, p, Ap must be device vectors of n entries");
  KHIP_REQUIRE(n == 0 || (x != r && x != p && x != Ap && r != p && r != Ap && p != Ap), "cg_workspace_adopt: x, r, p, Ap must be distinct");
  khip_cg_workspace *ws = new khip_cg_workspace();
  ws->ctx = ctx; ws->m = m; ws->n = n;
  ws->x = x; ws->r = r; ws->p = p; ws->Ap = Ap;
  for (double *v : {x, r, p, Ap}) ws->borrowed.add(v);
  (void)take_alloc_seconds();
  ws->box.st.allocation_timer = 0.0;                            // the caller allocated (and timed) the vectors
  *out = ws;
  return KHIP_OK;
}

int khip_cg_workspace_adopt_vector(khip_cg_workspace *ws, const char *name, double *ptr) {
  KHIP_REQUIRE(ws && name, "cg_workspace_adopt_vector: null argument");
  struct { const char *k; double **slot; bool required; } tab[] = {
      {"x", &ws->x, true}, {"r", &ws->r, true}, {"p", &ws->p, true}, {"Ap", &ws->Ap, true},
      {"z", &ws->z, false}, {"dx", &ws->dx, false}, {"npc_dir", &ws->npc_dir, false}};
  for (auto &e : tab)
    if (strcmp(e.k, name) == 0) {
      KHIP_REQUIRE(ptr || !e.required, "cg_workspace_adopt_vector: x, r, p, Ap cannot be emptied");
      if (const char *other = held_elsewhere(tab, e.slot, ptr, nullptr)) {
        set_error("cg_workspace_adopt_vector: the pointer for '%s' already is the workspace's '%s' (every vector needs its own storage)", name, other);
        return KHIP_ERR_INVALID;
      }
      adopt_into(ws->ctx, ws->borrowed, e.slot, ptr);
      return KHIP_OK;
    }
  set_error("cg_workspace_adopt_vector: unknown vector '%s' (x, r, p, Ap, z, dx, npc_dir)", name);
  return KHIP_ERR_INVALID;
}

int khip_cg_workspace_destroy(khip_cg_workspace *ws) {
  if (!ws) return KHIP_OK;
  for (double *v : {ws->dx, ws->x, ws->r, ws->npc_dir, ws->p, ws->Ap, ws->z, ws->pz, ws->pq}) free_unless_borrowed(ws->ctx, ws->borrowed, v);
  if (ws->dev_state) (void)hipFree(ws->dev_state);
  if (ws->snap) (void)hipHostFree(ws->snap);
  if (ws->cgcg_state) (void)hipFree(ws->cgcg_state);
  if (ws->cgcg_snap) (void)hipHostFree(ws->cgcg_snap);
  if (ws->hist_dev) (void)hipFree(ws->hist_dev);
  for (auto e : ws->snap_ev) if (e) (void)hipEventDestroy(e);
  delete ws;
  return KHIP_OK;
}

int khip_cg_warm_start(khip_cg_workspace *ws, const double *x0) {
  KHIP_REQUIRE(ws && x0, "cg_warm_start: null argument");
  if (!ws->dx) KHIP_TRY(alloc_vec(ws->ctx, ws->n, &ws->dx));
  if (x0 != ws->dx) KHIP_TRY(khip_copy(ws->ctx, ws->n, ws->dx, x0));   // x0 == dx: an adopted Δx that warm_start! already filled
  ws->warm_start = true;
  return KHIP_OK;
}

double *khip_cg_solution(khip_cg_workspace *ws) { return ws ? ws->x : nullptr; }
const khip_stats *khip_cg_stats(khip_cg_workspace *ws) { return ws ? &ws->box.st : nullptr; }
int khip_cg_last_path(khip_cg_workspace *ws) { return ws ? ws->box.path : -1; }
double *khip_cg_vector(khip_cg_workspace *ws, const char *name) {
  if (!ws || !name) return nullptr;
  struct { const char *k; double *p; } tab[] = {{"x", ws->x}, {"r", ws->r}, {"p", ws->p}, {"Ap", ws->Ap},
                                                {"z", ws->z}, {"dx", ws->dx}, {"npc_dir", ws->npc_dir}};
  for (auto &e : tab) if (strcmp(e.k, name) == 0) return e.p;
  return nullptr;
}
size_t khip_cg_workspace_bytes(khip_cg_workspace *ws) {
  if (!ws) return 0;
  size_t cnt = 0;
  for (double *v : {ws->dx, ws->x, ws->r, ws->npc_dir, ws->p, ws->Ap, ws->z}) cnt += v ? 1 : 0;
  return cnt * sizeof(double) * (size_t)ws->n;
}

// to_boundary with M === I (src/krylov_utils.jl:375-402) + roots_quadratic (:110-152)
static int roots_quadratic(double q2, double q1, double q0, double *r1, double *r2, int nitref = 1) {
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
  const double rhs = std::sqrt(kEps) * q1 * q1;
  if (std::fabs(q0 * q2) > rhs) {
    const double rho = q1 * q1 - 4 * q2 * q0;
    if (rho < 0) return -1;
    const double d = -(q1 + std::copysign(std::sqrt(rho), q1)) / 2;
    root1 = d / q2;
    root2 = q0 / d;
  } else {
    root1 = -q1 / q2;
    root2 = 0.0;
  }
  for (int it = 0; it < nitref; ++it) {
    const double q = (q2 * root1 + q1) * root1 + q0, dq = 2 * q2 * root1 + q1;
    if (dq == 0.0) continue;
    root1 = root1 - q / dq;
  }
  for (int it = 0; it < nitref; ++it) {
    const double q = (q2 * root2 + q1) * root2 + q0, dq = 2 * q2 * root2 + q1;
    if (dq == 0.0) continue;
    root2 = root2 - q / dq;
  }
  *r1 = root1; *r2 = root2;
  return 0;
}

}  // extern "C"

// to_boundary(n, x, d, z, radius; flip, xNorm2, dNorm2) with M === I (src/krylov_utils.jl:375-402): the two sigma with
// ||x + sigma d|| = radius.  The dots run on the device (kdotr); a zero xNorm2 / dNorm2 argument means "compute it".
// Returns 0, or a negative code with the reference's error text in *err.
static int to_boundary_dev(khip_ctx *ctx, int64_t n, const double *x, const double *d, double radius, bool flip, double xNorm2,
                           double dNorm2, double *s1, double *s2, const char **err) {
  if (!(radius > 0)) { *err = "radius must be positive"; return -1; }
  double rxd;
  KHIP_TRY(khip_dot(ctx, n, x, d, &rxd));
  if (dNorm2 == 0.0) KHIP_TRY(khip_dot(ctx, n, d, d, &dNorm2));
  if (xNorm2 == 0.0) KHIP_TRY(khip_dot(ctx, n, x, x, &xNorm2));
  if (dNorm2 == 0.0) { *err = "zero direction"; return -2; }
  if (flip) rxd = -rxd;
  const double radius2 = radius * radius;
  if (!(xNorm2 <= radius2)) { *err = "outside of the trust region"; return -3; }
  if (roots_quadratic(dNorm2, 2 * rxd, xNorm2 - radius2, s1, s2)) { *err = "The quadratic `q` doesn't have real roots."; return -4; }
  return 0;
}

extern "C" {

// Test-only exports of the scalar helpers the solver loops use (tests/test_abi.py, tests/test_gpu_solvers.py hold the
// reference's exact known answers, test/test_aux.jl:3-117, against THESE copies, not only against the oracle's).
int khip_test_sym_givens(double a, double b, double *c, double *s, double *rho);      // defined next to sym_givens below
int khip_test_roots_quadratic(double q2, double q1, double q0, int nitref, double *root1, double *root2) {
  KHIP_REQUIRE(root1 && root2, "test_roots_quadratic: null argument");
  if (roots_quadratic(q2, q1, q0, root1, root2, nitref)) { set_error("The quadratic `q` doesn't have real roots."); return KHIP_ERR_NUMERIC; }
  return KHIP_OK;
}
int khip_test_to_boundary(khip_ctx *ctx, int64_t n, const double *x, const double *d, double radius, int flip, double *sigma1,
                          double *sigma2) {
  KHIP_REQUIRE(ctx && x && d && sigma1 && sigma2, "test_to_boundary: null argument");
  const char *err = "";
  const int rc = to_boundary_dev(ctx, n, x, d, radius, flip != 0, 0.0, 0.0, sigma1, sigma2, &err);
  if (rc < 0) { set_error("%s", err); return KHIP_ERR_NUMERIC; }
  return rc;
}

int khip_cg_solve(khip_cg_workspace *ws, const khip_operator *A, const khip_operator *M, const double *b,
                  const khip_options *opts_in) {
  KHIP_REQUIRE(ws && A && b, "cg_solve: null argument");
  khip_ctx *ctx = ws->ctx;
  khip_options o = opts_in ? *opts_in : khip_default_options();
  const double t0 = now_s();
  const double timemax = timemax_of(o);
  const int64_t n = ws->n;
  khip_stats *st = &ws->box.st;
  const double atol = tol_or_default(o.atol), rtol = tol_or_default(o.rtol);
  const double radius = o.radius;
  const bool linesearch = o.linesearch != 0;
  const bool fused = o.fused != 0;
  const int verbose = o.verbose;
  (void)take_alloc_seconds();

  if (ws->m != ws->n) return ws->box.fail(KHIP_ERR_INVALID, "System must be square");
  if (o.variant != 0 && o.variant != 1 && o.variant != 2)
    return ws->box.fail(KHIP_ERR_INVALID, "cg: options.variant must be 0 (cg! recurrence), 1 (single-reduction CG) or 2 (pipelined CG)");
  if (o.variant != 0 && verbose > 0)       // the log's pAp / alpha / sigma columns belong to cg!'s own recurrence
    return ws->box.fail(KHIP_ERR_UNSUPPORTED, "cg: verbose > 0 prints the rows of the reference recurrence: use variant = 0");
  if (A->csr && !A->apply) {
    int64_t am, an;
    khip_csr_shape(A->csr, &am, &an, nullptr);
    if (am != ws->m) return ws->box.fail(KHIP_ERR_INVALID, "(workspace.m, workspace.n) is inconsistent with size(A)");
  }
  if (linesearch && radius > 0)
    return ws->box.fail(KHIP_ERR_INVALID, "`linesearch` set to `true` but trust-region radius > 0");
  if (ws->warm_start && linesearch)
    return ws->box.fail(KHIP_ERR_INVALID, "warm_start and linesearch cannot be used together");
  if (verbose > 0) klogf(o.log_fd, "CG: system of %lld equations in %lld variables\n", (long long)n, (long long)n);      // src/cg.jl:132, after the argument checks as there
  const bool MisI = (M == nullptr);
  if (!MisI && radius > 0)
    return ws->box.fail(KHIP_ERR_UNSUPPORTED, "radius > 0 with a preconditioner needs ldiv!(M): not available through a callback");

  if (!MisI && !ws->z) K(alloc_vec(ctx, n, &ws->z));                               // src/cg.jl:142
  if ((linesearch || radius > 0) && !ws->npc_dir) K(alloc_vec(ctx, n, &ws->npc_dir));  // :143
  double *dx = ws->dx, *x = ws->x, *r = ws->r, *p = ws->p, *Ap = ws->Ap;
  const bool warm_start = ws->warm_start;
  ws->box.reset();
  double *z = MisI ? r : ws->z;
  double *npc_dir = ws->npc_dir;

  double gamma;
  // :153-162 in one pass when nothing sits between the four primitives (M = I, no warm start; fused paths only: fused = 0
  // issues the reference's sequence): same x, r, p, gamma, 32n instead of 48n bytes -- 0.35 of the 1.05 ms a cg! call
  // costs beyond its iterations at 512^3 (profiles/r03_cg_solve_overhead.log)
  const bool one_pass_setup = fused && MisI && !warm_start && ctx->tune.cg_setup_fused && b != x && b != r && b != p;
  if (one_pass_setup) {
    K(khip_cg_setup(ctx, n, b, x, r, p, &gamma));
  } else {
  K(khip_fill(ctx, n, x, 0.0));                                                    // :153
  if (warm_start) {
    K(apply_op(ctx, A, dx, r));
    K(khip_axpby(ctx, n, 1.0, b, -1.0, r));
  } else {
    K(khip_copy(ctx, n, r, b));
  }
  if (!MisI) K(apply_op(ctx, M, r, z));
  K(khip_copy(ctx, n, p, z));
  K(khip_dot(ctx, n, r, z, &gamma));                                               // :162
  }
  if (!(gamma >= 0))
    return ws->box.fail(KHIP_ERR_NUMERIC,
                        "The linear operator `A` or the preconditioner `M` is not symmetric positive definite.");
  double rNorm = std::sqrt(gamma);
  if (o.history) ws->box.push(rNorm);
  if (gamma == 0) {                                                                // :166-174
    st->niter = 0; st->solved = 1; st->inconsistent = 0;
    st->timer = now_s() - t0;
    snprintf(st->status, sizeof(st->status), "x is a zero-residual solution");
    if (warm_start) K(khip_axpy(ctx, n, 1.0, dx, x));
    ws->warm_start = false;
    ws->box.publish();
    return KHIP_OK;
  }

  int64_t iter = 0;
  const int64_t itmax = o.itmax == 0 ? 2 * global_rows(ctx, A, n) : o.itmax;   // 2n of the GLOBAL system on every rank
  double pAp = 0.0;
  double pNorm2 = gamma;
  const double eps_tol = atol + rtol * rNorm;

  if (verbose > 0) klogf(o.log_fd, "    k      \xe2\x80\x96r\xe2\x80\x96       pAp         \xce\xb1         \xcf\x83  timer\n");   // :182  k ‖r‖ pAp α σ timer
  if (kdisplay(iter, verbose)) klogf(o.log_fd, "%5lld  %7.1e", (long long)iter, rNorm);                                   // :183

  bool solved = rNorm <= eps_tol;
  bool tired = iter >= itmax;
  bool inconsistent = false, on_boundary = false, zero_curvature = false, user_requested_exit = false,
       overtimed = false;
  const char *status = "unknown";

  if (o.variant == 1) {                                                           // single-reduction CG, opt-in
    if (A->apply || !A->csr || !MisI || radius != 0 || linesearch || o.callback)
      return ws->box.fail(KHIP_ERR_UNSUPPORTED, "cg variant 1 needs a CSR operator, M = I, no trust region / linesearch / callback");
    if (!(solved || tired)) {
      if (!ws->z) K(alloc_vec(ctx, n, &ws->z));                                     // w = A r lives in the (unused) z slot
      K(khip_fill(ctx, n, Ap, 0.0));                                                // s = A p starts as 0 (beta_0 = 0)
      double two[2];
      const int slot = take_slots(ctx, 2);
      K(spmv_any(ctx, A->csr, r, ws->z, slot, nullptr, 2));                         // w = A r ; (r.w, r.r)
      K(fetch_results(ctx, slot, 2, two));
      if (!(two[0] > 0))
        return ws->box.fail(KHIP_ERR_NUMERIC,
                            "The linear operator `A` or the preconditioner `M` is not symmetric positive definite.");
      CgcgDevState fin;
      K(cg_single_reduction_loop(ws, A->csr, gamma, two[0], eps_tol, itmax, o.history != 0, t0, timemax, &fin, &overtimed));
      iter = fin.iter;
      rNorm = fin.rNorm;
      solved = fin.solved != 0;
      inconsistent = false;
      tired = iter >= itmax;
      if (fin.breakdown && !solved) { zero_curvature = true; inconsistent = true; }   // as the cg! path reports it (:203-205, linesearch = false)
    }
  }
  // (a verbose solve prints alpha, pAp and sigma of every displayed iteration: it runs the host-driven loop below)
  if (o.variant == 2) {                                                           // pipelined CG, opt-in
    if (A->apply || !A->csr || !MisI || radius != 0 || linesearch || o.callback)
      return ws->box.fail(KHIP_ERR_UNSUPPORTED, "cg variant 2 needs a CSR operator, M = I, no trust region / linesearch / callback");
    if (!(solved || tired)) {
      if (!ws->z) K(alloc_vec(ctx, n, &ws->z));                                     // w = A r lives in the (unused) z slot
      if (!ws->pz) K(alloc_vec(ctx, n, &ws->pz));
      if (!ws->pq) K(alloc_vec(ctx, n, &ws->pq));
      K(khip_fill(ctx, n, Ap, 0.0));                                                // s, z, p start as 0 (beta_0 = 0: the first update overwrites them)
      K(khip_fill(ctx, n, ws->pz, 0.0));
      K(khip_fill(ctx, n, p, 0.0));
      double two[2];
      const int slot = take_slots(ctx, 2);
      K(spmv_any(ctx, A->csr, r, ws->z, slot, nullptr, 2));                         // w = A r ; (r.w, r.r)
      K(fetch_results(ctx, slot, 2, two));
      if (!(two[0] > 0))
        return ws->box.fail(KHIP_ERR_NUMERIC,
                            "The linear operator `A` or the preconditioner `M` is not symmetric positive definite.");
      CgcgDevState fin;
      K(cg_pipelined_loop(ws, A->csr, gamma, two[0], eps_tol, itmax, o.history != 0, t0, timemax, &fin, &overtimed));
      iter = fin.iter;
      rNorm = fin.rNorm;
      solved = fin.solved != 0;
      inconsistent = false;
      tired = iter >= itmax;
      if (fin.breakdown && !solved) { zero_curvature = true; inconsistent = true; }
    }
  }
  const bool device_loop = o.variant == 0 && o.fused >= 2 && !A->apply && A->csr && MisI && radius == 0 && !linesearch && !o.callback && verbose <= 0;
  ws->box.path = device_loop ? 2 : (fused ? 1 : 0);
  if (device_loop && !(solved || tired)) {
    CgDevState fin;
    K(cg_device_loop(ws, A->csr, gamma, eps_tol, itmax, o.history != 0, t0, timemax, &fin, &overtimed));
    if (fin.not_spd)
      return ws->box.fail(KHIP_ERR_NUMERIC,
                          "The linear operator `A` or the preconditioner `M` is not symmetric positive definite.");
    iter = fin.iter;
    rNorm = fin.rNorm;
    solved = fin.solved != 0;
    zero_curvature = fin.zero_curvature != 0;
    inconsistent = fin.inconsistent != 0;
    tired = iter >= itmax;
  }

  while (!device_loop && o.variant == 0 && !(solved || tired || zero_curvature || user_requested_exit || overtimed)) {
    if (fused && !A->apply) {
      K(khip_spmv_dot(ctx, A->csr, p, Ap, &pAp));                                  // :196-197 fused
    } else {
      K(apply_op(ctx, A, p, Ap));                                                  // :196
      K(khip_dot(ctx, n, p, Ap, &pAp));                                            // :197
    }
    if ((pAp <= kEps * pNorm2) && (radius == 0)) {                                 // :198-211
      if (std::fabs(pAp) <= kEps * pNorm2) {
        zero_curvature = true;
        inconsistent = !linesearch;
      }
      if (linesearch) {
        if (iter == 0) K(khip_copy(ctx, n, x, p));
        K(khip_copy(ctx, n, npc_dir, p));
        st->npcCount = 1;
        st->indefinite = 1;
        solved = true;
      }
    }
    if (zero_curvature || solved) continue;

    double alpha = gamma / pAp;                                                    // :213
    double sigma;
    if (radius == 0) {
      sigma = alpha;
    } else {                                                                       // to_boundary(n, x, p, z, radius, dNorm2=pNorm²) :216
      double s1, s2;
      const char *err = "";
      const int rcb = to_boundary_dev(ctx, n, x, p, radius, false, 0.0, pNorm2, &s1, &s2, &err);
      if (rcb < 0) return ws->box.fail(KHIP_ERR_NUMERIC, err);
      if (rcb > 0) return ws->box.fail_rc(rcb);
      sigma = s1 > s2 ? s1 : s2;
    }
    if (kdisplay(iter, verbose)) klogf(o.log_fd, "  %8.1e  %8.1e  %8.1e  %.2fs\n", pAp, alpha, sigma, now_s() - t0);        // :224
    if ((radius > 0) && ((pAp <= 0) || (alpha > sigma))) {                         // :229-237
      alpha = sigma;
      if (pAp <= 0) {
        K(khip_copy(ctx, n, npc_dir, p));
        st->npcCount = 1;
        st->indefinite = 1;
      }
      on_boundary = true;
    }

    double gamma_next;
    bool x_pending = false;
    if (fused && MisI) {
      // :240,:242 fused; the x update (:239) is carried to the p update below -- nothing reads x in between
      K(khip_axpy_sqnorm(ctx, n, -alpha, Ap, r, &gamma_next));
      x_pending = true;
    } else {
      K(khip_axpy(ctx, n, alpha, p, x));                                           // :239
      K(khip_axpy(ctx, n, -alpha, Ap, r));                                         // :240
      if (!MisI) K(apply_op(ctx, M, r, z));                                        // :241
      K(khip_dot(ctx, n, r, z, &gamma_next));                                      // :242
    }
    if (!(gamma_next >= 0))
      return ws->box.fail(KHIP_ERR_NUMERIC,
                          "The linear operator `A` or the preconditioner `M` is not symmetric positive definite.");
    rNorm = std::sqrt(gamma_next);
    if (o.history) ws->box.push(rNorm);

    const bool resid_decrease_mach = (rNorm + 1.0 <= 1.0);
    const bool resid_decrease_lim = rNorm <= eps_tol;
    const bool resid_decrease = resid_decrease_lim || resid_decrease_mach;
    solved = resid_decrease || on_boundary;

    if (!solved) {                                                                 // :255-260
      const double beta = gamma_next / gamma;
      pNorm2 = gamma_next + beta * beta * pNorm2;
      gamma = gamma_next;
      if (x_pending) K(khip_cg_update(ctx, n, alpha, beta, r, p, x));               // :239 + :259 in one pass over p
      else K(khip_axpby(ctx, n, 1.0, z, beta, p));
    } else if (x_pending) {
      K(khip_axpy(ctx, n, alpha, p, x));                                           // :239
    }

    iter = iter + 1;
    tired = iter >= itmax;
    if (o.callback) { ws->box.publish(); user_requested_exit = o.callback(ws, o.callback_data) != 0; }    // :264
    overtimed = time_limit_reached(ctx, now_s() - t0, timemax);
    if (kdisplay(iter, verbose)) klogf(o.log_fd, "%5lld  %7.1e", (long long)iter, rNorm);                                 // :267
  }
  if (verbose > 0) { klogf(o.log_fd, "\n\n"); klog_flush(o.log_fd); }                                                         // :269

  if (solved && on_boundary) status = "on trust-region boundary";
  if (solved && st->indefinite) status = "nonpositive curvature";
  if (solved && strcmp(status, "unknown") == 0) status = "solution good enough given atol and rtol";
  if (zero_curvature) status = "zero curvature detected";
  if (tired) status = "maximum number of iterations exceeded";
  if (user_requested_exit) status = "user-requested exit";
  if (overtimed) status = "time limit exceeded";

  if (warm_start) K(khip_axpy(ctx, n, 1.0, dx, x));
  ws->warm_start = false;
  K(khip_ctx_sync(ctx));

  st->niter = (int)iter;
  st->solved = solved;
  st->inconsistent = inconsistent;
  st->timer = now_s() - t0;
  st->allocation_timer += take_alloc_seconds();                 // lazy allocations of this solve (allocate_if)
  snprintf(st->status, sizeof(st->status), "%s", status);
  ws->box.publish();
  return KHIP_OK;
}

}  // extern "C"

// ================================================================== GMRES =======
struct khip_gmres_workspace {
  khip_ctx *ctx;
  int64_t m, n, stride;
  int mem;                       // length(c)
  double *dx = nullptr, *x = nullptr, *w = nullptr, *p = nullptr, *q = nullptr;
  std::vector<double *> V;       // V[i] -> slice of a slab
  std::vector<double *> slabs;   // owned allocations backing V
  std::vector<int> slab_count;
  std::vector<double> c, s, z, R;
  std::vector<double> look;      // mem + 1 doubles: landing buffer of the look-ahead fetch (k coefficients + ||q||^2); no per-iteration allocation
  int inner_iter = 0;
  bool warm_start = false;
  Borrowed borrowed;             // vectors of a caller's GmresWorkspace (khip_gmres_workspace_adopt)
  khip_grow_fn grow = nullptr;   // adopted workspaces: the caller's push!(V, similar(x)) (src/gmres.jl:319-324)
  void *grow_data = nullptr;
  StatsBox box;
};

static int gmres_grow_basis(khip_gmres_workspace *ws, int count) {   // push!(V, similar(x)) in slabs
  if (ws->grow) {                // the basis belongs to the caller: one push!(V, similar(x)) per missing vector
    for (int i = 0; i < count; ++i) {
      const double t_alloc = now_s();
      double *v = ws->grow(ws->grow_data);
      g_alloc_seconds += now_s() - t_alloc;
      if (!v) { set_error("gmres: the workspace's grow callback returned no vector"); return KHIP_ERR_INVALID; }
      ws->borrowed.add(v);
      ws->V.push_back(v);
    }
    return KHIP_OK;
  }
  double *slab = nullptr;
  const double t_alloc = now_s();
  KHIP_TRY(khip_malloc(ws->ctx, sizeof(double) * (size_t)ws->stride * (size_t)count, reinterpret_cast<void **>(&slab)));
  g_alloc_seconds += now_s() - t_alloc;
  ws->slabs.push_back(slab);
  ws->slab_count.push_back(count);
  for (int i = 0; i < count; ++i) ws->V.push_back(slab + (size_t)i * ws->stride);
  return KHIP_OK;
}

// sym_givens(a, b) for reals, src/krylov_utils.jl:21-51
static void sym_givens(double a, double b, double &c, double &s, double &rho) {
  auto sgn = [](double v) { return v > 0 ? 1.0 : (v < 0 ? -1.0 : 0.0); };
  if (b == 0.0) {
    c = sgn(a) + (a == 0.0 ? 1.0 : 0.0);
    s = 0.0;
    rho = std::fabs(a);
  } else if (a == 0.0) {
    c = 0.0;
    s = sgn(b);
    rho = std::fabs(b);
  } else if (std::fabs(b) > std::fabs(a)) {
    const double t = a / b;
    s = sgn(b) / std::sqrt(1.0 + t * t);
    c = s * t;
    rho = b / s;
  } else {
    const double t = b / a;
    c = sgn(a) / std::sqrt(1.0 + t * t);
    s = c * t;
    rho = a / c;
  }
}

extern "C" {

int khip_test_sym_givens(double a, double b, double *c, double *s, double *rho) {
  KHIP_REQUIRE(c && s && rho, "test_sym_givens: null argument");
  sym_givens(a, b, *c, *s, *rho);
  return KHIP_OK;
}

int khip_gmres_workspace_create(khip_ctx *ctx, int64_t m, int64_t n, int memory, khip_gmres_workspace **out) {
  KHIP_REQUIRE(ctx && out && m >= 0 && n >= 0, "gmres_workspace_create: bad argument");
  if (memory <= 0) memory = 20;
  if ((int64_t)memory > m) memory = (int)m;                          // memory = min(m, memory) :2900
  khip_gmres_workspace *ws = new khip_gmres_workspace();
  ws->ctx = ctx; ws->m = m; ws->n = n; ws->mem = memory;
  ws->stride = padded(n > 0 ? n : 1);
  const double t_alloc0 = now_s();
  int rc = alloc_vec(ctx, n, &ws->x);
  if (!rc) rc = alloc_vec(ctx, n, &ws->w);
  if (!rc && memory > 0) rc = gmres_grow_basis(ws, memory);         // the basis is ONE slab: V[i] are consecutive slices
  if (rc) { khip_gmres_workspace_destroy(ws); return rc; }
  ws->c.assign(memory, 0.0); ws->s.assign(memory, 0.0); ws->z.assign(memory, 0.0);
  ws->R.assign((size_t)memory * (memory + 1) / 2, 0.0);
  ws->look.assign((size_t)memory + 1, 0.0);
  (void)take_alloc_seconds();
  ws->box.st.allocation_timer = now_s() - t_alloc0;             // x, w, the basis slab and the host arrays (:2898-2918)
  *out = ws;
  return KHIP_OK;
}

// GmresWorkspace on the CALLER's vectors: x, w and the `memory` basis vectors V_host[0 .. memory) (device pointers in a host
// array; a Julia GmresWorkspace{Float64,Float64,HIPVector} keeps them as `V::Vector{S}`, src/krylov_workspaces.jl:2857-2873).
// The host arrays c, s, z, R of the reference stay inside the library (khip_gmres_host_state copies them out).
int khip_gmres_workspace_adopt(khip_ctx *ctx, int64_t m, int64_t n, int memory, double *x, double *w,
                               double *const *V_host, khip_gmres_workspace **out) {
  KHIP_REQUIRE(ctx && out && m >= 0 && n >= 0 && memory >= 0, "gmres_workspace_adopt: bad argument");
  KHIP_REQUIRE(n == 0 || (x && w && x != w), "gmres_workspace_adopt: x and w must be distinct device vectors of n entries");
  KHIP_REQUIRE(memory == 0 || V_host, "gmres_workspace_adopt: V_host must hold `memory` device pointers");
  khip_gmres_workspace *ws = new khip_gmres_workspace();
  ws->ctx = ctx; ws->m = m; ws->n = n; ws->mem = memory;
  ws->stride = padded(n > 0 ? n : 1);
  ws->x = x; ws->w = w;
  ws->borrowed.add(x); ws->borrowed.add(w);
  for (int i = 0; i < memory; ++i) {
    if (!V_host[i]) { delete ws; set_error("gmres_workspace_adopt: V_host[%d] is null", i); return KHIP_ERR_INVALID; }
    ws->V.push_back(V_host[i]);
    ws->borrowed.add(V_host[i]);
  }
  ws->c.assign(memory, 0.0); ws->s.assign(memory, 0.0); ws->z.assign(memory, 0.0);
  ws->R.assign((size_t)memory * (memory + 1) / 2, 0.0);
  ws->look.assign((size_t)memory + 1, 0.0);
  (void)take_alloc_seconds();
  ws->box.st.allocation_timer = 0.0;
  *out = ws;
  return KHIP_OK;
}

int khip_gmres_workspace_adopt_vector(khip_gmres_workspace *ws, const char *name, double *ptr) {
  KHIP_REQUIRE(ws && name, "gmres_workspace_adopt_vector: null argument");
  struct { const char *k; double **slot; bool required; } tab[] = {
      {"x", &ws->x, true}, {"w", &ws->w, true}, {"p", &ws->p, false}, {"q", &ws->q, false}, {"dx", &ws->dx, false}};
  for (auto &e : tab)
    if (strcmp(e.k, name) == 0) {
      KHIP_REQUIRE(ptr || !e.required, "gmres_workspace_adopt_vector: x and w cannot be emptied");
      if (const char *other = held_elsewhere(tab, e.slot, ptr, &ws->V)) {
        set_error("gmres_workspace_adopt_vector: the pointer for '%s' already is the workspace's '%s' (every vector needs its own storage)", name, other);
        return KHIP_ERR_INVALID;
      }
      adopt_into(ws->ctx, ws->borrowed, e.slot, ptr);
      return KHIP_OK;
    }
  set_error("gmres_workspace_adopt_vector: unknown vector '%s' (x, w, p, q, dx)", name);
  return KHIP_ERR_INVALID;
}

// The basis of an adopted workspace as the caller holds it NOW (k >= the workspace's memory: after a solve with
// restart = false the caller's V has grown through the grow callback, and the next solve starts from all of it).
int khip_gmres_workspace_adopt_basis(khip_gmres_workspace *ws, int k, double *const *V_host) {
  KHIP_REQUIRE(ws && k >= 0 && (k == 0 || V_host), "gmres_workspace_adopt_basis: bad argument");
  KHIP_REQUIRE(ws->slabs.empty(), "gmres_workspace_adopt_basis: this workspace owns its basis (khip_gmres_workspace_create)");
  KHIP_REQUIRE(k >= ws->mem, "gmres_workspace_adopt_basis: fewer vectors than the workspace's memory");
  for (int i = 0; i < k; ++i) {
    KHIP_REQUIRE(V_host[i] != nullptr, "gmres_workspace_adopt_basis: null basis vector");
    for (const double *named : {ws->x, ws->w, ws->p, ws->q, ws->dx})
      KHIP_REQUIRE(V_host[i] != named, "gmres_workspace_adopt_basis: a basis vector is also one of x, w, p, q, dx");
    for (int j = 0; j < i; ++j) KHIP_REQUIRE(V_host[i] != V_host[j], "gmres_workspace_adopt_basis: the same vector twice");
  }
  for (double *v : ws->V) ws->borrowed.drop(v);
  ws->V.assign(V_host, V_host + k);
  for (double *v : ws->V) ws->borrowed.add(v);
  return KHIP_OK;
}

int khip_gmres_workspace_set_grow(khip_gmres_workspace *ws, khip_grow_fn grow, void *userdata) {
  KHIP_REQUIRE(ws, "gmres_workspace_set_grow: null workspace");
  KHIP_REQUIRE(ws->slabs.empty() || !grow, "gmres_workspace_set_grow: this workspace owns its basis");
  ws->grow = grow; ws->grow_data = userdata;
  return KHIP_OK;
}

// Host state of the last solve in the reference's own storage (src/krylov_workspaces.jl:2866-2871): c, s, z of *len entries
// each, R packed upper triangular (len (len + 1) / 2), inner_iter.  Any output may be null; cap = entries c / s / z can take
// (R: cap (cap + 1) / 2).  *len = current length (it exceeds the workspace's memory after restart = false grew the basis).
int khip_gmres_host_state(khip_gmres_workspace *ws, int cap, double *c_host, double *s_host, double *z_host, double *R_host,
                          int *len, int *inner_iter) {
  KHIP_REQUIRE(ws && cap >= 0, "gmres_host_state: bad argument");
  const int L = (int)ws->c.size();
  if (len) *len = L;
  if (inner_iter) *inner_iter = ws->inner_iter;
  const int k = L < cap ? L : cap;
  if (c_host) memcpy(c_host, ws->c.data(), sizeof(double) * (size_t)k);
  if (s_host) memcpy(s_host, ws->s.data(), sizeof(double) * (size_t)k);
  if (z_host) memcpy(z_host, ws->z.data(), sizeof(double) * (size_t)(ws->z.size() < (size_t)k ? ws->z.size() : (size_t)k));
  if (R_host) {
    const size_t nr = (size_t)k * (k + 1) / 2;
    memcpy(R_host, ws->R.data(), sizeof(double) * (ws->R.size() < nr ? ws->R.size() : nr));
  }
  return KHIP_OK;
}

int khip_gmres_workspace_destroy(khip_gmres_workspace *ws) {
  if (!ws) return KHIP_OK;
  for (double *v : {ws->dx, ws->x, ws->w, ws->p, ws->q}) free_unless_borrowed(ws->ctx, ws->borrowed, v);
  for (double *s : ws->slabs) khip_free(ws->ctx, s);
  delete ws;
  return KHIP_OK;
}

int khip_gmres_warm_start(khip_gmres_workspace *ws, const double *x0) {
  KHIP_REQUIRE(ws && x0, "gmres_warm_start: null argument");
  if (!ws->dx) KHIP_TRY(alloc_vec(ws->ctx, ws->n, &ws->dx));
  if (x0 != ws->dx) KHIP_TRY(khip_copy(ws->ctx, ws->n, ws->dx, x0));
  ws->warm_start = true;
  return KHIP_OK;
}

double *khip_gmres_solution(khip_gmres_workspace *ws) { return ws ? ws->x : nullptr; }
const khip_stats *khip_gmres_stats(khip_gmres_workspace *ws) { return ws ? &ws->box.st : nullptr; }
int khip_gmres_last_path(khip_gmres_workspace *ws) { return ws ? ws->box.path : -1; }
size_t khip_gmres_workspace_bytes(khip_gmres_workspace *ws) {
  if (!ws) return 0;
  size_t cnt = ws->V.size();
  for (double *v : {ws->dx, ws->x, ws->w, ws->p, ws->q}) cnt += v ? 1 : 0;
  return cnt * sizeof(double) * (size_t)ws->n +
         sizeof(double) * (ws->c.size() + ws->s.size() + ws->z.size() + ws->R.size() + ws->look.size());
}

int khip_gmres_solve(khip_gmres_workspace *ws, const khip_operator *A, const khip_operator *M,
                     const khip_operator *N, const double *b, const khip_options *opts_in) {
  KHIP_REQUIRE(ws && A && b, "gmres_solve: null argument");
  khip_ctx *ctx = ws->ctx;
  khip_options o = opts_in ? *opts_in : khip_default_options();
  const double t0 = now_s();
  const double timemax = timemax_of(o);
  const int64_t n = ws->n;
  khip_stats *st = &ws->box.st;
  const double atol = tol_or_default(o.atol), rtol = tol_or_default(o.rtol);
  const bool restart = o.restart != 0, reorth = o.reorthogonalization != 0, fused = o.fused != 0;
  const int verbose = o.verbose;
  (void)take_alloc_seconds();
  if (ws->m != ws->n) return ws->box.fail(KHIP_ERR_INVALID, "System must be square");

  if (o.variant != 0 && o.variant != 1 && o.variant != 2)
    return ws->box.fail(KHIP_ERR_INVALID, "gmres: options.variant must be 0 (gmres! recurrence, modified Gram-Schmidt), 1 (CGS2) or 2 (s-step)");
  if (o.variant == 2 && (!restart || M || N || reorth || o.callback || A->apply || !A->csr))
    return ws->box.fail(KHIP_ERR_UNSUPPORTED, "gmres variant 2 (s-step) needs restart = true, a CSR operator, M = N = I, no reorthogonalization / callback");
  if (verbose > 0) klogf(o.log_fd, "GMRES: system of size %lld\n", (long long)n);                                          // src/gmres.jl:131, after the argument checks
  const bool MisI = (M == nullptr), NisI = (N == nullptr);
  // (a verbose solve keeps the host in step with every inner iteration, like the cg! / bicgstab! device loops)
  const bool look = o.variant == 0 && fused && o.fused >= 2 && MisI && NisI && !reorth && !o.callback && !A->apply && A->csr && verbose <= 0;
  ws->box.path = look ? 2 : (fused ? 1 : 0);
  if (!MisI && !ws->q) K(alloc_vec(ctx, n, &ws->q));                               // src/gmres.jl:142-144
  if (!NisI && !ws->p) K(alloc_vec(ctx, n, &ws->p));
  if (restart && !ws->dx) K(alloc_vec(ctx, n, &ws->dx));
  double *dx = ws->dx, *x = ws->x, *w = ws->w;
  std::vector<double *> &V = ws->V;
  std::vector<double> &c = ws->c, &s = ws->s, &z = ws->z, &R = ws->R;
  const bool warm_start = ws->warm_start;
  ws->box.reset();
  double *q = MisI ? w : ws->q;
  double *r0 = MisI ? w : ws->q;
  double *xr = restart ? dx : x;

  K(khip_fill(ctx, n, x, 0.0));                                                    // :155
  if (warm_start) {
    K(apply_op(ctx, A, dx, w));
    K(khip_axpby(ctx, n, 1.0, b, -1.0, w));
    if (restart) K(khip_axpy(ctx, n, 1.0, dx, x));
  } else {
    K(khip_copy(ctx, n, w, b));
  }
  if (!MisI) K(apply_op(ctx, M, w, r0));
  double beta;
  K(khip_nrm2(ctx, n, r0, &beta));                                                 // :166
  double rNorm = beta;
  if (o.history) ws->box.push(beta);
  const double eps_tol = atol + rtol * rNorm;

  if (beta == 0) {
    st->niter = 0; st->solved = 1; st->inconsistent = 0;
    st->timer = now_s() - t0;
    snprintf(st->status, sizeof(st->status), "x is a zero-residual solution");
    if (warm_start) K(khip_axpy(ctx, n, 1.0, dx, x));
    ws->warm_start = false;
    ws->box.publish();
    return KHIP_OK;
  }

  const int mem = (int)c.size();                                                   // :188
  int npass = 0;
  int64_t iter = 0;
  int inner_iter = 0;
  const int64_t itmax = o.itmax == 0 ? 2 * global_rows(ctx, A, n) : o.itmax;   // 2n of the GLOBAL system on every rank
  int64_t inner_itmax = itmax;
  const double btol = std::pow(kEps, 0.75);                                        // :195

  bool breakdown = false, inconsistent = false;
  // :191-192   pass  k  ‖rₖ‖  hₖ₊₁.ₖ  timer ; the first row shows "✗ ✗ ✗ ✗" in the h column
  if (verbose > 0) klogf(o.log_fd, " pass      k     \xe2\x80\x96r\xe2\x82\x96\xe2\x80\x96   h\xe2\x82\x96\xe2\x82\x8a\xe2\x82\x81.\xe2\x82\x96  timer\n");
  if (kdisplay(iter, verbose)) klogf(o.log_fd, "%5d  %5lld  %7.1e  \xe2\x9c\x97 \xe2\x9c\x97 \xe2\x9c\x97 \xe2\x9c\x97  %.2fs\n", npass, (long long)iter, rNorm, now_s() - t0);

  bool solved = rNorm <= eps_tol;
  bool tired = iter >= itmax;
  bool inner_tired = inner_iter >= inner_itmax;
  bool user_requested_exit = false, overtimed = false;
  const char *status = "unknown";

  // ---------------------------------------------------------------------------------------------------------------------
  // variant 2: s-step (communication-avoiding) GMRES, opt-in (Hoemmen 2010, monomial basis).  Per block of s
  // inner iterations: the s products z_j = A z_{j-1} / theta (z_0 = the last basis vector), ONE batched classical Gram-Schmidt
  // pass of all s vectors against the k basis vectors, repeated once (CGS2), and a CholeskyQR2 of the block -- 2 + 2
  // reductions per s iterations on N GPUs (each one all-reduce per 64 scalars) instead of 3 per iteration (variant 1) or k + 1
  // (the reference's modified Gram-Schmidt).  With z_j = Q+ g_j (g_j = the Gram-Schmidt and QR coefficients, g_0 = e_k) and
  // A z_j = theta z_{j+1}, the Hessenberg columns follow on the host: h_c = (theta g_{j+1} - sum_{i<c} g_j[i] h_i) / g_j[c],
  // c = k + j; Givens rotations, the residual estimate and the stopping test column by column as in gmres!.  The residual at
  // a restart is the true one (no Givens estimate in the normalisation).  Same Krylov spaces as gmres!, different rounding
  // and a basis whose conditioning grows with s (s <= 8): own parity budget in the tests.
  if (o.variant == 2 && !(solved || tired)) {
    const int sblk = ctx->tune.gmres_sstep < 1 ? 4 : (ctx->tune.gmres_sstep > 8 ? 8 : ctx->tune.gmres_sstep);
    const bool multi = comm_nranks(ctx) > 1;
    // checked BEFORE the first cycle (k never exceeds mem): a solve must not fail after x was already updated
    if (((mem + 3) & ~3) * sblk > kResultSlots)
      return ws->box.fail(KHIP_ERR_UNSUPPORTED, "gmres variant 2: memory x s too large for the device scalar ring (memory x s <= 256)");
    // the padding slots between the columns of a batch are shipped by the all-reduce as well: keep them zero, not stale
    auto zero_slots = [&](int slot, int count) -> int {
      if (!multi) return KHIP_OK;
      KHIP_CHECK_HIP(hipMemsetAsync(ctx->results_dd + slot, 0, sizeof(dd) * (size_t)count, ctx->stream));
      return KHIP_OK;
    };
    auto allreduce_slots = [&](int slot, int count) -> int {            // device-side all-reduce of a slot range, 64 scalars at a time
      for (int b0 = 0; b0 < count; b0 += kMaxRedOut) {
        const int cnt = count - b0 < kMaxRedOut ? count - b0 : kMaxRedOut;
        KHIP_TRY(comm_allreduce_dd_device(ctx, slot + b0, cnt));
      }
      return KHIP_OK;
    };
    double theta = 0.0;
    std::vector<std::vector<double>> Hc;       // Hc[c-1]: column c of the Hessenberg matrix of this cycle (c + 1 entries), unrotated
    std::vector<std::vector<double>> Rr;       // rotated columns (upper triangular part)
    std::vector<double> cs, sn, zr, tmp;
    while (!(solved || tired || overtimed)) {
      // cycle start: w holds the residual b - A x, beta its norm (first cycle: computed above)
      if (npass >= 1) {
        K(apply_op(ctx, A, x, w));
        K(khip_axpby(ctx, n, 1.0, b, -1.0, w));
        K(khip_nrm2(ctx, n, w, &beta));
        if (beta == 0.0) { solved = true; rNorm = 0.0; break; }
      }
      npass = npass + 1;
      K(khip_fill(ctx, n, xr, 0.0));
      K(khip_divcopy(ctx, n, V[0], w, beta));
      Hc.clear(); Rr.clear(); cs.clear(); sn.clear();
      zr.assign(1, beta);
      int k = 1, cols = 0;                     // basis vectors held, Hessenberg columns processed in this cycle
      const int64_t lim = (int64_t)mem < inner_itmax ? (int64_t)mem : inner_itmax;
      bool block_failed = false;
      while (cols < lim && !solved && !overtimed) {
        int sb = (int)std::min<int64_t>(sblk, lim - cols);
        if (block_failed) sb = 1;
        if ((int)V.size() < k + sb) K(gmres_grow_basis(ws, k + sb - (int)V.size()));
        // (1) matrix powers
        for (int j = 1; j <= sb; ++j) {
          K(apply_op(ctx, A, V[k - 2 + j], V[k - 1 + j]));
          if (theta == 0.0) {                  // once per solve: ||A v_1|| as the scale of the monomial basis
            K(khip_nrm2(ctx, n, V[k - 1 + j], &theta));
            if (!(theta > 0.0)) theta = 1.0;
          }
          K(khip_scal(ctx, n, 1.0 / theta, V[k - 1 + j]));
        }
        // (2) block CGS2 against V[0 .. k-1]: all dots of a pass first (one reduction), then all updates
        const int kpad = (k + 3) & ~3;
        std::vector<double> Cm((size_t)k * sb, 0.0);                       // C[i + k j], both passes accumulated
        for (int pass = 0; pass < 2; ++pass) {
          const int sp = take_slots(ctx, kpad * sb);
          K(zero_slots(sp, kpad * sb));
          for (int j = 0; j < sb; ++j) K(launch_multi_dot(ctx, n, k, V.data(), V[k + j], sp + j * kpad));
          if (multi) K(allreduce_slots(sp, kpad * sb));
          for (int j = 0; j < sb; ++j) K(launch_multi_axpy_dev(ctx, n, k, ctx->results + sp + j * kpad, V.data(), V[k + j]));
          tmp.resize((size_t)(kpad * sb));
          K(fetch_results(ctx, sp, kpad * sb, tmp.data(), /*already_global=*/multi));
          for (int j = 0; j < sb; ++j)
            for (int i = 0; i < k; ++i) Cm[(size_t)i + (size_t)k * j] += tmp[(size_t)j * kpad + i];
        }
        // (3) CholeskyQR2 of the block Z = V[k .. k+sb-1]
        const int spad = (sb + 3) & ~3;
        std::vector<double> Racc((size_t)sb * sb, 0.0), Rk((size_t)sb * sb), Gm((size_t)sb * sb), Rt((size_t)sb * sb);
        for (int i = 0; i < sb; ++i) Racc[(size_t)i * sb + i] = 1.0;       // row-major upper triangular
        bool qr_ok = true;
        for (int round = 0; round < 2 && qr_ok; ++round) {
          const int slotG = take_slots(ctx, spad * sb);
          K(zero_slots(slotG, spad * sb));
          for (int j = 0; j < sb; ++j) K(launch_multi_dot(ctx, n, sb, V.data() + k, V[k + j], slotG + j * spad));
          if (multi) K(allreduce_slots(slotG, spad * sb));
          tmp.resize((size_t)(spad * sb));
          K(fetch_results(ctx, slotG, spad * sb, tmp.data(), /*already_global=*/multi));
          for (int j = 0; j < sb; ++j)
            for (int i = 0; i < sb; ++i) Gm[(size_t)i * sb + j] = tmp[(size_t)j * spad + i];
          // Cholesky G = R' R, R upper (row-major)
          for (int i = 0; i < sb && qr_ok; ++i)
            for (int j = i; j < sb; ++j) {
              double sum = Gm[(size_t)i * sb + j];
              for (int l = 0; l < i; ++l) sum -= Rk[(size_t)l * sb + i] * Rk[(size_t)l * sb + j];
              if (j == i) {
                if (!(sum > 0.0) || !std::isfinite(sum) || (round == 0 && sum <= 1e-14 * Gm[(size_t)i * sb + i])) { qr_ok = false; break; }
                Rk[(size_t)i * sb + i] = std::sqrt(sum);
              } else {
                Rk[(size_t)i * sb + j] = sum / Rk[(size_t)i * sb + i];
              }
            }
          if (!qr_ok) break;
          for (int i = 0; i < sb; ++i) for (int j = 0; j < i; ++j) Rk[(size_t)i * sb + j] = 0.0;
          // Z <- Z R^-1, column by column: z_j <- (z_j - sum_{i<j} R_ij z_i') / R_jj
          for (int j = 0; j < sb; ++j) {
            if (j > 0) {
              std::vector<double> coef((size_t)j);
              for (int i = 0; i < j; ++i) coef[(size_t)i] = -Rk[(size_t)i * sb + j];
              K(khip_multi_axpy(ctx, n, j, coef.data(), V.data() + k, V[k + j]));
            }
            K(khip_scal(ctx, n, 1.0 / Rk[(size_t)j * sb + j], V[k + j]));
          }
          for (int i = 0; i < sb; ++i)                                      // Racc <- Rk * Racc
            for (int j = 0; j < sb; ++j) {
              double sum = 0.0;
              for (int l = i; l <= j; ++l) sum += Rk[(size_t)i * sb + l] * Racc[(size_t)l * sb + j];
              Rt[(size_t)i * sb + j] = sum;
            }
          Racc = Rt;
        }
        if (!qr_ok) {
          // the block lost rank (an exhausted Krylov space, or a monomial basis too ill-conditioned for this s): redo from
          // here one vector at a time; a single vector that cannot be normalised is a (lucky) breakdown
          if (sb == 1) { breakdown = true; break; }
          block_failed = true;
          continue;
        }
        // (4) Hessenberg columns, Givens rotations, residual estimates
        auto gcoef = [&](int j, int i) -> double {       // g_j[i], i 1-based basis index, j = 0 .. sb (j = 0: e_k)
          if (j == 0) return i == k ? 1.0 : 0.0;
          if (i <= k) return Cm[(size_t)(i - 1) + (size_t)k * (j - 1)];
          const int r = i - k - 1;                        // row of Racc
          return r <= j - 1 ? Racc[(size_t)r * sb + (j - 1)] : 0.0;
        };
        for (int j = 0; j < sb && !solved; ++j) {
          const int c = k + j;                            // column: A q_c in terms of q_1 .. q_{c+1}
          std::vector<double> h((size_t)c + 1, 0.0);
          for (int i = 1; i <= c + 1; ++i) h[(size_t)i - 1] = theta * gcoef(j + 1, i);
          for (int i = 1; i < c; ++i) {
            const double gi = gcoef(j, i);
            if (gi == 0.0) continue;
            const std::vector<double> &hi = Hc[(size_t)i - 1];
            for (size_t l = 0; l < hi.size(); ++l) h[l] -= gi * hi[l];
          }
          const double gc = gcoef(j, c);
          for (double &v : h) v /= gc;
          Hc.push_back(h);
          // rotate
          std::vector<double> rcol(h);
          for (int i = 0; i < c - 1; ++i) {
            const double t1 = cs[(size_t)i] * rcol[(size_t)i] + sn[(size_t)i] * rcol[(size_t)i + 1];
            rcol[(size_t)i + 1] = sn[(size_t)i] * rcol[(size_t)i] - cs[(size_t)i] * rcol[(size_t)i + 1];
            rcol[(size_t)i] = t1;
          }
          double cc, ss, rho;
          sym_givens(rcol[(size_t)c - 1], rcol[(size_t)c], cc, ss, rho);
          rcol[(size_t)c - 1] = rho;
          rcol.resize((size_t)c);
          Rr.push_back(rcol);
          cs.push_back(cc); sn.push_back(ss);
          const double zeta_next = ss * zr[(size_t)c - 1];
          zr[(size_t)c - 1] = cc * zr[(size_t)c - 1];
          zr.push_back(zeta_next);
          rNorm = std::fabs(zeta_next);
          if (o.history) ws->box.push(rNorm);
          cols = cols + 1;
          solved = (rNorm <= eps_tol) || (rNorm + 1.0 <= 1.0);
          if (kdisplay(iter + cols, verbose))
            klogf(o.log_fd, "%5d  %5lld  %7.1e  %7.1e  %.2fs\n", npass, (long long)(iter + cols), rNorm, h[(size_t)c], now_s() - t0);
          if (iter + cols >= itmax) break;
        }
        k += sb;
        block_failed = false;
        overtimed = time_limit_reached(ctx, now_s() - t0, timemax);
        if (iter + cols >= itmax) break;
      }
      // (5) y from the rotated triangular system, x += sum y_i q_i
      if (cols > 0) {
        std::vector<double> y(zr.begin(), zr.begin() + cols);
        for (int i = cols - 1; i >= 0; --i) {
          for (int j = i + 1; j < cols; ++j) y[(size_t)i] -= Rr[(size_t)j][(size_t)i] * y[(size_t)j];
          const double d = Rr[(size_t)i][(size_t)i];
          if (std::fabs(d) <= btol) { y[(size_t)i] = 0.0; inconsistent = true; } else y[(size_t)i] /= d;
        }
        K(khip_multi_axpy(ctx, n, cols, y.data(), V.data(), xr));
        K(khip_axpy(ctx, n, 1.0, xr, x));
      }
      inner_itmax = inner_itmax - cols;
      iter = iter + cols;
      tired = iter >= itmax;
      if (breakdown) break;
      if (!overtimed) overtimed = time_limit_reached(ctx, now_s() - t0, timemax);
    }
    if (breakdown && !solved) { solved = rNorm <= eps_tol; }
  }

  while (o.variant != 2 && !(solved || tired || breakdown || user_requested_exit || overtimed)) {
    int nr = 0;
    // :211-213 zero-fills the mem basis vectors here (8 n mem bytes of writes per cycle).  Every V[k] is written by
    // the kdivcopy! of :231 / :325 before anything reads it and this ABI has no accessor for the basis, so the fill is
    // unobservable and skipped.
    std::fill(s.begin(), s.end(), 0.0);
    std::fill(c.begin(), c.end(), 0.0);
    std::fill(R.begin(), R.end(), 0.0);
    std::fill(z.begin(), z.end(), 0.0);

    if (restart) {
      K(khip_fill(ctx, n, xr, 0.0));
      if (npass >= 1) {
        K(apply_op(ctx, A, x, w));
        K(khip_axpby(ctx, n, 1.0, b, -1.0, w));
        if (!MisI) K(apply_op(ctx, M, w, r0));
      }
    }

    K(khip_nrm2(ctx, n, r0, &beta));                                               // :229
    z[0] = beta;
    K(khip_divcopy(ctx, n, V[0], r0, rNorm));                                      // :231 divides by rNorm (estimate), not beta

    npass = npass + 1;
    ws->inner_iter = 0;
    inner_tired = false;
    bool spec_done = false;                    // w already holds A * V[inner_iter] (enqueued by the look-ahead)

    while (!(solved || inner_tired || breakdown || user_requested_exit || overtimed)) {
      ws->inner_iter = ws->inner_iter + 1;
      inner_iter = ws->inner_iter;

      if (!restart && (inner_iter > mem)) {                                        // :244-252
        for (int i = 0; i < inner_iter; ++i) R.push_back(0.0);
        s.push_back(0.0);
        c.push_back(0.0);
      }

      double *Vk = V[inner_iter - 1];
      double *pp = NisI ? Vk : ws->p;
      if (!spec_done) {
        if (!NisI) K(apply_op(ctx, N, Vk, pp));
        K(apply_op(ctx, A, pp, w));                                                // :257
        if (!MisI) K(apply_op(ctx, M, w, q));
      }
      spec_done = false;

      double Hbis;
      if (o.variant == 1) {
        // CGS2 (classical Gram-Schmidt, applied twice; opt-in, NOT the reference's recurrence -- src/gmres.jl:259-271 is the
        // modified form): h = V_k' q in one reduction per four basis vectors, q -= V_k h in one pass with device-resident
        // coefficients, the same again, R_k += h + h2.  Two all-reduces of k scalars and one of ||q||^2 per inner iteration on
        // N GPUs instead of k + 1 single-scalar ones.  Same Krylov space, different rounding: own parity budget in the tests.
        const int k = inner_iter;
        const int kpad = (k + 3) & ~3;
        if (2 * kpad + 1 > kResultSlots) return ws->box.fail(KHIP_ERR_UNSUPPORTED, "gmres variant 1: too many basis vectors for the device scalar ring");
        // with several ranks the k coefficients of a step travel in ONE all-reduce of at most kMaxRedOut scalars: refuse a
        // longer basis before the solve starts rather than abort in the middle of a cycle
        if (comm_nranks(ctx) > 1 && mem > kMaxRedOut)
          return ws->box.fail(KHIP_ERR_UNSUPPORTED, "gmres variant 1 on several ranks: memory must not exceed 64 (one all-reduce carries the coefficients of a step)");
        const int slot = take_slots(ctx, 2 * kpad + 1);
        const bool multi = comm_nranks(ctx) > 1;
        for (int pass = 0; pass < 2; ++pass) {
          const int sp = slot + pass * kpad;
          K(launch_multi_dot(ctx, n, k, V.data(), q, sp));
          if (multi) K(comm_allreduce_dd_device(ctx, sp, k));
          K(launch_multi_axpy_dev(ctx, n, k, ctx->results + sp, V.data(), q));
        }
        K(launch_nrm2sq(ctx, n, q, slot + 2 * kpad));
        if (multi) K(comm_allreduce_dd_device(ctx, slot + 2 * kpad, 1));
        if (ws->look.size() < (size_t)(2 * kpad + 1)) ws->look.resize((size_t)(2 * kpad + 1));
        double *tmp = ws->look.data();
        K(fetch_results(ctx, slot, 2 * kpad + 1, tmp, /*already_global=*/multi));
        for (int i = 0; i < k; ++i) R[nr + i] = tmp[i] + tmp[kpad + i];
        Hbis = std::sqrt(tmp[2 * kpad]);
      } else if (look && inner_iter + 1 <= kResultSlots) {
        // ONE-STEP LOOK-AHEAD: the MGS cascade leaves its coefficients and ||q||^2 on the device.  Before the host
        // waits for them, the next basis vector V[k+1] = q / ||q|| (device scalar) and its product A V[k+1] are
        // enqueued, so the queue stays busy while the host applies the Givens rotations and tests convergence.
        // If the test says stop, that work was wasted and touched nothing that is read again.  Same operations on the
        // same values as the plain sequence: bit-identical.
        int slot = 0;
        K(mgs_enqueue(ctx, n, inner_iter, V.data(), q, &slot));
        K(results_copy_begin(ctx, slot, inner_iter + 1));
        const int64_t lim = restart ? ((int64_t)mem < inner_itmax ? (int64_t)mem : inner_itmax) : inner_itmax;
        if (inner_iter < lim && (int)V.size() > inner_iter && (restart || inner_iter < mem)) {
          K(launch_divcopy_dev(ctx, n, ctx->results + slot + inner_iter, q, V[inner_iter]));   // :325
          K(apply_op(ctx, A, V[inner_iter], w));                                                // the next :257
          spec_done = true;
        }
        if (ws->look.size() < (size_t)inner_iter + 1) ws->look.resize((size_t)inner_iter + 1);   // only when the basis grew (restart = false)
        double *tmp = ws->look.data();
        K(results_copy_end(ctx, inner_iter + 1, tmp));
        for (int i = 0; i < inner_iter; ++i) R[nr + i] = tmp[i];
        Hbis = std::sqrt(tmp[inner_iter]);
      } else if (fused) {
        // MGS cascade with device-resident coefficients; ||q|| of :274 comes out of the last pass
        K(khip_mgs(ctx, n, inner_iter, V.data(), q, &R[nr], reorth ? nullptr : &Hbis, 0));
        if (reorth) K(khip_mgs(ctx, n, inner_iter, V.data(), q, &R[nr], &Hbis, 1));
      } else {
        for (int i = 0; i < inner_iter; ++i) {                                     // :259-262
          K(khip_dot(ctx, n, V[i], q, &R[nr + i]));
          K(khip_axpy(ctx, n, -R[nr + i], V[i], q));
        }
        if (reorth) {                                                              // :265-271
          for (int i = 0; i < inner_iter; ++i) {
            double Htmp;
            K(khip_dot(ctx, n, V[i], q, &Htmp));
            R[nr + i] += Htmp;
            K(khip_axpy(ctx, n, -Htmp, V[i], q));
          }
        }
        K(khip_nrm2(ctx, n, q, &Hbis));                                            // :274
      }

      for (int i = 0; i < inner_iter - 1; ++i) {                                   // :280-284
        const double Rtmp = c[i] * R[nr + i] + s[i] * R[nr + i + 1];
        R[nr + i + 1] = s[i] * R[nr + i] - c[i] * R[nr + i + 1];
        R[nr + i] = Rtmp;
      }
      const int k = inner_iter - 1;
      sym_givens(R[nr + k], Hbis, c[k], s[k], R[nr + k]);                          // :289

      const double zeta_next = s[k] * z[k];                                        // :292-293
      z[k] = c[k] * z[k];

      rNorm = std::fabs(zeta_next);
      if (o.history) ws->box.push(rNorm);
      nr = nr + inner_iter;

      const bool resid_decrease_mach = (rNorm + 1.0 <= 1.0);
      if (o.callback) { ws->box.publish(); user_requested_exit = o.callback(ws, o.callback_data) != 0; }
      const bool resid_decrease_lim = rNorm <= eps_tol;
      breakdown = Hbis <= btol;
      solved = resid_decrease_lim || resid_decrease_mach;
      if (restart) {
        const int64_t lim = (int64_t)mem < inner_itmax ? (int64_t)mem : inner_itmax;
        inner_tired = inner_iter >= lim;
      } else {
        inner_tired = inner_iter >= inner_itmax;
      }
      overtimed = time_limit_reached(ctx, now_s() - t0, timemax);
      if (kdisplay(iter + inner_iter, verbose))                                     // :315
        klogf(o.log_fd, "%5d  %5lld  %7.1e  %7.1e  %.2fs\n", npass, (long long)(iter + inner_iter), rNorm, Hbis, now_s() - t0);

      if (!(solved || inner_tired || breakdown || user_requested_exit || overtimed)) {
        if (!restart && (inner_iter >= mem)) {                                     // :319-324
          if ((int)V.size() <= inner_iter) K(gmres_grow_basis(ws, ws->grow ? inner_iter + 1 - (int)V.size() : (mem > 4 ? mem : 4)));
          if ((int)z.size() <= inner_iter) z.resize(inner_iter + 1, 0.0);
        }
        if (!spec_done) K(khip_divcopy(ctx, n, V[inner_iter], q, Hbis));           // :325 (already enqueued by the look-ahead)
        z[inner_iter] = zeta_next;
      } else {
        spec_done = false;                                                         // the speculative product is abandoned
      }
    }

    // back-substitution R y = z (:331-345), y aliases z
    std::vector<double> &y = z;
    for (int i = inner_iter; i >= 1; --i) {
      int pos = nr + i - inner_iter;          // 1-based position of r_{i,k}
      for (int j = inner_iter; j >= i + 1; --j) {
        y[i - 1] = y[i - 1] - R[pos - 1] * y[j - 1];
        pos = pos - j + 1;
      }
      if (std::fabs(R[pos - 1]) <= btol) {
        y[i - 1] = 0.0;
        inconsistent = true;
      } else {
        y[i - 1] = y[i - 1] / R[pos - 1];
      }
    }

    if (fused) {
      K(khip_multi_axpy(ctx, n, inner_iter, y.data(), V.data(), xr));              // :348-350 in one pass
    } else {
      for (int i = 0; i < inner_iter; ++i) K(khip_axpy(ctx, n, y[i], V[i], xr));
    }
    if (!NisI) {
      K(khip_copy(ctx, n, ws->p, xr));
      K(apply_op(ctx, N, ws->p, xr));
    }
    if (restart) K(khip_axpy(ctx, n, 1.0, xr, x));                                 // :355

    inner_itmax = inner_itmax - inner_iter;
    iter = iter + inner_iter;
    tired = iter >= itmax;
    overtimed = time_limit_reached(ctx, now_s() - t0, timemax);
  }

  if (verbose > 0) { klogf(o.log_fd, "\n"); klog_flush(o.log_fd); }                                                           // src/gmres.jl:364
  if (tired) status = "maximum number of iterations exceeded";
  if (solved) status = "solution good enough given atol and rtol";
  if (inconsistent) status = "found approximate least-squares solution";
  if (user_requested_exit) status = "user-requested exit";
  if (overtimed) status = "time limit exceeded";

  if (warm_start && !restart) K(khip_axpy(ctx, n, 1.0, dx, x));
  ws->warm_start = false;
  K(khip_ctx_sync(ctx));

  st->niter = (int)iter;
  st->solved = solved;
  st->inconsistent = inconsistent;
  st->timer = now_s() - t0;
  st->allocation_timer += take_alloc_seconds();                 // lazy allocations of this solve (allocate_if)
  snprintf(st->status, sizeof(st->status), "%s", status);
  ws->box.publish();
  return KHIP_OK;
}

}  // extern "C"

// ================================================================== BiCGSTAB ====
struct khip_bicgstab_workspace {
  khip_ctx *ctx;
  int64_t m, n;
  double *dx = nullptr, *x = nullptr, *r = nullptr, *p = nullptr, *v = nullptr, *s = nullptr, *qd = nullptr,
         *yz = nullptr, *t = nullptr;
  bool warm_start = false;
  Borrowed borrowed;                   // vectors of a caller's BicgstabWorkspace (khip_bicgstab_workspace_adopt)
  StatsBox box;
  // device-resident loop state (fused = 2), allocated on first use
  BicgDevState *dev_state = nullptr;
  BicgDevState *snap = nullptr;        // pinned host snapshots [2]
  double *hist_dev = nullptr;
  hipEvent_t snap_ev[2] = {nullptr, nullptr};
};

namespace {

// The loop of src/bicgstab.jl:213-253 (M = N = I, CSR operator, no callback) with rho, alpha, omega, beta and the
// stopping tests on the device: five passes per iteration, each carrying its sequence number; see cg_device_loop.
int bicgstab_device_loop(khip_bicgstab_workspace *ws, const khip_csr *A, const double *c, double rho0, double rNorm0,
                         double eps_tol, int64_t itmax, bool history, double t0, double timemax, BicgDevState *out,
                         bool *overtimed) {
  khip_ctx *ctx = ws->ctx;
  const int64_t n = ws->n;
  if (!ws->dev_state) {
    KHIP_CHECK_HIP(hipMalloc(&ws->dev_state, sizeof(BicgDevState)));
    KHIP_CHECK_HIP(hipHostMalloc(reinterpret_cast<void **>(&ws->snap), 2 * sizeof(BicgDevState), hipHostMallocDefault));
    KHIP_CHECK_HIP(hipMalloc(&ws->hist_dev, sizeof(double) * (size_t)kHistWindowMax));
    for (auto &e : ws->snap_ev) KHIP_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  long long window = ctx->tune.hist_window;
  if (window < kDevChunk) window = kDevChunk;
  if (window > kHistWindowMax) window = kHistWindowMax;
  BicgDevState *dev = ws->dev_state;
  BicgDevState h;
  memset(&h, 0, sizeof(h));
  h.rho = rho0; h.alpha = 1.0; h.omega = 1.0; h.rNorm = rNorm0; h.eps_tol = eps_tol;
  h.stop_seq = kSeqNever;
  h.hist = history ? ws->hist_dev : nullptr;
  h.hist_cap = window;
  KHIP_CHECK_HIP(hipMemcpyAsync(dev, &h, sizeof(h), hipMemcpyHostToDevice, ctx->stream));
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  double *x = ws->x, *r = ws->r, *p = ws->p, *v = ws->v, *s = ws->s, *t = ws->qd;
  int64_t enq = 0;
  long long hist_base = 0;
  std::vector<double> win;
  auto drain_history = [&](long long upto_iter) -> int {
    const long long cnt = upto_iter - hist_base;
    if (!history || cnt <= 0) return KHIP_OK;
    win.resize((size_t)cnt);
    KHIP_CHECK_HIP(hipMemcpy(win.data(), ws->hist_dev, sizeof(double) * (size_t)cnt, hipMemcpyDeviceToHost));
    for (double val : win) ws->box.push(val);
    return KHIP_OK;
  };
  int rc = KHIP_OK;
  bool stopped = false;
  for (int chunk = 0; !stopped; ++chunk) {
    const int64_t cnt = std::min<int64_t>(kDevChunk, itmax - enq);
    if (history && enq + cnt - hist_base > window) {
      KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
      BicgDevState cur;
      KHIP_CHECK_HIP(hipMemcpy(&cur, dev, sizeof(cur), hipMemcpyDeviceToHost));
      if (cur.stop_seq != kSeqNever) break;
      if ((rc = drain_history(cur.iter)) != KHIP_OK) break;
      hist_base = cur.iter;
      KHIP_CHECK_HIP(hipMemcpy(&dev->hist_base, &hist_base, sizeof(hist_base), hipMemcpyHostToDevice));
    }
    for (int64_t i = 0; i < cnt && rc == KHIP_OK; ++i) {
      const long long j = (long long)(enq + i);
      ctx->ctl = SeqCtl{&dev->stop_seq, 5 * j, EPI_BICG_A, dev};
      int slot = take_slots(ctx, 1);
      rc = spmv_any(ctx, A, p, v, slot, c, 0);                                  // :221-223  v = A p ; c.v -> alpha
      if (rc == KHIP_OK && ctx->comm) rc = comm_allreduce_dd_device(ctx, slot, 1);
      ctx->ctl = SeqCtl{};
      if (rc != KHIP_OK) break;
      rc = launch_bicg_sx(ctx, n, 0.0, r, v, p, s, x, dev, 5 * j + 1);               // :224-226
      if (rc != KHIP_OK) break;
      ctx->ctl = SeqCtl{&dev->stop_seq, 5 * j + 2, EPI_BICG_B, dev};
      slot = take_slots(ctx, 2);
      rc = spmv_any(ctx, A, s, t, slot, nullptr, 1);                             // :228-230  t = A s ; t.s, t.t -> omega
      if (rc == KHIP_OK && ctx->comm) rc = comm_allreduce_dd_device(ctx, slot, 2);
      if (rc != KHIP_OK) break;
      ctx->ctl = SeqCtl{&dev->stop_seq, 5 * j + 3, EPI_BICG_C, dev};
      slot = take_slots(ctx, 2);
      rc = launch_bicg_xr(ctx, n, 0.0, s, t, s, c, x, r, slot, dev);                // :231-234, :240 -> beta, tests
      if (rc == KHIP_OK && ctx->comm) rc = comm_allreduce_dd_device(ctx, slot, 2);
      ctx->ctl = SeqCtl{};
      if (rc != KHIP_OK) break;
      rc = launch_bicg_p(ctx, n, 0.0, 0.0, v, r, p, dev, 5 * j + 4);                // :236-237
    }
    ctx->ctl = SeqCtl{};
    if (rc != KHIP_OK) break;
    enq += cnt;
    const int b = chunk & 1;
    KHIP_CHECK_HIP(hipMemcpyAsync(&ws->snap[b], dev, sizeof(BicgDevState), hipMemcpyDeviceToHost, ctx->stream));
    KHIP_CHECK_HIP(hipEventRecord(ws->snap_ev[b], ctx->stream));
    if (chunk >= 1) {
      KHIP_CHECK_HIP(hipEventSynchronize(ws->snap_ev[b ^ 1]));
      if (ws->snap[b ^ 1].stop_seq != kSeqNever) stopped = true;
    }
    if (enq >= itmax) stopped = true;
    if (!stopped && time_limit_reached(ctx, now_s() - t0, timemax)) { *overtimed = true; stopped = true; }
  }
  ctx->ctl = SeqCtl{};
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  if (rc != KHIP_OK) return rc;
  KHIP_CHECK_HIP(hipMemcpy(out, dev, sizeof(BicgDevState), hipMemcpyDeviceToHost));
  return drain_history(out->iter);
}

}  // namespace

extern "C" {

int khip_bicgstab_workspace_create(khip_ctx *ctx, int64_t m, int64_t n, khip_bicgstab_workspace **out) {
  KHIP_REQUIRE(ctx && out && m >= 0 && n >= 0, "bicgstab_workspace_create: bad argument");
  khip_bicgstab_workspace *ws = new khip_bicgstab_workspace();
  ws->ctx = ctx; ws->m = m; ws->n = n;
  (void)take_alloc_seconds();
  int rc = 0;                                                        // x, r, p, v, s, qd (src/krylov_workspaces.jl:1605-1623)
  for (double **v : {&ws->x, &ws->r, &ws->p, &ws->v, &ws->s, &ws->qd})
    if (!rc) rc = alloc_vec(ctx, n, v);
  if (rc) { khip_bicgstab_workspace_destroy(ws); return rc; }
  ws->box.st.allocation_timer = take_alloc_seconds();
  *out = ws;
  return KHIP_OK;
}

// BicgstabWorkspace on the CALLER's six vectors x, r, p, v, s, qd (src/krylov_workspaces.jl:1568-1582); yz, t, dx are handed
// over with khip_bicgstab_workspace_adopt_vector when the caller has allocated them (src/bicgstab.jl:148-150).
int khip_bicgstab_workspace_adopt(khip_ctx *ctx, int64_t m, int64_t n, double *x, double *r, double *p, double *v, double *s,
                                  double *qd, khip_bicgstab_workspace **out) {
  KHIP_REQUIRE(ctx && out && m >= 0 && n >= 0, "bicgstab_workspace_adopt: bad argument");
  double *six[6] = {x, r, p, v, s, qd};
  for (int i = 0; i < 6 && n > 0; ++i) {
    KHIP_REQUIRE(six[i] != nullptr, "bicgstab_workspace_adopt: x, r, p, v, s, qd must be device vectors of n entries");
    for (int j = 0; j < i; ++j) KHIP_REQUIRE(six[i] != six[j], "bicgstab_workspace_adopt: x, r, p, v, s, qd must be distinct");
  }
  khip_bicgstab_workspace *ws = new khip_bicgstab_workspace();
  ws->ctx = ctx; ws->m = m; ws->n = n;
  ws->x = x; ws->r = r; ws->p = p; ws->v = v; ws->s = s; ws->qd = qd;
  for (double *u : six) ws->borrowed.add(u);
  (void)take_alloc_seconds();
  ws->box.st.allocation_timer = 0.0;
  *out = ws;
  return KHIP_OK;
}

int khip_bicgstab_workspace_adopt_vector(khip_bicgstab_workspace *ws, const char *name, double *ptr) {
  KHIP_REQUIRE(ws && name, "bicgstab_workspace_adopt_vector: null argument");
  struct { const char *k; double **slot; bool required; } tab[] = {
      {"x", &ws->x, true}, {"r", &ws->r, true}, {"p", &ws->p, true}, {"v", &ws->v, true}, {"s", &ws->s, true},
      {"qd", &ws->qd, true}, {"yz", &ws->yz, false}, {"t", &ws->t, false}, {"dx", &ws->dx, false}};
  for (auto &e : tab)
    if (strcmp(e.k, name) == 0) {
      KHIP_REQUIRE(ptr || !e.required, "bicgstab_workspace_adopt_vector: x, r, p, v, s, qd cannot be emptied");
      if (const char *other = held_elsewhere(tab, e.slot, ptr, nullptr)) {
        set_error("bicgstab_workspace_adopt_vector: the pointer for '%s' already is the workspace's '%s' (every vector needs its own storage)", name, other);
        return KHIP_ERR_INVALID;
      }
      adopt_into(ws->ctx, ws->borrowed, e.slot, ptr);
      return KHIP_OK;
    }
  set_error("bicgstab_workspace_adopt_vector: unknown vector '%s' (x, r, p, v, s, qd, yz, t, dx)", name);
  return KHIP_ERR_INVALID;
}

int khip_bicgstab_workspace_destroy(khip_bicgstab_workspace *ws) {
  if (!ws) return KHIP_OK;
  for (double *v : {ws->dx, ws->x, ws->r, ws->p, ws->v, ws->s, ws->qd, ws->yz, ws->t}) free_unless_borrowed(ws->ctx, ws->borrowed, v);
  if (ws->dev_state) (void)hipFree(ws->dev_state);
  if (ws->snap) (void)hipHostFree(ws->snap);
  if (ws->hist_dev) (void)hipFree(ws->hist_dev);
  for (auto e : ws->snap_ev) if (e) (void)hipEventDestroy(e);
  delete ws;
  return KHIP_OK;
}

int khip_bicgstab_warm_start(khip_bicgstab_workspace *ws, const double *x0) {
  KHIP_REQUIRE(ws && x0, "bicgstab_warm_start: null argument");
  if (!ws->dx) KHIP_TRY(alloc_vec(ws->ctx, ws->n, &ws->dx));
  if (x0 != ws->dx) KHIP_TRY(khip_copy(ws->ctx, ws->n, ws->dx, x0));
  ws->warm_start = true;
  return KHIP_OK;
}

double *khip_bicgstab_solution(khip_bicgstab_workspace *ws) { return ws ? ws->x : nullptr; }
const khip_stats *khip_bicgstab_stats(khip_bicgstab_workspace *ws) { return ws ? &ws->box.st : nullptr; }
int khip_bicgstab_last_path(khip_bicgstab_workspace *ws) { return ws ? ws->box.path : -1; }
size_t khip_bicgstab_workspace_bytes(khip_bicgstab_workspace *ws) {
  if (!ws) return 0;
  size_t cnt = 0;
  for (double *v : {ws->dx, ws->x, ws->r, ws->p, ws->v, ws->s, ws->qd, ws->yz, ws->t}) cnt += v ? 1 : 0;
  return cnt * sizeof(double) * (size_t)ws->n;
}

int khip_bicgstab_solve(khip_bicgstab_workspace *ws, const khip_operator *A, const khip_operator *M,
                        const khip_operator *N, const double *b, const double *c, const khip_options *opts_in) {
  KHIP_REQUIRE(ws && A && b, "bicgstab_solve: null argument");
  khip_ctx *ctx = ws->ctx;
  khip_options o = opts_in ? *opts_in : khip_default_options();
  const double t0 = now_s();
  const double timemax = timemax_of(o);
  const int64_t n = ws->n;
  khip_stats *st = &ws->box.st;
  const double atol = tol_or_default(o.atol), rtol = tol_or_default(o.rtol);
  const bool fused = o.fused != 0;
  const int verbose = o.verbose;
  (void)take_alloc_seconds();
  if (verbose > 0) klogf(o.log_fd, "BICGSTAB: system of size %lld\n", (long long)n);                                       // src/bicgstab.jl:135
  if (ws->m != ws->n) return ws->box.fail(KHIP_ERR_INVALID, "System must be square");
  if (o.variant != 0) return ws->box.fail(KHIP_ERR_INVALID, "bicgstab: options.variant must be 0 (there is no other recurrence)");
  if (!c) c = b;                                                                   // src/bicgstab.jl:105

  const bool MisI = (M == nullptr), NisI = (N == nullptr);
  if (!MisI && !ws->t) K(alloc_vec(ctx, n, &ws->t));
  if (!NisI && !ws->yz) K(alloc_vec(ctx, n, &ws->yz));
  double *dx = ws->dx, *x = ws->x, *r = ws->r, *p = ws->p, *v = ws->v, *s = ws->s;
  const bool warm_start = ws->warm_start;
  ws->box.reset();
  double *q = ws->qd, *d = ws->qd;                                                 // aliasing :153-157
  double *t = MisI ? d : ws->t;
  double *y = NisI ? p : ws->yz;
  double *z = NisI ? s : ws->yz;
  double *r0 = MisI ? r : ws->qd;

  if (warm_start) {
    K(apply_op(ctx, A, dx, r0));
    K(khip_axpby(ctx, n, 1.0, b, -1.0, r0));
  } else {
    K(khip_copy(ctx, n, r0, b));
  }
  K(khip_fill(ctx, n, x, 0.0));
  K(khip_fill(ctx, n, s, 0.0));
  K(khip_fill(ctx, n, v, 0.0));
  if (!MisI) K(apply_op(ctx, M, r0, r));
  K(khip_copy(ctx, n, p, r));

  double alpha = 1.0, omega = 1.0, rho = 1.0;
  double rNorm;
  K(khip_nrm2(ctx, n, r, &rNorm));                                                 // :177
  if (o.history) ws->box.push(rNorm);
  if (rNorm == 0) {
    st->niter = 0; st->solved = 1; st->inconsistent = 0;
    st->timer = now_s() - t0;
    snprintf(st->status, sizeof(st->status), "x is a zero-residual solution");
    if (warm_start) K(khip_axpy(ctx, n, 1.0, dx, x));
    ws->warm_start = false;
    ws->box.publish();
    return KHIP_OK;
  }

  int64_t iter = 0;
  const int64_t itmax = o.itmax == 0 ? 2 * global_rows(ctx, A, n) : o.itmax;   // 2n of the GLOBAL system on every rank
  const double eps_tol = atol + rtol * rNorm;
  // :193-194   k  ‖rₖ‖  |αₖ|  |ωₖ|  timer
  if (verbose > 0) klogf(o.log_fd, "    k     \xe2\x80\x96r\xe2\x82\x96\xe2\x80\x96      |\xce\xb1\xe2\x82\x96|      |\xcf\x89\xe2\x82\x96|  timer\n");
  if (kdisplay(iter, verbose)) klogf(o.log_fd, "%5lld  %7.1e  %8.1e  %8.1e  %.2fs\n", (long long)iter, rNorm, std::fabs(alpha), std::fabs(omega), now_s() - t0);

  double next_rho;
  K(khip_dot(ctx, n, c, r, &next_rho));                                            // :196
  if (next_rho == 0) {
    st->niter = 0; st->solved = 0; st->inconsistent = 0;
    st->timer = now_s() - t0;
    snprintf(st->status, sizeof(st->status), "Breakdown b\xe1\xb4\xb4""c = 0");
    if (warm_start) K(khip_axpy(ctx, n, 1.0, dx, x));
    ws->warm_start = false;
    ws->box.publish();
    return KHIP_OK;
  }

  bool solved = rNorm <= eps_tol;
  bool tired = iter >= itmax;
  bool breakdown = false, user_requested_exit = false, overtimed = false;
  const char *status = "unknown";

  const bool fast = fused && MisI && NisI && !A->apply && A->csr;
  const bool device_loop = fast && o.fused >= 2 && !o.callback && verbose <= 0;      // the log rows need alpha and omega on the host
  ws->box.path = device_loop ? 2 : (fused ? 1 : 0);
  if (device_loop && !(solved || tired)) {
    BicgDevState fin;
    K(bicgstab_device_loop(ws, A->csr, c, next_rho, rNorm, eps_tol, itmax, o.history != 0, t0, timemax, &fin, &overtimed));
    iter = fin.iter;
    rNorm = fin.rNorm;
    solved = fin.solved != 0;
    breakdown = fin.breakdown != 0;
    tired = iter >= itmax;
  }
  while (!device_loop && !(solved || tired || breakdown || user_requested_exit || overtimed)) {
    iter = iter + 1;
    rho = next_rho;

    if (fast) {
      // M = N = I on a CSR operator: the iteration in 5 passes (2 SpMV with their dots fused + 128n bytes of
      // vector work instead of 216n), every elementwise expression unchanged.  q = A p is written straight
      // into v (the reference copies it there, :222, and q's storage is reused for d in the same iteration).
      double cv, two[2];
      int slot = take_slots(ctx, 1);
      K(spmv_any(ctx, A->csr, p, v, slot, c, 0));                              // :221-223  v = A p ; c.v
      K(fetch_results(ctx, slot, 1, &cv));
      alpha = rho / cv;                                                            // :223
      K(launch_bicg_sx(ctx, n, alpha, r, v, p, s, x));                             // :224-226
      slot = take_slots(ctx, 2);
      K(spmv_any(ctx, A->csr, s, t, slot, nullptr, 1));                         // :228-230  t = A s ; t.s ; t.t
      K(fetch_results(ctx, slot, 2, two));
      omega = two[0] / two[1];                                                     // :230
      slot = take_slots(ctx, 2);
      K(launch_bicg_xr(ctx, n, omega, s, t, s, c, x, r, slot));                    // :231-234, :240
      K(fetch_results(ctx, slot, 2, two));
      next_rho = two[0];
      const double beta = (next_rho / rho) * (alpha / omega);                      // :235
      K(launch_bicg_p(ctx, n, omega, beta, v, r, p));                              // :236-237
      rNorm = std::sqrt(two[1]);                                                   // :240
      if (o.history) ws->box.push(rNorm);
      const bool rdm = (rNorm + 1.0 <= 1.0);
      if (o.callback) { ws->box.publish(); user_requested_exit = o.callback(ws, o.callback_data) != 0; }
      solved = (rNorm <= eps_tol) || rdm;
      tired = iter >= itmax;
      breakdown = (alpha == 0 || std::isnan(alpha));
      overtimed = time_limit_reached(ctx, now_s() - t0, timemax);
      continue;
    }

    if (!NisI) K(apply_op(ctx, N, p, y));
    K(apply_op(ctx, A, y, q));                                                     // :221
    if (MisI) K(khip_copy(ctx, n, v, q)); else K(apply_op(ctx, M, q, v));          // :222 (unguarded mul!(v, I, q))
    double cv;
    K(khip_dot(ctx, n, c, v, &cv));
    alpha = rho / cv;                                                              // :223
    if (fused) {
      K(khip_waxpy(ctx, n, s, r, -alpha, v));                                      // :224-225  s = r - alpha v
    } else {
      K(khip_copy(ctx, n, s, r));                                                  // :224
      K(khip_axpy(ctx, n, -alpha, v, s));                                          // :225
    }
    K(khip_axpy(ctx, n, alpha, y, x));                                             // :226
    if (!NisI) K(apply_op(ctx, N, s, z));
    K(apply_op(ctx, A, z, d));                                                     // :228
    if (!MisI) K(apply_op(ctx, M, d, t));
    if (fused) {
      double two[2];
      K(khip_dot2(ctx, n, t, s, two));                                             // :230 both dots in one pass
      omega = two[0] / two[1];
    } else {
      double ts, tt;
      K(khip_dot(ctx, n, t, s, &ts));
      K(khip_dot(ctx, n, t, t, &tt));
      omega = ts / tt;                                                             // :230
    }
    K(khip_axpy(ctx, n, omega, z, x));                                             // :231
    if (fused) {
      K(khip_waxpy(ctx, n, r, s, -omega, t));                                      // :232-233  r = s - omega t
    } else {
      K(khip_copy(ctx, n, r, s));                                                  // :232
      K(khip_axpy(ctx, n, -omega, t, r));                                          // :233
    }
    K(khip_dot(ctx, n, c, r, &next_rho));                                          // :234
    const double beta = (next_rho / rho) * (alpha / omega);                        // :235
    K(khip_axpy(ctx, n, -omega, v, p));                                            // :236
    K(khip_axpby(ctx, n, 1.0, r, beta, p));                                        // :237

    K(khip_nrm2(ctx, n, r, &rNorm));                                               // :240
    if (o.history) ws->box.push(rNorm);

    const bool resid_decrease_mach = (rNorm + 1.0 <= 1.0);
    if (o.callback) { ws->box.publish(); user_requested_exit = o.callback(ws, o.callback_data) != 0; }
    const bool resid_decrease_lim = rNorm <= eps_tol;
    solved = resid_decrease_lim || resid_decrease_mach;
    tired = iter >= itmax;
    breakdown = (alpha == 0 || std::isnan(alpha));
    overtimed = time_limit_reached(ctx, now_s() - t0, timemax);
    if (kdisplay(iter, verbose)) klogf(o.log_fd, "%5lld  %7.1e  %8.1e  %8.1e  %.2fs\n", (long long)iter, rNorm, std::fabs(alpha), std::fabs(omega), now_s() - t0);   // :255
  }

  if (verbose > 0) { klogf(o.log_fd, "\n"); klog_flush(o.log_fd); }                                                           // src/bicgstab.jl:257
  if (tired) status = "maximum number of iterations exceeded";
  if (breakdown) status = "breakdown \xce\xb1\xe2\x82\x96 == 0";
  if (solved) status = "solution good enough given atol and rtol";
  if (user_requested_exit) status = "user-requested exit";
  if (overtimed) status = "time limit exceeded";

  if (warm_start) K(khip_axpy(ctx, n, 1.0, dx, x));
  ws->warm_start = false;
  K(khip_ctx_sync(ctx));

  st->niter = (int)iter;
  st->solved = solved;
  st->inconsistent = 0;
  st->timer = now_s() - t0;
  st->allocation_timer += take_alloc_seconds();                 // lazy allocations of this solve (allocate_if)
  snprintf(st->status, sizeof(st->status), "%s", status);
  ws->box.publish();
  return KHIP_OK;
}

}  // extern "C"
