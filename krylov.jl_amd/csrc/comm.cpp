// comm.cpp -- 1-D row-partitioned multi-GPU support: RCCL over xGMI, one process per GPU.
//
// Decomposition (the reference's MPIVector recipe,
// docs/src/custom_workspaces.md:477-586): rank g owns a contiguous block of rows of A and the
// same slice of every work vector.  BLAS-1 kernels stay local.  kdot/knorm: local compensated
// (hi, lo) partial -> ncclAllGather of 16 bytes per rank -> the G partials are summed on the host in
// rank order with TwoSum, so every rank computes the bit-identical scalar and the result matches
// the single-GPU value to 1 ulp.  SpMV: only the remote x entries a rank's columns actually
// reference are exchanged (for a slab-partitioned 7-point grid: the two neighbouring planes)
// with grouped ncclSend/ncclRecv on a second stream, overlapped with the interior rows.
//
// RCCL is dlopen'ed on first use so that single-GPU users never map the 570 MB library and the
// .so loads on machines without it.  (Inside a PyTorch process the already-loaded librccl.so.1 is
// reused: same SONAME.)
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <condition_variable>
#include <cstdarg>
#include <map>
#include <mutex>

#include "khip_internal.hpp"

namespace khip {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char *get_error() { return g_err; }
static thread_local int g_hip_err = 0;
void note_hip_error(int e) { g_hip_err = e; }
int take_hip_error() { const int e = g_hip_err; g_hip_err = 0; return e; }

struct Rccl {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  // optional (absent in very old builds): rank count as the library sees it, and a second communicator for the halo
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommSplit)(ncclComm_t, int, int, ncclComm_t *, ncclConfig_t *) = nullptr;
};

static Rccl g_rccl;

static int load_rccl() {
  if (g_rccl.handle) return KHIP_OK;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void *h = nullptr;
  for (const char *nm : names) {
    h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) {
    set_error("cannot dlopen librccl: %s", dlerror());
    return KHIP_ERR_COMM;
  }
#define KHIP_SYM(field, name)                                         \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name)); \
  if (!g_rccl.field) { set_error("librccl lacks %s", name); return KHIP_ERR_COMM; }
  KHIP_SYM(GetUniqueId, "ncclGetUniqueId")
  KHIP_SYM(CommInitRank, "ncclCommInitRank")
  KHIP_SYM(CommDestroy, "ncclCommDestroy")
  KHIP_SYM(AllGather, "ncclAllGather")
  KHIP_SYM(Send, "ncclSend")
  KHIP_SYM(Recv, "ncclRecv")
  KHIP_SYM(GroupStart, "ncclGroupStart")
  KHIP_SYM(GroupEnd, "ncclGroupEnd")
  KHIP_SYM(GetErrorString, "ncclGetErrorString")
#undef KHIP_SYM
  g_rccl.CommCount = reinterpret_cast<decltype(g_rccl.CommCount)>(dlsym(h, "ncclCommCount"));
  g_rccl.CommSplit = reinterpret_cast<decltype(g_rccl.CommSplit)>(dlsym(h, "ncclCommSplit"));
  g_rccl.handle = h;
  return KHIP_OK;
}

#define KHIP_CHECK_NCCL(expr)                                                                  \
  do {                                                                                         \
    ncclResult_t r__ = (expr);                                                                 \
    if (r__ != ncclSuccess) {                                                                  \
      set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r__), __FILE__, __LINE__); \
      return KHIP_ERR_COMM;                                                                    \
    }                                                                                          \
  } while (0)

// In-process communicator ("local" backend): all ranks are contexts of ONE process, each driven by its
// own host thread, and exchange through device-to-device copies plus a host barrier.  It exists so that
// the whole distributed path (plan construction on the device, [owned | ghost] kernels, interior /
// boundary split, all-reduced dots) can be tested on a single GPU, where RCCL refuses two ranks on one
// device; it is also usable for one-process multi-GPU runs.  Only the three transport calls differ from
// the RCCL backend.
struct LocalHub {
  int nranks = 0;
  int joined = 0;
  std::mutex mu;
  std::condition_variable cv;
  int waiting = 0;
  long generation = 0;
  std::vector<std::vector<char>> stage;          // per-rank host staging for the small all-gathers
  std::vector<const double *> sendbuf;           // per-rank device send buffers of the current exchange
  std::vector<std::vector<int64_t>> send_off;
  void barrier() {
    std::unique_lock<std::mutex> lk(mu);
    const long gen = generation;
    if (++waiting == nranks) {
      waiting = 0;
      ++generation;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return generation != gen; });
    }
  }
};
static std::mutex g_hub_mu;
static std::map<int, LocalHub *> g_hubs;

// Stream / communicator discipline.  Every rank issues its RCCL calls from ONE host thread in
// program order, so the order of operations per communicator is the same on all ranks.  Two communicators:
//   comm       all-gathers of the (hi, lo) dot partials and the setup-time all-gathers, always on ctx->stream;
//   halo_comm  the grouped ncclSend / ncclRecv of the halo exchange (and the all-gather of x in gather mode), on
//              ctx->comm_stream when the exchange overlaps the interior rows.
// halo_comm is split off comm (ncclCommSplit, same ranks).  If the library cannot split, halo_comm == comm and the
// exchange is issued on ctx->stream as well (no overlap): one communicator is then only ever used from one stream.
struct Comm {
  int rank = 0, nranks = 1;
  ncclComm_t comm = nullptr;
  ncclComm_t halo_comm = nullptr;  // == comm when the split is unavailable
  int rccl_ranks = 0;              // ncclCommCount(comm): what the library itself believes (0 = local backend)
  LocalHub *hub = nullptr;         // non-null => local backend
  dd *gather_dev = nullptr;       // [nranks][kMaxRedOut]
  dd *gather_pinned = nullptr;
  void *scratch_dev = nullptr;    // setup-time allgather staging
  size_t scratch_bytes = 0;
};

int comm_nranks(const khip_ctx *ctx) { return ctx->comm ? ctx->comm->nranks : 1; }
int comm_rank_of(const khip_ctx *ctx) { return ctx->comm ? ctx->comm->rank : 0; }

// --------------------------------------------------------------------------- host plan
// Pure host logic, exported for the CPU (gloo) tests: given every rank's sorted list of needed
// remote columns, derive what this rank receives from / sends to each peer.
int build_halo_plan_host(int rank, int nranks, const int64_t *row_starts, const int32_t *ghost_all,
                         const int64_t *ghost_off, std::vector<int64_t> &recv_off, std::vector<int64_t> &send_off,
                         std::vector<int32_t> &send_idx) {
  recv_off.assign(nranks + 1, 0);
  send_off.assign(nranks + 1, 0);
  send_idx.clear();
  // my ghost list is sorted -> contiguous segments per owner
  const int32_t *mine = ghost_all + ghost_off[rank];
  const int64_t nmine = ghost_off[rank + 1] - ghost_off[rank];
  int64_t pos = 0;
  for (int r = 0; r < nranks; ++r) {
    recv_off[r] = pos;
    while (pos < nmine && (int64_t)mine[pos] < row_starts[r + 1]) {
      if ((int64_t)mine[pos] < row_starts[r]) return KHIP_ERR_INVALID;   // unsorted input
      ++pos;
    }
    if (r == rank && pos != recv_off[r]) return KHIP_ERR_INVALID;       // own columns are not ghosts
  }
  recv_off[nranks] = pos;
  if (pos != nmine) return KHIP_ERR_INVALID;
  // what peers need from me, in THEIR list order (that is the order they expect to receive)
  const int64_t lo = row_starts[rank], hi = row_starts[rank + 1];
  for (int r = 0; r < nranks; ++r) {
    send_off[r] = (int64_t)send_idx.size();
    if (r == rank) continue;
    const int32_t *theirs = ghost_all + ghost_off[r];
    const int64_t nt = ghost_off[r + 1] - ghost_off[r];
    const int32_t *b = std::lower_bound(theirs, theirs + nt, (int32_t)lo);
    for (const int32_t *p = b; p < theirs + nt && (int64_t)*p < hi; ++p) send_idx.push_back((int32_t)(*p - lo));
  }
  send_off[nranks] = (int64_t)send_idx.size();
  return KHIP_OK;
}

// --------------------------------------------------------------------------- setup-time allgather
static int allgather_host(khip_ctx *ctx, const void *in, void *out, size_t bytes_per_rank) {
  Comm *c = ctx->comm;
  if (c->hub) {
    LocalHub *h = c->hub;
    h->stage[c->rank].assign(static_cast<const char *>(in), static_cast<const char *>(in) + bytes_per_rank);
    h->barrier();
    for (int r = 0; r < c->nranks; ++r) memcpy(static_cast<char *>(out) + (size_t)r * bytes_per_rank, h->stage[r].data(), bytes_per_rank);
    h->barrier();
    return KHIP_OK;
  }
  const size_t need = bytes_per_rank * (size_t)(c->nranks + 1);
  if (need > c->scratch_bytes) {
    if (c->scratch_dev) KHIP_CHECK_HIP(hipFree(c->scratch_dev));
    KHIP_CHECK_HIP(hipMalloc(&c->scratch_dev, need));
    c->scratch_bytes = need;
  }
  char *send = static_cast<char *>(c->scratch_dev);
  char *recv = send + bytes_per_rank;
  KHIP_CHECK_HIP(hipMemcpyAsync(send, in, bytes_per_rank, hipMemcpyHostToDevice, ctx->stream));
  KHIP_CHECK_NCCL(g_rccl.AllGather(send, recv, bytes_per_rank, ncclUint8, c->comm, ctx->stream));
  KHIP_CHECK_HIP(hipMemcpyAsync(out, recv, bytes_per_rank * c->nranks, hipMemcpyDeviceToHost, ctx->stream));
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return KHIP_OK;
}

// Everything that can fail on ONE rank only (staging overflow, an inconsistent partition, the host plan) is carried as
// a status word through the next setup all-gather, so that all ranks leave together with the same error instead of one
// rank returning while its peers block in the collective.  (A failing HIP call is not recoverable either way.)
struct DevScratch {                        // frees setup-time device buffers on every path out
  std::vector<void *> ptrs;
  ~DevScratch() { for (void *p : ptrs) (void)hipFree(p); }
  template <typename T> int alloc(T **out, size_t count) {
    KHIP_CHECK_HIP(hipMalloc(reinterpret_cast<void **>(out), sizeof(T) * (count ? count : 1)));
    ptrs.push_back(*out);
    return KHIP_OK;
  }
};

enum PlanStatus : int64_t { PLAN_OK = 0, PLAN_WANT_GATHER = 1, PLAN_FAILED = 2 };

int comm_build_plan(khip_ctx *ctx, khip_csr *A, int local_rc) {
  Comm *c = ctx->comm;
  if (!c) { set_error("distributed operator needs khip_comm_init first"); return KHIP_ERR_INVALID; }
  const int G = c->nranks;
  DevScratch scratch;
  // 1. off-rank columns of my rows (device filter -> host sort/unique).  The staging buffer counts REFERENCES; a shard
  //    whose references do not fit is not an error: it asks for gather mode, which needs no list at all.
  //    Nothing returns before the first collective: a rank whose handle could not be created (local_rc != 0, A == null) or
  //    whose filter fails reports PLAN_FAILED in its status word, so that ALL ranks leave with an error instead of the
  //    healthy ones blocking in the all-gather for good.
  std::vector<int32_t> ghost;
  int64_t status = (local_rc == KHIP_OK && A) ? PLAN_OK : PLAN_FAILED;
  std::string local_error = status == PLAN_FAILED ? std::string(khip_last_error()) : std::string();
  auto collect = [&]() -> int {
    int64_t cap = A->nnz < (64ll << 20) ? A->nnz : (64ll << 20);
    if (cap < 1) cap = 1;
    int32_t *d_list = nullptr;
    unsigned long long *d_cnt = nullptr;
    KHIP_TRY(scratch.alloc(&d_list, (size_t)cap));
    KHIP_TRY(scratch.alloc(&d_cnt, 1));
    KHIP_CHECK_HIP(hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), ctx->stream));
    KHIP_TRY(launch_collect_offrank(ctx, A, A->row0, A->row0 + A->m, d_list, d_cnt, cap));
    unsigned long long cnt = 0;
    KHIP_CHECK_HIP(hipMemcpyAsync(&cnt, d_cnt, sizeof(cnt), hipMemcpyDeviceToHost, ctx->stream));
    KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    if ((int64_t)cnt > cap) {
      status = PLAN_WANT_GATHER;
    } else {
      ghost.resize((size_t)cnt);
      if (cnt) KHIP_CHECK_HIP(hipMemcpy(ghost.data(), d_list, sizeof(int32_t) * (size_t)cnt, hipMemcpyDeviceToHost));
      std::sort(ghost.begin(), ghost.end());
      ghost.erase(std::unique(ghost.begin(), ghost.end()), ghost.end());
    }
    return KHIP_OK;
  };
  if (status != PLAN_FAILED && collect() != KHIP_OK) {
    status = PLAN_FAILED;
    local_error = khip_last_error();
    ghost.clear();
  }
  // 2. partition, list sizes and status of every rank
  int64_t mine[4] = {A ? A->row0 : 0, A ? A->m : 0, (int64_t)ghost.size(), status};
  std::vector<int64_t> all(4 * (size_t)G);
  KHIP_TRY(allgather_host(ctx, mine, all.data(), sizeof(mine)));
  for (int r = 0; r < G; ++r)
    if (all[4 * r + 3] == PLAN_FAILED) {
      if (status == PLAN_FAILED) set_error("csr_create_dist: %s", local_error.c_str());
      else set_error("csr_create_dist: rank %d could not create its shard of the operator", r);
      return local_rc != KHIP_OK ? local_rc : KHIP_ERR_INVALID;
    }
  std::vector<int64_t> row_starts(G + 1), ghost_off(G + 1, 0);
  int64_t maxcnt = 1, maxm = 1;
  bool bad_partition = false, want_gather = ctx->tune.halo_mode == 2;
  for (int r = 0; r < G; ++r) {
    row_starts[r] = all[4 * r];
    if (r > 0 && all[4 * (r - 1)] + all[4 * (r - 1) + 1] != all[4 * r]) bad_partition = true;
    ghost_off[r + 1] = ghost_off[r] + all[4 * r + 2];
    maxcnt = std::max(maxcnt, all[4 * r + 2]);
    maxm = std::max(maxm, all[4 * r + 1]);
    if (all[4 * r + 3] == PLAN_WANT_GATHER) want_gather = true;
    if (ctx->tune.halo_mode == 0 && all[4 * r + 2] * 100 > all[4 * r + 1] * (int64_t)ctx->tune.halo_gather_pct && all[4 * r + 2] > 0)
      want_gather = true;                      // this rank would fetch more than halo_gather_pct % of its own size
  }
  row_starts[G] = all[4 * (G - 1)] + all[4 * (G - 1) + 1];
  A->part_starts = row_starts;
  if (G == 1 && ctx->tune.halo_self && status == PLAN_OK && (A->row0 != 0 || A->m != A->n_global)) {
    // ---- self halo (measurement hook, Tuning::halo_self): the slab's off-slab columns wrap onto its own rows and travel
    //      through the real neighbour exchange, from this rank to itself
    if (c->hub) { set_error("halo_self needs the RCCL backend"); return KHIP_ERR_UNSUPPORTED; }
    std::vector<int32_t> send_idx(ghost.size());
    for (size_t i = 0; i < ghost.size(); ++i) {
      int64_t loc = ((int64_t)ghost[i] - A->row0) % A->m;
      if (loc < 0) loc += A->m;
      send_idx[i] = (int32_t)loc;
    }
    A->recv_off = {0, (int64_t)ghost.size()};
    A->send_off = {0, (int64_t)ghost.size()};
    A->n_ghost = (int64_t)ghost.size();
    A->n_send = (int64_t)send_idx.size();
    A->ghost_gid = ghost;
    A->self_halo = true;
    int32_t *d_ghost_sorted = nullptr;
    KHIP_TRY(scratch.alloc(&d_ghost_sorted, (size_t)std::max<int64_t>(A->n_ghost, 1)));
    if (A->n_ghost)
      KHIP_CHECK_HIP(hipMemcpyAsync(d_ghost_sorted, ghost.data(), sizeof(int32_t) * (size_t)A->n_ghost, hipMemcpyHostToDevice, ctx->stream));
    KHIP_TRY(launch_col_remap(ctx, A, d_ghost_sorted, A->n_ghost));
    KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    KHIP_CHECK_HIP(hipMalloc(&A->ghost, sizeof(double) * (size_t)std::max<int64_t>(A->n_ghost, 1)));
    KHIP_CHECK_HIP(hipMalloc(&A->sendbuf, sizeof(double) * (size_t)std::max<int64_t>(A->n_send, 1)));
    KHIP_CHECK_HIP(hipMalloc(&A->send_idx, sizeof(int32_t) * (size_t)std::max<int64_t>(A->n_send, 1)));
    if (A->n_send)
      KHIP_CHECK_HIP(hipMemcpy(A->send_idx, send_idx.data(), sizeof(int32_t) * (size_t)A->n_send, hipMemcpyHostToDevice));
    KHIP_CHECK_HIP(hipMemsetAsync(A->ghost, 0, sizeof(double) * (size_t)std::max<int64_t>(A->n_ghost, 1), ctx->stream));
    int64_t lo_hi_self[2];
    KHIP_TRY(launch_row_ghost_range(ctx, A, lo_hi_self));
    A->interior_lo = lo_hi_self[0];
    A->interior_hi = lo_hi_self[1];
    return KHIP_OK;
  }
  if (bad_partition || row_starts[0] != 0 || row_starts[G] != A->n_global) {     // same data on every rank: all fail together
    set_error("row partition [%lld, %lld) is not contiguous or does not cover n_global = %lld", (long long)row_starts[0],
              (long long)row_starts[G], (long long)A->n_global);
    return KHIP_ERR_INVALID;
  }
  if (ctx->tune.halo_mode == 1 && want_gather) {
    // forced neighbour exchange although a rank's references overflowed the staging buffer: nothing to build a list from
    bool overflow = false;
    for (int r = 0; r < G; ++r) overflow |= all[4 * r + 3] == PLAN_WANT_GATHER;
    if (overflow) { set_error("halo plan: off-rank references exceed the staging buffer; use halo_mode 0 or 2 (all-gather of x)"); return KHIP_ERR_UNSUPPORTED; }
    want_gather = false;
  }
  A->gather = false;
  if (want_gather && G > 1) {
    // ---- gather mode: x is all-gathered before every product (the general fallback of SURVEY 8e / the reference's
    //      recipe docs/src/custom_workspaces.md:583-586); ghost region = receive buffer, stride maxm per rank
    if (A->m + (int64_t)G * maxm >= (1ll << 31)) { set_error("gather mode: %d x %lld columns exceed int32 indexing", G, (long long)maxm); return KHIP_ERR_UNSUPPORTED; }
    int64_t *d_starts = nullptr;
    KHIP_TRY(scratch.alloc(&d_starts, (size_t)G + 1));
    KHIP_CHECK_HIP(hipMemcpyAsync(d_starts, row_starts.data(), sizeof(int64_t) * (size_t)(G + 1), hipMemcpyHostToDevice, ctx->stream));
    KHIP_TRY(launch_col_remap_gather(ctx, A, d_starts, G, maxm));
    KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    A->gather = true;
    A->gather_maxm = maxm;
    A->gather_rows.assign((size_t)G, 0);
    for (int r = 0; r < G; ++r) A->gather_rows[(size_t)r] = all[4 * r + 1];
    A->n_ghost = (int64_t)G * maxm;
    A->n_send = maxm;                                  // staging for a short last slice (m < maxm)
    A->recv_off.assign(G + 1, 0); A->send_off.assign(G + 1, 0);
    KHIP_CHECK_HIP(hipMalloc(&A->ghost, sizeof(double) * (size_t)A->n_ghost));
    KHIP_CHECK_HIP(hipMalloc(&A->sendbuf, sizeof(double) * (size_t)maxm));
    KHIP_CHECK_HIP(hipMemsetAsync(A->ghost, 0, sizeof(double) * (size_t)A->n_ghost, ctx->stream));
    KHIP_CHECK_HIP(hipMemsetAsync(A->sendbuf, 0, sizeof(double) * (size_t)maxm, ctx->stream));
    int64_t lo_hi[2];
    KHIP_TRY(launch_row_ghost_range(ctx, A, lo_hi));
    A->interior_lo = lo_hi[0];
    A->interior_hi = lo_hi[1];
    return KHIP_OK;
  }
  // 3. every rank's ghost list (padded allgather)
  std::vector<int32_t> padded((size_t)maxcnt, 0), gathered((size_t)maxcnt * G);
  std::copy(ghost.begin(), ghost.end(), padded.begin());
  KHIP_TRY(allgather_host(ctx, padded.data(), gathered.data(), sizeof(int32_t) * (size_t)maxcnt));
  std::vector<int32_t> ghost_all((size_t)ghost_off[G]);
  for (int r = 0; r < G; ++r)
    std::copy(gathered.begin() + (size_t)r * maxcnt, gathered.begin() + (size_t)r * maxcnt + (ghost_off[r + 1] - ghost_off[r]),
              ghost_all.begin() + ghost_off[r]);
  // 4. plan (a function of the gathered lists; a failure here is nevertheless agreed on collectively)
  std::vector<int32_t> send_idx;
  int rc = build_halo_plan_host(c->rank, G, row_starts.data(), ghost_all.data(), ghost_off.data(), A->recv_off,
                                A->send_off, send_idx);
  double failed = rc != KHIP_OK ? 1.0 : 0.0;
  KHIP_TRY(comm_allreduce_sum_host(ctx, &failed, 1));
  if (failed > 0) { set_error("halo plan construction failed on %d rank(s)", (int)failed); return rc != KHIP_OK ? rc : KHIP_ERR_INVALID; }
  A->n_ghost = (int64_t)ghost.size();
  A->n_send = (int64_t)send_idx.size();
  A->ghost_gid = ghost;
  // 5. device state
  int32_t *d_ghost_sorted = nullptr;
  KHIP_TRY(scratch.alloc(&d_ghost_sorted, (size_t)std::max<int64_t>(A->n_ghost, 1)));
  if (A->n_ghost)
    KHIP_CHECK_HIP(hipMemcpyAsync(d_ghost_sorted, ghost.data(), sizeof(int32_t) * (size_t)A->n_ghost,
                                  hipMemcpyHostToDevice, ctx->stream));
  KHIP_TRY(launch_col_remap(ctx, A, d_ghost_sorted, A->n_ghost));
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  KHIP_CHECK_HIP(hipMalloc(&A->ghost, sizeof(double) * (size_t)std::max<int64_t>(A->n_ghost, 1)));
  KHIP_CHECK_HIP(hipMalloc(&A->sendbuf, sizeof(double) * (size_t)std::max<int64_t>(A->n_send, 1)));
  KHIP_CHECK_HIP(hipMalloc(&A->send_idx, sizeof(int32_t) * (size_t)std::max<int64_t>(A->n_send, 1)));
  if (A->n_send)
    KHIP_CHECK_HIP(hipMemcpy(A->send_idx, send_idx.data(), sizeof(int32_t) * (size_t)A->n_send, hipMemcpyHostToDevice));
  KHIP_CHECK_HIP(hipMemsetAsync(A->ghost, 0, sizeof(double) * (size_t)std::max<int64_t>(A->n_ghost, 1), ctx->stream));
  int64_t lo_hi[2];
  KHIP_TRY(launch_row_ghost_range(ctx, A, lo_hi));
  A->interior_lo = lo_hi[0];
  A->interior_hi = lo_hi[1];
  return KHIP_OK;
}

// --------------------------------------------------------------------------- setup-time all-to-all, distributed transpose
// Variable-size exchange of device buffers between all ranks (setup only): send_off / recv_off are BYTE offsets per peer
// (size nranks + 1).  RCCL: grouped Send / Recv on the main communicator; local backend: copies out of the peers' buffers.
static int alltoallv_dev(khip_ctx *ctx, const char *send_dev, const std::vector<int64_t> &send_off, char *recv_dev,
                         const std::vector<int64_t> &recv_off) {
  Comm *c = ctx->comm;
  const int G = c->nranks, me = c->rank;
  const int64_t self = send_off[me + 1] - send_off[me];
  if (self > 0) KHIP_CHECK_HIP(hipMemcpyAsync(recv_dev + recv_off[me], send_dev + send_off[me], (size_t)self, hipMemcpyDeviceToDevice, ctx->stream));
  if (c->hub) {
    LocalHub *h = c->hub;
    KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    h->sendbuf[me] = reinterpret_cast<const double *>(send_dev);
    h->send_off[me] = send_off;
    h->barrier();
    for (int r = 0; r < G; ++r) {
      if (r == me) continue;
      const int64_t nb = recv_off[r + 1] - recv_off[r];
      if (nb > 0)
        KHIP_CHECK_HIP(hipMemcpyAsync(recv_dev + recv_off[r], reinterpret_cast<const char *>(h->sendbuf[r]) + h->send_off[r][me], (size_t)nb,
                                      hipMemcpyDeviceToDevice, ctx->stream));
    }
    KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    h->barrier();
    return KHIP_OK;
  }
  KHIP_CHECK_NCCL(g_rccl.GroupStart());
  for (int r = 0; r < G; ++r) {
    if (r == me) continue;
    const int64_t ns = send_off[r + 1] - send_off[r], nr = recv_off[r + 1] - recv_off[r];
    if (ns > 0) KHIP_CHECK_NCCL(g_rccl.Send(send_dev + send_off[r], (size_t)ns, ncclUint8, r, c->comm, ctx->stream));
    if (nr > 0) KHIP_CHECK_NCCL(g_rccl.Recv(recv_dev + recv_off[r], (size_t)nr, ncclUint8, r, c->comm, ctx->stream));
  }
  KHIP_CHECK_NCCL(g_rccl.GroupEnd());
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return KHIP_OK;
}

// A' of a row-partitioned operator, partitioned the same way (the two-sided processes and solvers of the reference need
// mul!(y, A', x): src/krylov_processes.jl:133-222, docs/src/matrix_free.md:36-42).  Every entry (i, j, v) of my rows goes to the
// owner of row j of A'; the receiver lines its entries up by local row, peers in rank order and each peer's entries in its
// own (row-major) order, so that a row of A' holds a column of A in ascending row order -- the entry order of the
// single-GPU khip_csr_transpose and of the CSC matrix the reference's users hold; y = A' x is bit-identical to that.
// Set-up code: the bucketing runs on the host (one D2H of the shard), the exchange on the device buffers RCCL needs.
int comm_transpose_dist(khip_ctx *ctx, const khip_csr *A, khip_csr **out) {
  Comm *c = ctx->comm;
  if (!c || !A->dist || (int)A->part_starts.size() != c->nranks + 1) { set_error("csr_transpose: not a distributed handle of this communicator"); return KHIP_ERR_INVALID; }
  const int G = c->nranks, me = c->rank;
  const int64_t m = A->m, nnz = A->nnz, row0 = A->row0;
  const std::vector<int64_t> &starts = A->part_starts;
  std::vector<int32_t> rp((size_t)m + 1), col((size_t)nnz);
  std::vector<double> val((size_t)nnz);
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  KHIP_CHECK_HIP(hipMemcpy(rp.data(), A->rowptr, sizeof(int32_t) * (size_t)(m + 1), hipMemcpyDeviceToHost));
  if (nnz) {
    KHIP_CHECK_HIP(hipMemcpy(col.data(), A->col, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToHost));
    KHIP_CHECK_HIP(hipMemcpy(val.data(), A->val, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToHost));
  }
  // local column -> global column: [owned | ghost]; ghost slots are the sorted ghost list (neighbour mode) or the peers'
  // slices at stride gather_maxm (gather mode)
  auto global_col = [&](int32_t lc) -> int64_t {
    if (lc < m) return row0 + lc;
    const int64_t g = (int64_t)lc - m;
    if (A->gather) { const int64_t r = g / A->gather_maxm; return starts[(size_t)r] + (g - r * A->gather_maxm); }
    return A->ghost_gid[(size_t)g];
  };
  auto owner = [&](int64_t j) -> int { return (int)(std::upper_bound(starts.begin(), starts.end(), j) - starts.begin()) - 1; };
  struct Trip { int32_t row, col; double v; };             // (global row of A, global column of A, value): 16 bytes
  std::vector<int64_t> cnt((size_t)G, 0);
  for (int64_t q = 0; q < nnz; ++q) cnt[(size_t)owner(global_col(col[(size_t)q]))]++;
  std::vector<int64_t> send_off((size_t)G + 1, 0);
  for (int r = 0; r < G; ++r) send_off[(size_t)r + 1] = send_off[(size_t)r] + cnt[(size_t)r];
  std::vector<Trip> sendbuf((size_t)nnz);
  {
    std::vector<int64_t> cur(send_off.begin(), send_off.end() - 1);
    for (int64_t i = 0; i < m; ++i)
      for (int32_t q = rp[(size_t)i]; q < rp[(size_t)i + 1]; ++q) {
        const int64_t j = global_col(col[(size_t)q]);
        sendbuf[(size_t)cur[(size_t)owner(j)]++] = Trip{(int32_t)(row0 + i), (int32_t)j, val[(size_t)q]};
      }
  }
  // everybody's counts: recv_cnt[r] = what rank r sends to me
  std::vector<int64_t> all((size_t)G * G);
  KHIP_TRY(allgather_host(ctx, cnt.data(), all.data(), sizeof(int64_t) * (size_t)G));
  std::vector<int64_t> recv_off((size_t)G + 1, 0);
  for (int r = 0; r < G; ++r) recv_off[(size_t)r + 1] = recv_off[(size_t)r] + all[(size_t)r * G + me];
  const int64_t nnzT = recv_off[(size_t)G];
  // every rank sees every rank's counts: a shard too large for int32 row pointers fails the call on ALL ranks, before the
  // exchange (a rank leaving alone would leave the others in it for good)
  for (int r = 0; r < G; ++r) {
    int64_t tot = 0;
    for (int q = 0; q < G; ++q) tot += all[(size_t)q * G + r];
    if (tot >= (1ll << 31) - 64) { set_error("csr_transpose: the transposed shard of rank %d has %lld entries (int32 row pointers)", r, (long long)tot); return KHIP_ERR_UNSUPPORTED; }
  }
  DevScratch scratch;
  char *d_send = nullptr, *d_recv = nullptr;
  KHIP_TRY(scratch.alloc(&d_send, sizeof(Trip) * (size_t)std::max<int64_t>(nnz, 1)));
  KHIP_TRY(scratch.alloc(&d_recv, sizeof(Trip) * (size_t)std::max<int64_t>(nnzT, 1)));
  if (nnz) KHIP_CHECK_HIP(hipMemcpy(d_send, sendbuf.data(), sizeof(Trip) * (size_t)nnz, hipMemcpyHostToDevice));
  std::vector<int64_t> so((size_t)G + 1), ro((size_t)G + 1);
  for (int r = 0; r <= G; ++r) { so[(size_t)r] = send_off[(size_t)r] * (int64_t)sizeof(Trip); ro[(size_t)r] = recv_off[(size_t)r] * (int64_t)sizeof(Trip); }
  KHIP_TRY(alltoallv_dev(ctx, d_send, so, d_recv, ro));
  std::vector<Trip> got((size_t)nnzT);
  if (nnzT) KHIP_CHECK_HIP(hipMemcpy(got.data(), d_recv, sizeof(Trip) * (size_t)nnzT, hipMemcpyDeviceToHost));
  // stable counting sort by local row of A' (= global column of A - row0); `got` is already ordered (peer, row of A)
  std::vector<int64_t> rpT((size_t)m + 1, 0);
  for (const Trip &t : got) {
    const int64_t lr = (int64_t)t.col - row0;
    if (lr < 0 || lr >= m) {                                 // (internal) -- fail on all ranks through the next collective
      set_error("csr_transpose: received an entry of column %d outside [%lld, %lld)", t.col, (long long)row0, (long long)(row0 + m));
      return comm_build_plan(ctx, nullptr, KHIP_ERR_INVALID);
    }
    rpT[(size_t)lr + 1]++;
  }
  for (int64_t i = 0; i < m; ++i) rpT[(size_t)i + 1] += rpT[(size_t)i];
  std::vector<int32_t> colT((size_t)std::max<int64_t>(nnzT, 1));
  std::vector<double> valT((size_t)std::max<int64_t>(nnzT, 1));
  {
    std::vector<int64_t> cur(rpT.begin(), rpT.end() - 1);
    for (const Trip &t : got) {
      const int64_t k = cur[(size_t)((int64_t)t.col - row0)]++;
      colT[(size_t)k] = t.row;
      valT[(size_t)k] = t.v;
    }
  }
  return khip_csr_create_dist(ctx, A->n_global, row0, m, nnzT, rpT.data(), 64, colT.data(), valT.data(), 0, 0, out);
}

// --------------------------------------------------------------------------- per-SpMV / per-SpMM exchange
// width = 1: vectors (A->sendbuf / A->ghost).  width = p > 1: row-major panels, every exchanged entry is a
// panel row of p doubles (A->sendbuf_w / A->ghost_w, grown on demand).
static int ensure_panel_halo(khip_csr *A, int width) {
  if (width <= A->halo_w_cap) return KHIP_OK;
  if (A->sendbuf_w) KHIP_CHECK_HIP(hipFree(A->sendbuf_w));
  if (A->ghost_w) KHIP_CHECK_HIP(hipFree(A->ghost_w));
  A->sendbuf_w = nullptr; A->ghost_w = nullptr; A->halo_w_cap = 0;
  KHIP_CHECK_HIP(hipMalloc(&A->sendbuf_w, sizeof(double) * (size_t)std::max<int64_t>(A->n_send, 1) * width));
  KHIP_CHECK_HIP(hipMalloc(&A->ghost_w, sizeof(double) * (size_t)std::max<int64_t>(A->n_ghost, 1) * width));
  A->halo_w_cap = width;
  return KHIP_OK;
}

int comm_halo_exchange_begin(khip_ctx *ctx, const khip_csr *A_in, const double *x, int width) {
  Comm *c = ctx->comm;
  khip_csr *A = const_cast<khip_csr *>(A_in);
  if (!c || (c->nranks == 1 && !A->self_halo) || (A->n_send == 0 && A->n_ghost == 0)) return KHIP_OK;
  double *sendbuf = A->sendbuf, *ghost = A->ghost;
  if (width > 1) {
    KHIP_TRY(ensure_panel_halo(A, width));
    sendbuf = A->sendbuf_w; ghost = A->ghost_w;
  }
  const size_t w = (size_t)width;
  if (A->gather) {
    // all-gather of x (row-major panel: of X): every rank contributes its slice, padded to the stride maxm
    const int64_t maxm = A->gather_maxm;
    const double *src = x;
    if (A->m < maxm) {                                            // short slice: stage it (the pad is never referenced)
      KHIP_CHECK_HIP(hipMemcpyAsync(sendbuf, x, sizeof(double) * (size_t)A->m * w, hipMemcpyDeviceToDevice, ctx->stream));
      src = sendbuf;
    }
    if (c->hub) {
      LocalHub *h = c->hub;
      KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
      h->sendbuf[c->rank] = src;
      h->barrier();
      for (int r = 0; r < c->nranks; ++r) {
        if (r == c->rank) continue;
        const int64_t nr = A->gather_rows[(size_t)r];
        if (nr > 0)
          KHIP_CHECK_HIP(hipMemcpyAsync(ghost + (size_t)r * maxm * w, h->sendbuf[r], sizeof(double) * (size_t)nr * w,
                                        hipMemcpyDeviceToDevice, ctx->stream));
      }
      return KHIP_OK;
    }
    hipStream_t gs = (ctx->tune.overlap_halo && c->halo_comm != c->comm) ? ctx->comm_stream : ctx->stream;
    if (gs != ctx->stream) {
      ctx->ev_cur = (ctx->ev_cur + 1) % khip_ctx::kEvRing;
      KHIP_CHECK_HIP(hipEventRecord(ctx->ev_a[ctx->ev_cur], ctx->stream));
      KHIP_CHECK_HIP(hipStreamWaitEvent(gs, ctx->ev_a[ctx->ev_cur], 0));
    }
    {
      ProfScope prof_scope(ctx, kProfHaloXfer, gs);
      KHIP_CHECK_NCCL(g_rccl.AllGather(src, ghost, (size_t)maxm * w, ncclFloat64, c->halo_comm, gs));
    }
    if (gs != ctx->stream) KHIP_CHECK_HIP(hipEventRecord(ctx->ev_b[ctx->ev_cur], gs));
    return KHIP_OK;
  }
  {
    ProfScope prof_scope(ctx, kProfHaloPack);
    KHIP_TRY(launch_gather(ctx, A->n_send, A->send_idx, x, sendbuf, width));
  }
  if (c->hub) {
    LocalHub *h = c->hub;
    KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));          // my send buffer is packed
    h->sendbuf[c->rank] = sendbuf;
    h->send_off[c->rank] = A->send_off;
    h->barrier();                                                // everybody's buffers are packed and published
    for (int r = 0; r < c->nranks; ++r) {
      if (r == c->rank) continue;
      const int64_t nr = A->recv_off[r + 1] - A->recv_off[r];
      if (nr > 0)
        KHIP_CHECK_HIP(hipMemcpyAsync(ghost + A->recv_off[r] * w, h->sendbuf[r] + h->send_off[r][c->rank] * w,
                                      sizeof(double) * (size_t)nr * w, hipMemcpyDeviceToDevice, ctx->stream));
    }
    return KHIP_OK;
  }
  hipStream_t cs = (ctx->tune.overlap_halo && c->halo_comm != c->comm) ? ctx->comm_stream : ctx->stream;
  if (cs != ctx->stream) {
    ctx->ev_cur = (ctx->ev_cur + 1) % khip_ctx::kEvRing;           // begin/end pairs alternate strictly
    KHIP_CHECK_HIP(hipEventRecord(ctx->ev_a[ctx->ev_cur], ctx->stream));
    KHIP_CHECK_HIP(hipStreamWaitEvent(cs, ctx->ev_a[ctx->ev_cur], 0));
  }
  ProfScope prof_xfer(ctx, kProfHaloXfer, cs);
  KHIP_CHECK_NCCL(g_rccl.GroupStart());
  for (int r = 0; r < c->nranks; ++r) {
    if (r == c->rank && !A->self_halo) continue;
    const int64_t ns = A->send_off[r + 1] - A->send_off[r];
    const int64_t nr = A->recv_off[r + 1] - A->recv_off[r];
    if (ns > 0) KHIP_CHECK_NCCL(g_rccl.Send(sendbuf + A->send_off[r] * w, (size_t)ns * w, ncclFloat64, r, c->halo_comm, cs));
    if (nr > 0) KHIP_CHECK_NCCL(g_rccl.Recv(ghost + A->recv_off[r] * w, (size_t)nr * w, ncclFloat64, r, c->halo_comm, cs));
  }
  KHIP_CHECK_NCCL(g_rccl.GroupEnd());
  if (prof_xfer.stop) { (void)hipEventRecord(prof_xfer.stop, cs); prof_xfer.stop = nullptr; }      // the bracket closes behind the transfer, not at the end of this scope
  if (cs != ctx->stream) KHIP_CHECK_HIP(hipEventRecord(ctx->ev_b[ctx->ev_cur], cs));
  return KHIP_OK;
}

int comm_halo_exchange_end(khip_ctx *ctx, const khip_csr *A) {
  Comm *c = ctx->comm;
  if (!c || (c->nranks == 1 && !A->self_halo) || (A->n_send == 0 && A->n_ghost == 0)) return KHIP_OK;
  if (c->hub) {
    KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));          // my copies out of the peers' buffers are done
    c->hub->barrier();                                          // ... and so are theirs out of mine
    return KHIP_OK;
  }
  if (ctx->tune.overlap_halo && c->halo_comm != c->comm) KHIP_CHECK_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_b[ctx->ev_cur], 0));
  return KHIP_OK;
}

// --------------------------------------------------------------------------- scalar all-reduce
static inline void two_sum_h(double a, double b, double &s, double &e) {
  s = a + b;
  double z = s - a;
  e = (a - (s - z)) + (b - z);
}

int comm_allreduce_dd(khip_ctx *ctx, dd *vals_dev, int count, double *out_host) {
  Comm *c = ctx->comm;
  if (count > kMaxRedOut) { set_error("allreduce: too many scalars"); return KHIP_ERR_INVALID; }
  const int G = c->nranks;
  if (c->hub) {
    std::vector<dd> mine((size_t)count);
    KHIP_CHECK_HIP(hipMemcpyAsync(mine.data(), vals_dev, sizeof(dd) * (size_t)count, hipMemcpyDeviceToHost, ctx->stream));
    KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    KHIP_TRY(allgather_host(ctx, mine.data(), c->gather_pinned, sizeof(dd) * (size_t)count));
  } else {
    KHIP_CHECK_NCCL(g_rccl.AllGather(vals_dev, c->gather_dev, (size_t)count * 2, ncclFloat64, c->comm, ctx->stream));
    KHIP_CHECK_HIP(hipMemcpyAsync(c->gather_pinned, c->gather_dev, sizeof(dd) * (size_t)count * G, hipMemcpyDeviceToHost,
                                  ctx->stream));
    KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  }
  for (int i = 0; i < count; ++i) {
    double hi = 0.0, lo = 0.0;
    for (int r = 0; r < G; ++r) {   // fixed rank order -> identical on every rank
      const dd v = c->gather_pinned[(size_t)r * count + i];
      double s, e;
      two_sum_h(hi, v.hi, s, e);
      hi = s;
      lo += v.lo + e;
    }
    out_host[i] = hi + lo;
  }
  return KHIP_OK;
}

// In-place sum over ranks of `count` host doubles, added in rank order (identical on every rank): the p x p blocks of
// the panel products (src/block_gmres.jl:244-247 on a row-partitioned panel) and small integers such as row counts.
int comm_allreduce_sum_host(khip_ctx *ctx, double *vals, int count) {
  Comm *c = ctx->comm;
  if (!c || c->nranks == 1 || count <= 0) return KHIP_OK;
  std::vector<double> all((size_t)count * c->nranks);
  KHIP_TRY(allgather_host(ctx, vals, all.data(), sizeof(double) * (size_t)count));
  for (int i = 0; i < count; ++i) {
    double acc = 0.0;
    for (int r = 0; r < c->nranks; ++r) acc += all[(size_t)r * count + i];
    vals[i] = acc;
  }
  return KHIP_OK;
}

// Device-side variant for the device-resident solver loops: nothing waits on the host with RCCL.
// finish kernel -> results_dd[slot] (local partial) -> ncclAllGather (16 bytes per rank and scalar) ->
// combine kernel (rank-ordered TwoSum, then the solver's scalar epilogue).  The local backend has no
// device-side transport, so it goes through the host and then launches the epilogue alone.
int comm_allreduce_dd_device(khip_ctx *ctx, int slot, int count) {
  Comm *c = ctx->comm;
  if (!c) return launch_epilogue_only(ctx, slot);
  if (count > kMaxRedOut) { set_error("allreduce: too many scalars"); return KHIP_ERR_INVALID; }
  if (c->hub) {
    double vals[kMaxRedOut];
    KHIP_TRY(comm_allreduce_dd(ctx, ctx->results_dd + slot, count, vals));
    KHIP_CHECK_HIP(hipMemcpyAsync(ctx->results + slot, vals, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, ctx->stream));
    KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));      // vals lives on this stack frame
    return launch_epilogue_only(ctx, slot);
  }
  ProfScope prof_scope(ctx, kProfDotGather);
  KHIP_CHECK_NCCL(g_rccl.AllGather(ctx->results_dd + slot, c->gather_dev, (size_t)count * 2, ncclFloat64, c->comm, ctx->stream));
  return launch_combine(ctx, c->gather_dev, c->nranks, count, slot);
}

int comm_allreduce_dd_device_begin(khip_ctx *ctx, int slot, int count) {
  Comm *c = ctx->comm;
  ctx->allreduce_pending = false;
  if (!c || c->hub || c->halo_comm == c->comm || !ctx->tune.overlap_halo) return comm_allreduce_dd_device(ctx, slot, count);
  if (count > kMaxRedOut) { set_error("allreduce: too many scalars"); return KHIP_ERR_INVALID; }
  // The (hi, lo) partials are final on the main stream; everything from here to the epilogue runs on the communication
  // stream, on the communicator that lives there (halo_comm: its collectives are issued from one stream, in one order on
  // all ranks -- this all-gather first, then whatever halo exchange the overlapped product needs).
  hipStream_t cs = ctx->comm_stream;
  ctx->ev_cur = (ctx->ev_cur + 1) % khip_ctx::kEvRing;
  KHIP_CHECK_HIP(hipEventRecord(ctx->ev_a[ctx->ev_cur], ctx->stream));
  KHIP_CHECK_HIP(hipStreamWaitEvent(cs, ctx->ev_a[ctx->ev_cur], 0));
  {
    ProfScope prof_scope(ctx, kProfDotGather, cs);
    KHIP_CHECK_NCCL(g_rccl.AllGather(ctx->results_dd + slot, c->gather_dev, (size_t)count * 2, ncclFloat64, c->halo_comm, cs));
    KHIP_TRY(launch_combine(ctx, c->gather_dev, c->nranks, count, slot, cs));
  }
  if (!ctx->ev_red) KHIP_CHECK_HIP(hipEventCreateWithFlags(&ctx->ev_red, hipEventDisableTiming));
  KHIP_CHECK_HIP(hipEventRecord(ctx->ev_red, cs));
  ctx->allreduce_pending = true;      // state of THIS context, next to ev_red
  return KHIP_OK;
}
int comm_allreduce_dd_device_end(khip_ctx *ctx) {
  if (!ctx->allreduce_pending) return KHIP_OK;
  ctx->allreduce_pending = false;
  KHIP_CHECK_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_red, 0));
  return KHIP_OK;
}

}  // namespace khip

using namespace khip;

extern "C" {

const char *khip_last_error(void) { return khip::get_error(); }

int khip_comm_unique_id(void *id128_host) {
  KHIP_REQUIRE(id128_host, "comm_unique_id: null buffer");
  KHIP_TRY(load_rccl());
  ncclUniqueId id;
  KHIP_CHECK_NCCL(g_rccl.GetUniqueId(&id));
  memcpy(id128_host, id.internal, NCCL_UNIQUE_ID_BYTES);
  return KHIP_OK;
}

int khip_comm_init(khip_ctx *ctx, int rank, int nranks, const void *id128_host) {
  KHIP_REQUIRE(ctx && id128_host && nranks >= 1 && rank >= 0 && rank < nranks, "comm_init: bad arguments");
  KHIP_REQUIRE(!ctx->comm, "comm_init: context already has a communicator");
  KHIP_TRY(load_rccl());
  KHIP_CHECK_HIP(hipSetDevice(ctx->device));
  Comm *c = new Comm();
  c->rank = rank;
  c->nranks = nranks;
  ncclUniqueId id;
  memcpy(id.internal, id128_host, NCCL_UNIQUE_ID_BYTES);
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
  if (r != ncclSuccess) {
    set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, g_rccl.GetErrorString(r));
    delete c;
    return KHIP_ERR_COMM;
  }
  c->halo_comm = c->comm;
  if (g_rccl.CommSplit && (nranks > 1 || ctx->tune.halo_self)) {
    ncclComm_t h2 = nullptr;
    if (g_rccl.CommSplit(c->comm, 0, rank, &h2, nullptr) == ncclSuccess && h2) c->halo_comm = h2;
  }
  c->rccl_ranks = nranks;
  if (g_rccl.CommCount) {
    int cnt = 0;
    if (g_rccl.CommCount(c->comm, &cnt) == ncclSuccess) c->rccl_ranks = cnt;
  }
  KHIP_CHECK_HIP(hipMalloc(&c->gather_dev, sizeof(dd) * (size_t)kMaxRedOut * nranks));
  KHIP_CHECK_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->gather_pinned), sizeof(dd) * (size_t)kMaxRedOut * nranks,
                               hipHostMallocDefault));
  ctx->comm = c;
  return KHIP_OK;
}

int khip_comm_info(khip_ctx *ctx, int *rank, int *nranks, int *rccl_ranks, int *local_backend, int *halo_comm_separate) {
  KHIP_REQUIRE(ctx, "comm_info: null context");
  const Comm *c = ctx->comm;
  if (rank) *rank = c ? c->rank : 0;
  if (nranks) *nranks = c ? c->nranks : 1;
  if (rccl_ranks) *rccl_ranks = c ? c->rccl_ranks : 0;
  if (local_backend) *local_backend = (c && c->hub) ? 1 : 0;
  if (halo_comm_separate) *halo_comm_separate = (c && !c->hub && c->halo_comm != c->comm) ? 1 : 0;
  return KHIP_OK;
}

int khip_comm_init_local(khip_ctx *ctx, int rank, int nranks, int hub_id) {
  KHIP_REQUIRE(ctx && nranks >= 1 && rank >= 0 && rank < nranks, "comm_init_local: bad arguments");
  KHIP_REQUIRE(!ctx->comm, "comm_init_local: context already has a communicator");
  KHIP_CHECK_HIP(hipSetDevice(ctx->device));
  LocalHub *h = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_hub_mu);
    auto it = g_hubs.find(hub_id);
    if (it == g_hubs.end()) {
      h = new LocalHub();
      h->nranks = nranks;
      h->stage.resize(nranks);
      h->sendbuf.assign(nranks, nullptr);
      h->send_off.resize(nranks);
      g_hubs[hub_id] = h;
    } else {
      h = it->second;
    }
    KHIP_REQUIRE(h->nranks == nranks, "comm_init_local: hub %d was created for %d ranks", hub_id, h->nranks);
    h->joined++;
  }
  Comm *c = new Comm();
  c->rank = rank;
  c->nranks = nranks;
  c->hub = h;
  KHIP_CHECK_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->gather_pinned), sizeof(dd) * (size_t)kMaxRedOut * nranks,
                               hipHostMallocDefault));
  ctx->comm = c;
  return KHIP_OK;
}

int khip_comm_rank(khip_ctx *ctx, int *rank, int *nranks) {
  KHIP_REQUIRE(ctx, "comm_rank: null context");
  if (rank) *rank = ctx->comm ? ctx->comm->rank : 0;
  if (nranks) *nranks = ctx->comm ? ctx->comm->nranks : 1;
  return KHIP_OK;
}

int khip_comm_barrier(khip_ctx *ctx) {
  KHIP_REQUIRE(ctx, "comm_barrier: null context");
  if (!ctx->comm) return khip_ctx_sync(ctx);
  int64_t v = 0;
  std::vector<int64_t> all((size_t)ctx->comm->nranks);
  return allgather_host(ctx, &v, all.data(), sizeof(v));
}

// internal: called from khip_ctx_destroy
int khip_comm_destroy_internal(khip_ctx *ctx) {
  Comm *c = ctx->comm;
  if (!c) return KHIP_OK;
  if (c->halo_comm && c->halo_comm != c->comm && !c->hub) g_rccl.CommDestroy(c->halo_comm);
  if (c->comm && !c->hub) g_rccl.CommDestroy(c->comm);
  if (c->gather_dev) (void)hipFree(c->gather_dev);
  if (c->gather_pinned) (void)hipHostFree(c->gather_pinned);
  if (c->scratch_dev) (void)hipFree(c->scratch_dev);
  delete c;
  ctx->comm = nullptr;
  return KHIP_OK;
}

// Host-only helpers exported for the CPU (gloo) tests of the partition / halo logic.
int khip_ghost_columns_host(const int64_t *rowptr, const int32_t *col, int64_t m, int64_t row0, int32_t *out,
                            int64_t cap, int64_t *count) {
  KHIP_REQUIRE(rowptr && col && count, "ghost_columns_host: null argument");
  std::vector<int32_t> g;
  for (int64_t j = rowptr[0]; j < rowptr[m]; ++j)
    if (col[j] < row0 || col[j] >= row0 + m) g.push_back(col[j]);
  std::sort(g.begin(), g.end());
  g.erase(std::unique(g.begin(), g.end()), g.end());
  *count = (int64_t)g.size();
  KHIP_REQUIRE((int64_t)g.size() <= cap || out == nullptr, "ghost_columns_host: output too small");
  if (out) std::copy(g.begin(), g.end(), out);
  return KHIP_OK;
}

int khip_halo_plan_host(int rank, int nranks, const int64_t *row_starts, const int32_t *ghost_all,
                        const int64_t *ghost_off, int64_t *recv_off, int64_t *send_off, int32_t *send_idx,
                        int64_t send_cap) {
  KHIP_REQUIRE(row_starts && ghost_off && recv_off && send_off, "halo_plan_host: null argument");
  KHIP_REQUIRE(ghost_all || ghost_off[nranks] == 0, "halo_plan_host: null ghost list");
  std::vector<int64_t> ro, so;
  std::vector<int32_t> si;
  int rc = build_halo_plan_host(rank, nranks, row_starts, ghost_all, ghost_off, ro, so, si);
  if (rc != KHIP_OK) { set_error("halo_plan_host: inconsistent input"); return rc; }
  KHIP_REQUIRE((int64_t)si.size() <= send_cap, "halo_plan_host: send_idx capacity too small");
  std::copy(ro.begin(), ro.end(), recv_off);
  std::copy(so.begin(), so.end(), send_off);
  if (send_idx) std::copy(si.begin(), si.end(), send_idx);
  return KHIP_OK;
}

}  // extern "C"
