// api.cpp -- extern "C" entry points: context, device buffers, CSR handles, the k* primitives.
#include <cmath>
#include <cstdlib>

#include "khip_internal.hpp"

using namespace khip;

extern "C" int khip_comm_destroy_internal(khip_ctx *ctx);

namespace khip {
enum MapOpHost { H_COPY = 0, H_FILL, H_SCAL, H_SCALCOPY, H_DIVCOPY, H_AXPY, H_AXPBY, H_REF, H_WAXPY, H_VMUL, H_VDIV };
constexpr int kPadHost = 8;
}  // namespace khip

namespace khip {
static int g_optional_build_failures = 0;
void optional_build(int rc) {
  if (rc == KHIP_OK) return;
  const hipError_t e = (hipError_t)take_hip_error();            // the cause, recorded at the failing call (not inferred afterwards)
  (void)hipGetLastError();                                      // clears the runtime's sticky error of the failed call
  if (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation) return;   // a full device: the other kernels serve the product
  ++g_optional_build_failures;
  fprintf(stderr, "libkrylov_hip: an optional accelerator build failed (%d: %s; HIP: %s) -- falling back to the plain kernels\n", rc,
          khip_last_error(), hipGetErrorString(e));
}
}  // namespace khip

extern "C" {

// test-only (include/krylov_hip_test.h): the self-halo measurement hook is not an option of the product
int khip_test_set_halo_self(khip_ctx *ctx, int enable) {
  KHIP_REQUIRE(ctx, "test_set_halo_self: null context");
  ctx->tune.halo_self = enable ? 1 : 0;
  return KHIP_OK;
}

int khip_test_optional_build_failures(int *count) {
  KHIP_REQUIRE(count, "test_optional_build_failures: null output");
  *count = khip::g_optional_build_failures;
  return KHIP_OK;
}

void khip_version(int *major, int *minor) {
  if (major) *major = KHIP_VERSION_MAJOR;
  if (minor) *minor = KHIP_VERSION_MINOR;
}

// ------------------------------------------------------------------ context ----
int khip_device_count(int *count) {
  KHIP_REQUIRE(count, "device_count: null output");
  int ndev = 0;
  const hipError_t e = hipGetDeviceCount(&ndev);
  *count = e == hipSuccess ? ndev : 0;
  return KHIP_OK;
}

// PCI bus id of a visible device ("0000:05:00.0"): what identifies the PHYSICAL GPU behind a device index, so a launcher can
// tell two ranks that were handed the same GPU from two ranks on different ones (bench.py refuses the former).
int khip_device_pci_id(int device, char *buf, size_t cap) {
  KHIP_REQUIRE(buf && cap >= 16, "device_pci_id: buffer of at least 16 bytes needed");
  const hipError_t e = hipDeviceGetPCIBusId(buf, (int)cap, device);
  if (e != hipSuccess) { set_error("hipDeviceGetPCIBusId(%d): %s", device, hipGetErrorString(e)); return KHIP_ERR_HIP; }
  return KHIP_OK;
}

int khip_ctx_create(int device, void *stream, khip_ctx **out) {
  KHIP_REQUIRE(out, "ctx_create: null output");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) {
    set_error("no HIP device visible (%s): libkrylov_hip has no CPU fallback", hipGetErrorString(e));
    return KHIP_ERR_HIP;
  }
  KHIP_REQUIRE(device >= 0 && device < ndev, "ctx_create: device %d out of range (0..%d)", device, ndev - 1);
  KHIP_CHECK_HIP(hipSetDevice(device));
  khip_ctx *ctx = new khip_ctx();
  ctx->device = device;
  if (stream) {
    ctx->stream = static_cast<hipStream_t>(stream);
  } else {
    KHIP_CHECK_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    ctx->own_stream = true;
  }
  // The communication stream carries the halo exchange that is meant to run UNDER the interior rows of a product.  At default
  // priority its (small) RCCL kernel is dispatched behind the product's remaining workgroups and finishes only when the product
  // drains (rocprofv3 trace of the N = 8 slab iteration, profiles/r05b_slab8_one_iteration_trace.txt: 260 us under a 263 us
  // product) -- i.e. the transfer would start late on real links.  Highest priority lets its workgroups in first.
  // KHIP_COMM_PRIORITY=0 keeps the default priority (A/B).
  // Both streams exist; ctx option "comm_priority" (1 / 0) picks the one in use, so that bench.py can A/B it inside one launch.
  {
    int least = 0, greatest = 0;
    const char *pe = getenv("KHIP_COMM_PRIORITY");
    ctx->tune.comm_priority = (!pe || atoi(pe) != 0) ? 1 : 0;
    KHIP_CHECK_HIP(hipStreamCreateWithFlags(&ctx->comm_stream_lo, hipStreamNonBlocking));
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least) {
      KHIP_CHECK_HIP(hipStreamCreateWithPriority(&ctx->comm_stream_hi, hipStreamNonBlocking, greatest));
    } else {
      (void)hipGetLastError();
      ctx->comm_stream_hi = ctx->comm_stream_lo;
    }
    ctx->comm_stream = ctx->tune.comm_priority ? ctx->comm_stream_hi : ctx->comm_stream_lo;
  }
  for (int i = 0; i < khip_ctx::kEvRing; ++i) {
    KHIP_CHECK_HIP(hipEventCreateWithFlags(&ctx->ev_a[i], hipEventDisableTiming));
    KHIP_CHECK_HIP(hipEventCreateWithFlags(&ctx->ev_b[i], hipEventDisableTiming));
  }
  if (const char *e = getenv("KHIP_SPMV_DELTA")) ctx->tune.spmv_delta = atoi(e);   // likewise for the block-delta column stream
  if (const char *e = getenv("KHIP_SPMV_CODES")) ctx->tune.spmv_codes = atoi(e);   // tests force the coded column stream on small operators too (tests/conftest.py)
  hipDeviceProp_t prop;
  KHIP_CHECK_HIP(hipGetDeviceProperties(&prop, device));
  ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  KHIP_TRY(ensure_reduction_scratch(ctx, 1 << 18, 1));
  KHIP_CHECK_HIP(hipMalloc(&ctx->results, sizeof(double) * kResultSlots));
  KHIP_CHECK_HIP(hipMalloc(&ctx->results_dd, sizeof(dd) * kResultSlots));
  KHIP_CHECK_HIP(hipMemset(ctx->results, 0, sizeof(double) * kResultSlots));
  KHIP_CHECK_HIP(hipMemset(ctx->results_dd, 0, sizeof(dd) * kResultSlots));
  KHIP_CHECK_HIP(hipHostMalloc(reinterpret_cast<void **>(&ctx->results_pinned), sizeof(double) * kResultSlots * 2,
                               hipHostMallocDefault));
  *out = ctx;
  return KHIP_OK;
}

int khip_ctx_destroy(khip_ctx *ctx) {
  if (!ctx) return KHIP_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  khip_comm_destroy_internal(ctx);
  panel_scratch_destroy(ctx);
  for (hipEvent_t e : ctx->prof_events) (void)hipEventDestroy(e);
  (void)hipFree(ctx->partials);
  (void)hipFree(ctx->partials2);
  (void)hipFree(ctx->tickets);
  (void)hipFree(ctx->results);
  (void)hipFree(ctx->results_dd);
  (void)hipHostFree(ctx->results_pinned);
  if (ctx->ev_fetch) (void)hipEventDestroy(ctx->ev_fetch);
  if (ctx->ev_red) (void)hipEventDestroy(ctx->ev_red);
  for (int i = 0; i < khip_ctx::kEvRing; ++i) {
    if (ctx->ev_a[i]) (void)hipEventDestroy(ctx->ev_a[i]);
    if (ctx->ev_b[i]) (void)hipEventDestroy(ctx->ev_b[i]);
  }
  if (ctx->comm_stream_hi && ctx->comm_stream_hi != ctx->comm_stream_lo) (void)hipStreamDestroy(ctx->comm_stream_hi);
  if (ctx->comm_stream_lo) (void)hipStreamDestroy(ctx->comm_stream_lo);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return KHIP_OK;
}

int khip_ctx_sync(khip_ctx *ctx) {
  KHIP_REQUIRE(ctx, "ctx_sync: null context");
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return KHIP_OK;
}

void *khip_ctx_stream(khip_ctx *ctx) { return ctx ? ctx->stream : nullptr; }

static int *tuning_field(khip_ctx *ctx, const char *key) {
  Tuning &t = ctx->tune;
  struct { const char *k; int *p; } tab[] = {
      {"spmv_kernel", &t.spmv_kernel}, {"spmv_rows", &t.spmv_rows}, {"spmv_vec", &t.spmv_vec},
      {"spmv_nt", &t.spmv_nt},         {"spmv_xcd", &t.spmv_xcd},   {"spmv_lanes", &t.spmv_lanes},
      {"compensated", &t.compensated}, {"nt_min_elems", &t.nt_min_elems}, {"overlap_halo", &t.overlap_halo},
      {"profile_spmv", &t.profile_spmv}, {"comm_priority", &t.comm_priority}, {"spmv_persist", &t.spmv_persist}, {"spmv_nty", &t.spmv_nty}, {"spmv_dot_early", &t.spmv_dot_early}, {"spmv_blockptr", &t.spmv_blockptr}, {"spmv_codes", &t.spmv_codes}, {"spmv_delta", &t.spmv_delta}, {"spmv_blk_pub", &t.spmv_blk_pub}, {"spmv_stream_nt", &t.spmv_stream_nt}, {"spmv_sell", &t.spmv_sell}, {"spmv_sell_narrow", &t.spmv_sell_narrow}, {"spmv_sell_pair", &t.spmv_sell_pair}, {"cg_setup_fused", &t.cg_setup_fused}, {"spmv_wide", &t.spmv_wide}, {"spmv_pipe", &t.spmv_pipe}, {"spmv_fake_gather", &t.spmv_fake_gather}, {"spmv_tiles", &t.spmv_tiles}, {"spmv_lds_pad", &t.spmv_lds_pad}, {"spmv_cap", &t.spmv_cap}, {"spmv_template", &t.spmv_template}, {"spmv_sweep_s", &t.spmv_sweep_s}, {"spmv_sweep_w", &t.spmv_sweep_w}, {"spmv_tmpl_rows", &t.spmv_tmpl_rows}, {"mgs_keep", &t.mgs_keep}, {"spmm_sweep", &t.spmm_sweep}, {"spmm_win_sweep", &t.spmm_win_sweep}, {"spmm_wide", &t.spmm_wide}, {"panel_fuse", &t.panel_fuse}, {"panel_signs", &t.panel_signs}, {"panel_qr_tsqr", &t.panel_qr_tsqr}, {"panel_a_lds", &t.panel_a_lds}, {"panel_nt", &t.panel_nt}, {"gmres_sstep", &t.gmres_sstep}, {"panel_multi_tiles", &t.panel_multi_tiles}, {"ilu_blocks", &t.ilu_blocks}, {"halo_mode", &t.halo_mode}, {"halo_gather_pct", &t.halo_gather_pct}, {"spmm_window", &t.spmm_window}, {"spmm_tile", &t.spmm_tile}, {"spmm_tile_exp", &t.spmm_tile_exp}, {"spmm_tile_nt", &t.spmm_tile_nt}, {"spmm_tile_slices", &t.spmm_tile_slices}, {"spmm_tile_pencil", &t.spmm_tile_pencil}, {"spmm_tile_slide", &t.spmm_tile_slide}, {"spmm_tile_ahead", &t.spmm_tile_ahead}, {"spmm_tile_xcd", &t.spmm_tile_xcd}, {"spmm_tile_dbuf", &t.spmm_tile_dbuf}, {"spmm_tile_pair", &t.spmm_tile_pair}, {"spmm_tile_waves", &t.spmm_tile_waves}, {"spmm_tile_grid", &t.spmm_tile_grid}, {"spmm_tile_shape", &t.spmm_tile_shape}, {"spmm_window_grid", &t.spmm_window_grid}, {"spmm_sweep_s", &t.spmm_sweep_s}, {"spmm_sweep_w", &t.spmm_sweep_w}, {"red_u", &t.red_u}, {"hist_window", &t.hist_window}};
  for (auto &e : tab)
    if (strcmp(e.k, key) == 0) return e.p;
  return nullptr;
}

int khip_ctx_set_option(khip_ctx *ctx, const char *key, int value) {
  KHIP_REQUIRE(ctx && key, "set_option: null argument");
  int *p = tuning_field(ctx, key);
  KHIP_REQUIRE(p, "set_option: unknown key '%s'", key);
  *p = value;
  if (p == &ctx->tune.comm_priority) {            // switch the communication stream (everything on both has drained first)
    KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    KHIP_CHECK_HIP(hipStreamSynchronize(ctx->comm_stream_hi));
    KHIP_CHECK_HIP(hipStreamSynchronize(ctx->comm_stream_lo));
    ctx->comm_stream = value ? ctx->comm_stream_hi : ctx->comm_stream_lo;
  }
  return KHIP_OK;
}

int khip_ctx_get_option(khip_ctx *ctx, const char *key, int *value) {
  KHIP_REQUIRE(ctx && key && value, "get_option: null argument");
  int *p = tuning_field(ctx, key);
  KHIP_REQUIRE(p, "get_option: unknown key '%s'", key);
  *value = *p;
  return KHIP_OK;
}

// ------------------------------------------------------------------ buffers ----
int khip_malloc(khip_ctx *ctx, size_t bytes, void **dptr) {
  KHIP_REQUIRE(ctx && dptr, "malloc: null argument");
  KHIP_CHECK_HIP(hipSetDevice(ctx->device));
  *dptr = nullptr;
  if (bytes == 0) return KHIP_OK;   // S(undef, 0): the lazily-allocated vectors of the workspaces
  KHIP_CHECK_HIP(hipMalloc(dptr, bytes));
  return KHIP_OK;
}

int khip_free(khip_ctx *ctx, void *dptr) {
  KHIP_REQUIRE(ctx, "free: null context");
  if (!dptr) return KHIP_OK;
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  KHIP_CHECK_HIP(hipFree(dptr));
  return KHIP_OK;
}

int khip_memcpy_h2d(khip_ctx *ctx, void *dst, const void *src_host, size_t bytes) {
  KHIP_REQUIRE(ctx && (bytes == 0 || (dst && src_host)), "memcpy_h2d: null argument");
  if (!bytes) return KHIP_OK;
  KHIP_CHECK_HIP(hipMemcpyAsync(dst, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return KHIP_OK;
}

int khip_memcpy_d2h(khip_ctx *ctx, void *dst_host, const void *src, size_t bytes) {
  KHIP_REQUIRE(ctx && (bytes == 0 || (dst_host && src)), "memcpy_d2h: null argument");
  if (!bytes) return KHIP_OK;
  KHIP_CHECK_HIP(hipMemcpyAsync(dst_host, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return KHIP_OK;
}

int khip_memcpy_d2d(khip_ctx *ctx, void *dst, const void *src, size_t bytes) {
  KHIP_REQUIRE(ctx && (bytes == 0 || (dst && src)), "memcpy_d2d: null argument");
  if (!bytes) return KHIP_OK;
  KHIP_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return KHIP_OK;
}

int khip_mem_info(khip_ctx *ctx, size_t *free_bytes, size_t *total_bytes) {
  KHIP_REQUIRE(ctx, "mem_info: null context");
  size_t f = 0, t = 0;
  KHIP_CHECK_HIP(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return KHIP_OK;
}

// ------------------------------------------------------------------ CSR --------
static int csr_upload(khip_ctx *ctx, khip_csr *A, const void *rowptr, int rowptr_bits, const int32_t *col,
                      const double *val, int index_base, int on_device) {
  const int64_t m = A->m, nnz = A->nnz;
  const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  KHIP_CHECK_HIP(hipMalloc(&A->rowptr, sizeof(int32_t) * (size_t)(m + 1)));
  KHIP_CHECK_HIP(hipMalloc(&A->col, sizeof(int32_t) * (size_t)(nnz + kPadHost)));
  KHIP_CHECK_HIP(hipMalloc(&A->val, sizeof(double) * (size_t)(nnz + kPadHost)));
  KHIP_CHECK_HIP(hipMemsetAsync(A->col + nnz, 0, sizeof(int32_t) * kPadHost, ctx->stream));
  KHIP_CHECK_HIP(hipMemsetAsync(A->val + nnz, 0, sizeof(double) * kPadHost, ctx->stream));
  if (nnz) {
    KHIP_CHECK_HIP(hipMemcpyAsync(A->col, col, sizeof(int32_t) * (size_t)nnz, kind, ctx->stream));
    KHIP_CHECK_HIP(hipMemcpyAsync(A->val, val, sizeof(double) * (size_t)nnz, kind, ctx->stream));
  }
  if (rowptr_bits == 32) {
    KHIP_CHECK_HIP(hipMemcpyAsync(A->rowptr, rowptr, sizeof(int32_t) * (size_t)(m + 1), kind, ctx->stream));
  } else {
    // 64-bit row pointers are narrowed on the host (a shard's nnz fits int32 by contract)
    std::vector<int64_t> tmp64((size_t)(m + 1));
    if (on_device) KHIP_CHECK_HIP(hipMemcpy(tmp64.data(), rowptr, sizeof(int64_t) * (size_t)(m + 1), hipMemcpyDeviceToHost));
    else memcpy(tmp64.data(), rowptr, sizeof(int64_t) * (size_t)(m + 1));
    std::vector<int32_t> tmp32((size_t)(m + 1));
    for (int64_t i = 0; i <= m; ++i) {
      const int64_t v = tmp64[(size_t)i];
      if (v < INT32_MIN || v > INT32_MAX) {
        set_error("csr_create: row pointer %lld at row %lld does not fit the shard's int32 indexing", (long long)v, (long long)i);
        return KHIP_ERR_INVALID;
      }
      tmp32[(size_t)i] = (int32_t)v;
    }
    KHIP_CHECK_HIP(hipMemcpy(A->rowptr, tmp32.data(), sizeof(int32_t) * (size_t)(m + 1), hipMemcpyHostToDevice));
  }
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  return KHIP_OK;
}

}  // extern "C"

namespace khip {
int launch_index_shift(khip_ctx *ctx, int32_t *data, int64_t n, int32_t delta);   // spmv.hip
int comm_nranks(const khip_ctx *ctx);                                              // comm.cpp
}

extern "C" {

static int csr_create_common(khip_ctx *ctx, int64_t m, int64_t n, int64_t nnz, const void *rowptr, int rowptr_bits,
                             const int32_t *col, const double *val, int index_base, int on_device, khip_csr **out) {
  KHIP_REQUIRE(ctx && out && rowptr, "csr_create: null argument");
  KHIP_REQUIRE(m >= 0 && n >= 0 && nnz >= 0, "csr_create: negative size");
  KHIP_REQUIRE(nnz == 0 || (col && val), "csr_create: null col/val");
  KHIP_REQUIRE(rowptr_bits == 32 || rowptr_bits == 64, "csr_create: rowptr_bits must be 32 or 64");
  KHIP_REQUIRE(index_base == 0 || index_base == 1, "csr_create: index_base must be 0 or 1");
  KHIP_REQUIRE(nnz < (1ll << 31) - 64 && n < (1ll << 31) && m < (1ll << 31),
               "csr_create: shard exceeds int32 indexing (nnz=%lld)", (long long)nnz);
  KHIP_CHECK_HIP(hipSetDevice(ctx->device));
  khip_csr *A = new khip_csr();
  A->ctx = ctx; A->m = m; A->n = n; A->nnz = nnz;
  int rc = csr_upload(ctx, A, rowptr, rowptr_bits, col, val, index_base, on_device);
  if (rc == KHIP_OK && index_base != 0) {
    rc = launch_index_shift(ctx, A->rowptr, m + 1, -index_base);
    if (rc == KHIP_OK) rc = launch_index_shift(ctx, A->col, nnz, -index_base);
  }
  if (rc == KHIP_OK) rc = csr_finalize(ctx, A);
  if (rc != KHIP_OK) { khip_csr_destroy(A); return rc; }
  *out = A;
  return KHIP_OK;
}

int khip_csr_create(khip_ctx *ctx, int64_t m, int64_t n, int64_t nnz, const void *rowptr, int rowptr_bits,
                    const int32_t *col, const double *val, int index_base, int on_device, khip_csr **out) {
  return csr_create_common(ctx, m, n, nnz, rowptr, rowptr_bits, col, val, index_base, on_device, out);
}

int khip_csr_create_dist(khip_ctx *ctx, int64_t n_global, int64_t row0, int64_t m, int64_t nnz, const void *rowptr,
                         int rowptr_bits, const int32_t *col, const double *val, int index_base, int on_device,
                         khip_csr **out) {
  KHIP_REQUIRE(ctx && ctx->comm, "csr_create_dist: call khip_comm_init first");
  KHIP_REQUIRE(row0 >= 0 && row0 + m <= n_global, "csr_create_dist: row range outside the operator");
  khip_csr *A = nullptr;
  // a failure of this rank alone (validation, index narrowing, a HIP error) still enters the plan's first collective, with
  // the failure in its status word: all ranks return an error together instead of the others blocking in the all-gather
  const int rc_local = csr_create_common(ctx, m, n_global, nnz, rowptr, rowptr_bits, col, val, index_base, on_device, &A);
  if (rc_local == KHIP_OK) {
    A->dist = true;
    A->n_global = n_global;
    A->row0 = row0;
  } else {
    A = nullptr;
  }
  int rc = comm_build_plan(ctx, A, rc_local);
  if (rc != KHIP_OK) { khip_csr_destroy(A); return rc; }
  *out = A;
  return KHIP_OK;
}

int khip_csr_transpose(khip_ctx *ctx, const khip_csr *A, khip_csr **out) {
  KHIP_REQUIRE(ctx && A && out, "csr_transpose: null argument");
  if (A->dist) return comm_transpose_dist(ctx, A, out);       // row-partitioned: all-to-all of the entries, same partition
  khip_csr *T = new khip_csr();
  int rc = csr_transpose(ctx, A, T);
  if (rc != KHIP_OK) { khip_csr_destroy(T); return rc; }
  *out = T;
  return KHIP_OK;
}

int khip_csr_destroy(khip_csr *A) {
  if (!A) return KHIP_OK;
  if (A->ctx) (void)hipStreamSynchronize(A->ctx->stream);
  (void)hipFree(A->rowptr); (void)hipFree(A->col); (void)hipFree(A->val); (void)hipFree(A->blockptr);
  (void)hipFree(A->ghost); (void)hipFree(A->sendbuf); (void)hipFree(A->send_idx);
  (void)hipFree(A->ghost_w); (void)hipFree(A->sendbuf_w);
  csr_free_templates(A);
  csr_free_window(A);
  csr_free_tiles(A);
  csr_free_delta(A);
  csr_free_codes(A);
  delete A;
  return KHIP_OK;
}

int khip_csr_shape(const khip_csr *A, int64_t *m, int64_t *n, int64_t *nnz) {
  KHIP_REQUIRE(A, "csr_shape: null handle");
  if (m) *m = A->m;
  if (n) *n = A->n;
  if (nnz) *nnz = A->nnz;
  return KHIP_OK;
}

int khip_csr_halo_info(const khip_csr *A, int *gather_mode, int64_t *n_ghost, int64_t *n_send) {
  KHIP_REQUIRE(A, "csr_halo_info: null handle");
  if (gather_mode) *gather_mode = A->gather ? 1 : 0;
  if (n_ghost) *n_ghost = A->n_ghost;
  if (n_send) *n_send = A->gather ? A->m : A->n_send;
  return KHIP_OK;
}

int khip_csr_sell_info(const khip_csr *A, int *state, int *units_per_slice, int64_t *total_units) {
  KHIP_REQUIRE(A, "csr_sell_info: null handle");
  if (state) *state = A->sell_state;
  if (units_per_slice) *units_per_slice = A->sell_state == 1 ? A->sell_units : 0;
  if (total_units) *total_units = A->sell_state == 1 ? A->sell_total_units : 0;
  return KHIP_OK;
}

int khip_csr_sell_narrow(const khip_csr *A, int *narrow) {
  KHIP_REQUIRE(A && narrow, "csr_sell_narrow: null argument");
  *narrow = (A->sell_state == 1 && A->sell_c4) ? 1 : 0;
  return KHIP_OK;
}

int khip_csr_sell32_info(const khip_csr *A, int *state, int *units_per_slice, int64_t *total_units) {
  KHIP_REQUIRE(A, "csr_sell32_info: null handle");
  if (state) *state = A->sell32_state;
  if (units_per_slice) *units_per_slice = A->sell32_state == 1 ? A->sell32_units : 0;
  if (total_units) *total_units = A->sell32_state == 1 ? A->sell32_total_units : 0;
  return KHIP_OK;
}

int khip_csr_code_info(const khip_csr *A, int *bits, int *diagonals) {
  KHIP_REQUIRE(A, "csr_code_info: null handle");
  if (bits) *bits = A->code_state == 1 ? A->code_bits : 32;
  if (diagonals) *diagonals = A->code_state == 1 ? A->code_T : 0;
  return KHIP_OK;
}

int khip_csr_delta_info(const khip_csr *A, int *bits, int *rows, int64_t *escapes) {
  KHIP_REQUIRE(A, "csr_delta_info: null handle");
  const bool on = A->delta_state == 1;
  if (bits) *bits = on ? A->delta_bits : 32;
  if (rows) *rows = on ? A->delta_rows : 0;
  if (escapes) *escapes = on ? A->delta_esc : 0;
  return KHIP_OK;
}

int khip_spmv_kernel_info(khip_ctx *ctx, const khip_csr *A, int *kernel) {
  KHIP_REQUIRE(ctx && A && kernel, "spmv_kernel_info: null argument");
  *kernel = spmv_kernel_choice(ctx, A);
  return KHIP_OK;
}

int khip_csr_tile_info(const khip_csr *A, int *state, int *window, int *grid_tiles, int64_t *groups, int64_t *direct_groups,
                       double *reuse) {
  KHIP_REQUIRE(A, "csr_tile_info: null handle");
  const bool on = A->tile_state == 1;
  if (state) *state = A->tile_state;
  if (window) *window = on ? A->tile_cap : 0;
  if (grid_tiles) *grid_tiles = on ? A->tile_grid : 0;
  if (groups) *groups = on ? A->tile_groups : 0;
  if (direct_groups) *direct_groups = on ? A->tile_direct : 0;
  if (reuse) *reuse = on ? A->tile_reuse : 0.0;
  return KHIP_OK;
}

int khip_spmv_bytes(const khip_csr *A, int64_t *bytes) {
  KHIP_REQUIRE(A && bytes, "spmv_bytes: null argument");
  const int64_t ncols_read = A->dist ? A->m + A->n_ghost : A->n;
  *bytes = 12 * A->nnz + 4 * (A->m + 1) + 8 * ncols_read + 8 * A->m;
  return KHIP_OK;
}

// y <- A x (optionally fused with x . y into results[dot_slot]), handling the halo exchange and the
// interior / boundary split of distributed handles.  The up to three launches of a split product feed a
// single finish kernel, so every rank contributes exactly one partial to the all-reduce.
}  // extern "C"
int khip::spmv_any(khip_ctx *ctx, const khip_csr *A, const double *x, double *y, int dot_slot, const double *dotw,
                   int dot_sq) {
  if (dot_sq && spmv_kernel_choice(ctx, A) != 4 && spmv_kernel_choice(ctx, A) != 5 && spmv_kernel_choice(ctx, A) != 1 && spmv_kernel_choice(ctx, A) != 6) {   // the ordered / vector kernels do not carry the second reduction
    // two reductions instead of one: inside a device-resident loop the scalar epilogue must see BOTH results, so it is
    // taken off the two finish kernels and run on its own afterwards (it used to run after the first one, on a stale second
    // scalar: bicgstab!(fused = 2) diverged on operators that do not take the staged kernel)
    const int epi = ctx->ctl.epi;
    ctx->ctl.epi = 0;
    int rc = spmv_any(ctx, A, x, y, dot_slot, dotw, 0);
    if (rc == KHIP_OK) rc = launch_nrm2sq(ctx, A->m, dot_sq == 1 ? y : (dotw ? dotw : x), dot_slot + 1);
    ctx->ctl.epi = epi;
    // with a communicator the cross-rank combine (comm_allreduce_dd_device) owns the epilogue and runs it ONCE on the
    // global values, exactly as launch_finish leaves it to it; running it here too applied it twice, first on rank-local
    // scalars
    if (rc == KHIP_OK && epi != 0 && !ctx->comm) rc = launch_epilogue_only(ctx, dot_slot);
    return rc;
  }
  if (!A->dist || !ctx->comm) return launch_spmv(ctx, A, x, y, dot_slot, 0, A->m, nullptr, true, dotw, dot_sq);
  KHIP_TRY(comm_halo_exchange_begin(ctx, A, x));
  const bool split = ctx->tune.overlap_halo && A->interior_hi > A->interior_lo;
  if (!split) {
    KHIP_TRY(comm_halo_exchange_end(ctx, A));
    return launch_spmv(ctx, A, x, y, dot_slot, 0, A->m, nullptr, true, dotw, dot_sq);
  }
  int64_t cursor = 0;
  KHIP_TRY(launch_spmv(ctx, A, x, y, dot_slot, A->interior_lo, A->interior_hi, &cursor, false, dotw, dot_sq));
  KHIP_TRY(comm_halo_exchange_end(ctx, A));
  // the two boundary ranges [0, interior_lo) and [interior_hi, m) as ONE launch where the kernel takes two ranges (round 5: one
  // launch and ~10 us less per product at the N = 8 slab shape); same partials in the same order: the fused dot is unchanged
  ctx->prof_spmv_tag = kProfSpmvBoundary;
  const int rc_b = launch_spmv(ctx, A, x, y, dot_slot, 0, A->m, &cursor, true, dotw, dot_sq, A->interior_lo, A->interior_hi);
  ctx->prof_spmv_tag = kProfSpmv;
  return rc_b;
}
extern "C" {

int khip_spmv(khip_ctx *ctx, const khip_csr *A, const double *x, double *y) {
  KHIP_REQUIRE(ctx && A && x && y, "spmv: null argument");
  KHIP_REQUIRE(x != y, "spmv: x and y must not alias");
  return spmv_any(ctx, A, x, y, -1);
}

int khip_spmv_dot(khip_ctx *ctx, const khip_csr *A, const double *x, double *y, double *result_host) {
  KHIP_REQUIRE(ctx && A && x && y && result_host, "spmv_dot: null argument");
  KHIP_REQUIRE(x != y, "spmv_dot: x and y must not alias");
  const int slot = take_slots(ctx, 1);
  KHIP_TRY(spmv_any(ctx, A, x, y, slot));
  return fetch_results(ctx, slot, 1, result_host);
}

int khip_spmv_dotw(khip_ctx *ctx, const khip_csr *A, const double *x, double *y, const double *w, double *result_host) {
  KHIP_REQUIRE(ctx && A && x && y && w && result_host, "spmv_dotw: null argument");
  KHIP_REQUIRE(x != y && w != y, "spmv_dotw: y must not alias x or w");
  const int slot = take_slots(ctx, 1);
  KHIP_TRY(spmv_any(ctx, A, x, y, slot, w, 0));
  return fetch_results(ctx, slot, 1, result_host);
}

int khip_spmv_dot2(khip_ctx *ctx, const khip_csr *A, const double *x, double *y, double *result_host) {
  KHIP_REQUIRE(ctx && A && x && y && result_host, "spmv_dot2: null argument");
  KHIP_REQUIRE(x != y, "spmv_dot2: x and y must not alias");
  const int slot = take_slots(ctx, 2);
  KHIP_TRY(spmv_any(ctx, A, x, y, slot, nullptr, 1));
  return fetch_results(ctx, slot, 2, result_host);
}

int khip_spmv_bytes_stored(const khip_csr *A, int64_t *bytes) {
  KHIP_REQUIRE(A && bytes, "spmv_bytes_stored: null argument");
  const int64_t ncols_read = A->dist ? A->m + A->n_ghost : A->n;
  const int sell_opt = A->ctx ? A->ctx->tune.spmv_sell : 0, codes_opt = A->ctx ? A->ctx->tune.spmv_codes : 1;
  const bool coded = A->code_state == 1 && codes_opt != 0;               // what launch_spmv reads under the context's CURRENT options
  const int64_t slices = (A->m + 63) / 64;
  if (A->tmpl_id) *bytes = 2 * A->m + 8 * ncols_read + 8 * A->m;        // template id + x + y
  else if (coded && A->sell_state == 1 && sell_opt)                     // sliced form of the coded operator: 512 B per unit (+ 4 B per slice of offsets, + 4 B per row of narrow codes)
    *bytes = 512 * A->sell_total_units + (A->sell_off ? 4 * (slices + 1) : 0) + (A->sell_c4 ? 4 * 64 * slices : 0) + 8 * ncols_read + 8 * A->m;
  else if (!coded && A->sell32_state == 1 && sell_opt)                  // sliced form with int32 columns
    *bytes = 512 * A->sell32_total_units + (A->sell32_off ? 4 * (slices + 1) : 0) + 8 * ncols_read + 8 * A->m;
  else if (coded)                                                       // coded columns (colcode.hip): 1 or 2 B per entry
    *bytes = (8 + A->code_bits / 8) * A->nnz + 4 * (A->m + 1) + 8 * ncols_read + 8 * A->m;
  else if (A->delta_state == 1)                                         // block-delta columns (coldelta.hip): codes + 6 B per escape + 8 B per block
    *bytes = (8 + A->delta_bits / 8) * A->nnz + 6 * A->delta_esc + 8 * ((A->m + A->delta_rows - 1) / A->delta_rows) +
             4 * (A->m + 1) + 8 * ncols_read + 8 * A->m;
  else *bytes = 12 * A->nnz + 4 * (A->m + 1) + 8 * ncols_read + 8 * A->m;
  return KHIP_OK;
}

int khip_profile_kernels(khip_ctx *ctx, int ntags, int64_t *launches, double *total_ms) {
  KHIP_REQUIRE(ctx && launches && total_ms && ntags >= 1, "profile_kernels: bad argument");
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->comm_stream) KHIP_CHECK_HIP(hipStreamSynchronize(ctx->comm_stream));      // halo / dot brackets live there
  for (int t = 0; t < ntags; ++t) { launches[t] = 0; total_ms[t] = 0.0; }
  for (size_t i = 0; i + 1 < ctx->prof_used; i += 2) {
    float ms = 0;
    KHIP_CHECK_HIP(hipEventElapsedTime(&ms, ctx->prof_events[i], ctx->prof_events[i + 1]));
    const int tag = i / 2 < ctx->prof_tags.size() ? ctx->prof_tags[i / 2] : 0;
    if (tag < ntags) { launches[tag] += 1; total_ms[tag] += ms; }
  }
  ctx->prof_used = 0;
  return KHIP_OK;
}

int khip_profile_spmv(khip_ctx *ctx, int64_t *launches, double *total_ms) {      // the SpMV brackets alone (tag 0); resets every counter
  return khip_profile_kernels(ctx, 1, launches, total_ms);
}

int khip_spmm(khip_ctx *ctx, const khip_csr *A, const double *X, double *Y, int p) {
  KHIP_REQUIRE(ctx && A && X && Y, "spmm: null argument");
  return launch_spmm(ctx, A, X, Y, p);
}

// ------------------------------------------------------------------ BLAS-1 -----
int khip_dot(khip_ctx *ctx, int64_t n, const double *x, const double *y, double *result_host) {
  KHIP_REQUIRE(ctx && result_host && (n == 0 || (x && y)), "dot: null argument");
  const int slot = take_slots(ctx, 1);
  KHIP_TRY(launch_dot(ctx, n, x, y, slot));
  return fetch_results(ctx, slot, 1, result_host);
}

int khip_nrm2(khip_ctx *ctx, int64_t n, const double *x, double *result_host) {
  KHIP_REQUIRE(ctx && result_host && (n == 0 || x), "nrm2: null argument");
  const int slot = take_slots(ctx, 1);
  KHIP_TRY(launch_nrm2sq(ctx, n, x, slot));
  double sq = 0;
  KHIP_TRY(fetch_results(ctx, slot, 1, &sq));
  *result_host = std::sqrt(sq);
  return KHIP_OK;
}

int khip_dot2(khip_ctx *ctx, int64_t n, const double *x, const double *y, double *result_host) {
  KHIP_REQUIRE(ctx && result_host && (n == 0 || (x && y)), "dot2: null argument");
  const int slot = take_slots(ctx, 2);
  KHIP_TRY(launch_dot2(ctx, n, x, y, slot));
  return fetch_results(ctx, slot, 2, result_host);
}

#define KHIP_VEC_ARGS(name, cond) KHIP_REQUIRE(ctx && (n == 0 || (cond)), name ": null argument")

int khip_scal(khip_ctx *ctx, int64_t n, double s, double *x) {
  KHIP_VEC_ARGS("scal", x);
  return launch_map(ctx, H_SCAL, n, s, 0, nullptr, x, nullptr);
}
int khip_div(khip_ctx *ctx, int64_t n, double *x, double s) {   // kdiv! = kscal!(1/s)  (src/krylov_utils.jl:325)
  KHIP_VEC_ARGS("div", x);
  return launch_map(ctx, H_SCAL, n, 1.0 / s, 0, nullptr, x, nullptr);
}
int khip_copy(khip_ctx *ctx, int64_t n, double *y, const double *x) {
  KHIP_VEC_ARGS("copy", x && y);
  if (x == y) return KHIP_OK;
  return launch_map(ctx, H_COPY, n, 0, 0, x, y, nullptr);
}
int khip_scalcopy(khip_ctx *ctx, int64_t n, double *y, double s, const double *x) {
  KHIP_VEC_ARGS("scalcopy", x && y);
  return launch_map(ctx, H_SCALCOPY, n, s, 0, x, y, nullptr);
}
int khip_divcopy(khip_ctx *ctx, int64_t n, double *y, const double *x, double s) {
  KHIP_VEC_ARGS("divcopy", x && y);
  return launch_map(ctx, H_DIVCOPY, n, s, 0, x, y, nullptr);
}
int khip_axpy(khip_ctx *ctx, int64_t n, double s, const double *x, double *y) {
  KHIP_VEC_ARGS("axpy", x && y);
  return launch_map(ctx, H_AXPY, n, s, 0, x, y, nullptr);
}
int khip_axpby(khip_ctx *ctx, int64_t n, double s, const double *x, double t, double *y) {
  KHIP_VEC_ARGS("axpby", x && y);
  return launch_map(ctx, H_AXPBY, n, s, t, x, y, nullptr);
}
int khip_fill(khip_ctx *ctx, int64_t n, double *x, double val) {
  KHIP_VEC_ARGS("fill", x);
  return launch_map(ctx, H_FILL, n, val, 0, nullptr, x, nullptr);
}
int khip_ref(khip_ctx *ctx, int64_t n, double *x, double *y, double c, double s) {
  KHIP_VEC_ARGS("ref", x && y);
  KHIP_REQUIRE(x != y, "ref: x and y must be distinct");
  return launch_map(ctx, H_REF, n, c, s, x, y, nullptr);
}
int khip_waxpy(khip_ctx *ctx, int64_t n, double *w, const double *x, double b, const double *y) {
  KHIP_VEC_ARGS("waxpy", w && x && y);
  // launch_map's WAXPY computes w = fma(b, Y, X) with X = x-argument, Y = y-argument
  return launch_map(ctx, H_WAXPY, n, 0, b, x, const_cast<double *>(y), w);
}

int khip_vmul(khip_ctx *ctx, int64_t n, double *w, const double *x, const double *y) {
  KHIP_VEC_ARGS("vmul", w && x && y);
  return launch_map(ctx, H_VMUL, n, 0, 0, x, const_cast<double *>(y), w);
}
int khip_vdiv(khip_ctx *ctx, int64_t n, double *w, const double *x, const double *y) {
  KHIP_VEC_ARGS("vdiv", w && x && y);
  return launch_map(ctx, H_VDIV, n, 0, 0, x, const_cast<double *>(y), w);
}
int khip_csr_diagonal(khip_ctx *ctx, const khip_csr *A, double *diag) {
  KHIP_REQUIRE(ctx && A && diag, "csr_diagonal: null argument");
  return launch_diagonal(ctx, A, diag);
}

// Jacobi preconditioner as a khip_operator: z <- r ./ diag(A)
struct khip_jacobi { khip_ctx *ctx; int64_t n; double *diag; };
static int jacobi_apply(void *self, const double *x, double *y) {
  khip_jacobi *j = static_cast<khip_jacobi *>(self);
  return khip_vdiv(j->ctx, j->n, y, x, j->diag);
}
int khip_jacobi_create(khip_ctx *ctx, const khip_csr *A, khip_operator *op_out) {
  KHIP_REQUIRE(ctx && A && op_out, "jacobi_create: null argument");
  khip_jacobi *j = new khip_jacobi{ctx, A->m, nullptr};
  int rc = khip_malloc(ctx, sizeof(double) * (size_t)((A->m + 33) & ~(int64_t)31), reinterpret_cast<void **>(&j->diag));
  if (!rc) rc = launch_diagonal(ctx, A, j->diag);
  if (rc) { khip_free(ctx, j->diag); delete j; return rc; }
  op_out->csr = nullptr;
  op_out->apply = jacobi_apply;
  op_out->self = j;
  return KHIP_OK;
}
int khip_jacobi_destroy(khip_operator *op) {
  if (!op || op->apply != jacobi_apply || !op->self) return KHIP_OK;
  khip_jacobi *j = static_cast<khip_jacobi *>(op->self);
  khip_free(j->ctx, j->diag);
  delete j;
  op->self = nullptr;
  return KHIP_OK;
}

int khip_axpy2_dot(khip_ctx *ctx, int64_t n, double a, const double *p, const double *q, double *x, double *r,
                   double *result_host) {
  KHIP_REQUIRE(ctx && result_host && (n == 0 || (p && q && x && r)), "axpy2_dot: null argument");
  const int slot = take_slots(ctx, 1);
  KHIP_TRY(launch_axpy2_dot(ctx, n, a, p, q, x, r, slot));
  return fetch_results(ctx, slot, 1, result_host);
}

int khip_axpy_sqnorm(khip_ctx *ctx, int64_t n, double a, const double *x, double *y, double *result_host) {
  KHIP_REQUIRE(ctx && result_host && n >= 0 && (n == 0 || (x && y)), "axpy_sqnorm: null argument");
  const int slot = take_slots(ctx, 1);
  KHIP_TRY(launch_axpy_sqnorm(ctx, n, a, x, y, slot));
  return fetch_results(ctx, slot, 1, result_host);
}

int khip_cg_setup(khip_ctx *ctx, int64_t n, const double *b, double *x, double *r, double *p, double *result_host) {
  KHIP_REQUIRE(ctx && result_host && n >= 0 && (n == 0 || (b && x && r && p)), "cg_setup: null argument");
  KHIP_REQUIRE(b != x && b != r && b != p && x != r && x != p && r != p, "cg_setup: the four vectors must be distinct");
  const int slot = take_slots(ctx, 1);
  KHIP_TRY(launch_cg_setup(ctx, n, b, x, r, p, slot));
  return fetch_results(ctx, slot, 1, result_host);
}

int khip_cg_update(khip_ctx *ctx, int64_t n, double a, double b, const double *r, double *p, double *x) {
  KHIP_REQUIRE(ctx && n >= 0 && (n == 0 || (r && p && x)), "cg_update: null argument");
  return launch_cg_update(ctx, n, a, b, r, p, x);
}

int khip_bicgstab_sx(khip_ctx *ctx, int64_t n, double alpha, const double *r, const double *v, const double *y, double *s,
                     double *x) {
  KHIP_REQUIRE(ctx && n >= 0 && (n == 0 || (r && v && y && s && x)), "bicgstab_sx: null argument");
  return launch_bicg_sx(ctx, n, alpha, r, v, y, s, x);
}

int khip_bicgstab_xr(khip_ctx *ctx, int64_t n, double omega, const double *s, const double *t, const double *z,
                     const double *c, double *x, double *r, double *result_host) {
  KHIP_REQUIRE(ctx && result_host && n >= 0 && (n == 0 || (s && t && z && c && x && r)), "bicgstab_xr: null argument");
  const int slot = take_slots(ctx, 2);
  KHIP_TRY(launch_bicg_xr(ctx, n, omega, s, t, z, c, x, r, slot));
  return fetch_results(ctx, slot, 2, result_host);
}

int khip_bicgstab_p(khip_ctx *ctx, int64_t n, double omega, double beta, const double *v, const double *r, double *p) {
  KHIP_REQUIRE(ctx && n >= 0 && (n == 0 || (v && r && p)), "bicgstab_p: null argument");
  return launch_bicg_p(ctx, n, omega, beta, v, r, p);
}

int khip_mgs(khip_ctx *ctx, int64_t n, int k, const double *const *V_host, double *q, double *h_host,
             double *nrm_host, int accumulate) {
  KHIP_REQUIRE(ctx && q && (k == 0 || (V_host && h_host)), "mgs: null argument");
  KHIP_REQUIRE(k >= 0, "mgs: negative k");
  if (k + 1 > kResultSlots) {
    // a basis larger than the device scalar ring: the reference's own sequence, one sync per coefficient
    for (int i = 0; i < k; ++i) {
      double h;
      KHIP_TRY(khip_dot(ctx, n, V_host[i], q, &h));
      KHIP_TRY(khip_axpy(ctx, n, -h, V_host[i], q));
      h_host[i] = accumulate ? h_host[i] + h : h;
    }
    if (nrm_host) KHIP_TRY(khip_nrm2(ctx, n, q, nrm_host));
    return KHIP_OK;
  }
  // chain through device-resident scalars, one host sync at the end.
  //   h_0 = V_0 . q ; then for i: q -= h_i V_i fused with h_{i+1} = V_{i+1} . q (or ||q||^2 at the end)
  // Multi-GPU: every coefficient is all-reduced ON THE DEVICE (ncclAllGather of the 16-byte partials + combine
  // kernel, comm_allreduce_dd_device) before the next step reads it -- still no host in the cascade.
  if (k == 0) {
    if (nrm_host) return khip_nrm2(ctx, n, q, nrm_host);
    return KHIP_OK;
  }
  const bool multi = comm_nranks(ctx) > 1;
  int slot = 0;
  KHIP_TRY(mgs_enqueue(ctx, n, k, V_host, q, &slot));
  std::vector<double> tmp((size_t)k + 1);
  KHIP_TRY(fetch_results(ctx, slot, k + 1, tmp.data(), /*already_global=*/multi));
  for (int i = 0; i < k; ++i) h_host[i] = accumulate ? h_host[i] + tmp[i] : tmp[i];
  if (nrm_host) *nrm_host = std::sqrt(tmp[k]);
  return KHIP_OK;
}

}  // extern "C"
int khip::mgs_enqueue(khip_ctx *ctx, int64_t n, int k, const double *const *V_host, double *q, int *slot_out) {
  const bool multi = comm_nranks(ctx) > 1;
  const int slot = take_slots(ctx, k + 1);
  KHIP_TRY(launch_dot(ctx, n, V_host[0], q, slot));
  if (multi) KHIP_TRY(comm_allreduce_dd_device(ctx, slot, 1));
  for (int i = 0; i < k; ++i) {
    const double *znext = (i + 1 < k) ? V_host[i + 1] : q;   // last step: ||q||^2
    KHIP_TRY(launch_axpy_dev_dot(ctx, n, ctx->results + slot + i, V_host[i], q, znext, slot + i + 1));
    if (multi) KHIP_TRY(comm_allreduce_dd_device(ctx, slot + i + 1, 1));
  }
  *slot_out = slot;
  return KHIP_OK;
}
extern "C" {

int khip_multi_axpy(khip_ctx *ctx, int64_t n, int k, const double *y_host, const double *const *V_host, double *x) {
  KHIP_REQUIRE(ctx && (k == 0 || (y_host && V_host)) && (n == 0 || x), "multi_axpy: null argument");
  return launch_multi_axpy(ctx, n, k, y_host, V_host, x);
}

}  // extern "C"
