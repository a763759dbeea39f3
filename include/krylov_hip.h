/*
 * krylov_hip.h -- C ABI of the MI355X-native Krylov inner-loop engine (libkrylov_hip.so).
 *
 * This is the drop-in boundary behind Krylov.jl's own extension contract
 * (docs/src/custom_workspaces.md:107-300): a Julia device
 * vector type forwards each `Krylov.k*` method to one entry point below with a
 * `ccall`; INTEGRATION.md shows that glue.  Plain pointers and sizes only: every
 * `double*` argument is a DEVICE pointer (gfx950 HBM) unless the name ends in
 * `_host`.  Element type: Float64 real (T = FC = Float64), int32 column indices.
 *
 * Conventions (mirror docs/src/interfaces/reference.md:144-165): every function
 * returns an int, 0 = success, < 0 = error; nothing is thrown across the ABI;
 * khip_last_error() holds the message of the last failure on this thread.
 * All calls on one khip_ctx must come from one host thread (the reference's C
 * library has the same rule, reference.md:172).  Kernels run on the context's HIP
 * stream; only the entry points that return a host scalar synchronise.
 *
 * "ref:" comments cite the reference interface each entry point replaces
 * (paths relative to the Krylov.jl v0.10.8 tree).
 */
#ifndef KRYLOV_HIP_H
#define KRYLOV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KHIP_OK               0
#define KHIP_ERR_INVALID     -1   /* bad argument / inconsistent sizes (Julia error(...)) */
#define KHIP_ERR_HIP         -2   /* a HIP runtime call failed */
#define KHIP_ERR_UNSUPPORTED -3
#define KHIP_ERR_COMM        -4   /* RCCL failure */
#define KHIP_ERR_NUMERIC     -5   /* e.g. operator not SPD (src/cg.jl:163,243) */

#define KHIP_VERSION_MAJOR 0
#define KHIP_VERSION_MINOR 4   /* 2: khip_options gained log_fd (round 4); 3: khip_*_workspace_adopt & co. (round 5); 4: khip_*_last_path,
                                * history published before every callback (round 6).  Clients check
                                * khip_version() against the header they were built with */

typedef struct khip_ctx khip_ctx;   /* device + stream + scratch + (optional) communicator */
typedef struct khip_csr khip_csr;   /* CSR operator resident in HBM */

const char *khip_last_error(void);
void        khip_version(int *major, int *minor);

/* ---------------------------------------------------------------- context ---- */
/* stream: a hipStream_t to borrow (e.g. the caller's current stream) or NULL to create one. */
int   khip_device_count(int *count);            /* HIP devices visible to this process (0 without a GPU; never fails) */
int   khip_device_pci_id(int device, char *buf, size_t cap);   /* "0000:05:00.0": the physical GPU behind a device index (cap >= 16) */
int   khip_ctx_create(int device, void *stream, khip_ctx **out);
int   khip_ctx_destroy(khip_ctx *ctx);
int   khip_ctx_sync(khip_ctx *ctx);
void *khip_ctx_stream(khip_ctx *ctx);
/* tuning / behaviour knobs, see DESIGN.md ("spmv_rows", "spmv_vec", "spmv_nt", "spmv_xcd",
 * "compensated", "blas1_blocks", ...).  Unknown key -> KHIP_ERR_INVALID. */
int   khip_ctx_set_option(khip_ctx *ctx, const char *key, int value);
int   khip_ctx_get_option(khip_ctx *ctx, const char *key, int *value);

/* ------------------------------------------------ device buffers ------------- */
/* ref: S(undef, n) / similar(x) in every workspace constructor
 *      (src/krylov_workspaces.jl:269-285,1605-1623,2898-2918; allocate_if src/krylov_utils.jl:281-299) */
int khip_malloc(khip_ctx *ctx, size_t bytes, void **dptr);
int khip_free(khip_ctx *ctx, void *dptr);
int khip_memcpy_h2d(khip_ctx *ctx, void *dst, const void *src_host, size_t bytes);
int khip_memcpy_d2h(khip_ctx *ctx, void *dst_host, const void *src, size_t bytes);
int khip_memcpy_d2d(khip_ctx *ctx, void *dst, const void *src, size_t bytes);
int khip_mem_info(khip_ctx *ctx, size_t *free_bytes, size_t *total_bytes);

/* ------------------------------------------------ CSR operator --------------- */
/* ref: the sparse matrix handed to kmul!(y, A, x) = mul!(y, A, x) (src/krylov_utils.jl:305);
 *      on the existing GPU path a ROCSparseMatrixCSR (docs/src/gpu.md:165-225).
 * rowptr has m+1 entries of `rowptr_bits` (32 or 64) bits, col/val have nnz entries;
 * index_base is 0 or 1 (Julia arrays are 1-based).  The arrays are COPIED (from host when
 * on_device == 0, from device otherwise); the handle owns its HBM copy.  nnz of one handle
 * must be < 2^31 (one GPU's shard; SURVEY.md section 7 "int32 limits"). */
int khip_csr_create(khip_ctx *ctx, int64_t m, int64_t n, int64_t nnz, const void *rowptr,
                    int rowptr_bits, const int32_t *col, const double *val, int index_base,
                    int on_device, khip_csr **out);
/* Distributed form (ref recipe docs/src/custom_workspaces.md:477-586):
 * this rank owns global rows [row0, row0+m) of an n_global-square operator; col[] holds
 * GLOBAL indices.  Requires khip_comm_init on ctx.  Builds the halo plan (needed remote
 * columns, send lists) and remaps columns to [owned | ghost] numbering. */
int khip_csr_create_dist(khip_ctx *ctx, int64_t n_global, int64_t row0, int64_t m, int64_t nnz,
                         const void *rowptr, int rowptr_bits, const int32_t *col,
                         const double *val, int index_base, int on_device, khip_csr **out);
/* The adjoint operator A' as its own handle (built on the device; entries of a row of A' in increasing column
 * order = the order of a column of A).  khip_spmv(At, x, y) is then `mul!(y, A', x)` (docs/src/matrix_free.md:36-42),
 * what MINRES-QLP / LSQR / LSMR / BiLQ / QMR ... ask of an operator besides `mul!(y, A, x)`.
 * A row-partitioned handle (khip_csr_create_dist) gives a row-partitioned A' with the SAME partition: every rank's entries
 * travel to the owners of their columns (one all-to-all at set-up, collective: call it on every rank), and y = A' x is
 * bit-identical to the single-GPU product.  ref: the two-sided processes and solvers, src/krylov_processes.jl:133-222. */
int khip_csr_transpose(khip_ctx *ctx, const khip_csr *A, khip_csr **At_out);
int khip_csr_destroy(khip_csr *A);
/* Optional internal re-encoding of a handle whose rows repeat few (column - row, value) sequences (stencils):
 * one 16-bit row-template id per row instead of 12 bytes per nonzero (csrc/template.hip).  *templates = number
 * of distinct templates, 0 when the operator is not compressible (it then stays plain CSR).  Results of
 * khip_spmv & co. are bit-identical either way; ctx option "spmv_template" = 0 ignores the compressed form. */
int khip_csr_compress(khip_ctx *ctx, khip_csr *A, int *templates);
/* bytes one SpMV of this handle moves in its CURRENT representation (khip_spmv_bytes: always the CSR formula) */
int khip_spmv_bytes_stored(const khip_csr *A, int64_t *bytes);
int khip_csr_shape(const khip_csr *A, int64_t *m, int64_t *n, int64_t *nnz);
/* How the staged SpMV reads the column indices of this handle (csrc/colcode.hip): *bits = 32 (the int32 CSR columns),
 * or 8 / 16 when the operator's entries lie on at most 256 / 2048 distinct diagonals (column - row) and the handle keeps,
 * next to its CSR arrays, one / two bytes per entry (the rank of the entry's diagonal in a sorted table; col = row +
 * table[code], exact).  The coded stream is built by the first khip_spmv that can use it; *diagonals = table size
 * (0 before that, or when the operator has too many).  y is bit-identical either way; ctx option "spmv_codes": 1 (default)
 * codes operators of at least 4 M entries (smaller ones are latency bound and keep the int32 stream), 2 codes whatever the
 * size, 16 forces two-byte codes, 0 keeps the int32 stream (environment variable KHIP_SPMV_CODES sets the initial value).  ref: the product is kmul!(y, A, x), src/krylov_utils.jl:305. */
int khip_csr_code_info(const khip_csr *A, int *bits, int *diagonals);
/* The SLICED form of a handle with 8-bit codes (csrc/colcode.hip csr_build_sell, read by spmv_sell_kernel): the entries of every 64
 * consecutive rows stored a second time transposed -- per slice, per lane, W = ceil(L / 8) words of eight codes then L values
 * (L = longest row of the slice; a unit = 64 words = 512 bytes) -- so that every lane loads the entries of its own row with
 * coalesced 8-byte loads: no LDS window and no barrier in the row walk.  Built by the first khip_spmv that can use it when the
 * padding costs at most 12 % (ctx option "spmv_sell": 2 default = non-temporal loads of the matrix words, 1 = default policy,
 * 0 = off: the coded CSR stream).  *state = 1 built, 0 not tried, -1 not usable; *units_per_slice > 0: every slice is padded to
 * that many units (no offset array), 0: per-slice offsets; *total_units: units stored.  y and the fused dots are bit-identical to
 * the coded CSR kernel's.  khip_spmv_bytes_stored counts 512 B per unit (+ 4 B per slice of offsets) while the form is in use. */
int khip_csr_sell_info(const khip_csr *A, int *state, int *units_per_slice, int64_t *total_units);
/* *narrow = 1 when the sliced form keeps 4-bit codes: operators with at most 15 diagonals and at most 8 entries per row store ONE
 * 32-bit word of eight codes per row (indexed by the row) and the slices hold values only -- 60 instead of 64 B per 7-point row
 * (ctx option "spmv_sell_narrow" = 1; default 0: it measured 8-10 % slower than the byte-coded words at 512^3, whose slices are 4 KB blocks). */
int khip_csr_sell_narrow(const khip_csr *A, int *narrow);
/* The same for the int32 column stream -- operators that are not coded (more than 2048 diagonals, fewer than 4 M entries under
 * "spmv_codes" = 1, or "spmv_codes" = 0) with short rows (the staged kernel's operators: at most 12 entries per row on average):
 * per slice, per lane, ceil(L / 2) words of two int32 columns (-1 = no entry) then L values; same padding rule, same size rule as
 * the codes (at least 4 M entries unless "spmv_codes" = 2 or "spmv_sell" >= 3), same option.  y and the fused dots are
 * bit-identical to the staged CSR kernel's. */
int khip_csr_sell32_info(const khip_csr *A, int *state, int *units_per_slice, int64_t *total_units);
/* Block-delta column stream of the stream SpMV (operators that are not stencils: more than 2048 diagonals; built at the first
 * product that can use it, ctx option "spmv_delta"): bits = 8 / 16 (32: not in use), rows = rows per block, escapes = entries
 * that stay int32 (6 B each).  khip_spmv_bytes_stored counts what that kernel streams. */
int khip_csr_delta_info(const khip_csr *A, int *bits, int *rows, int64_t *escapes);
/* Which SpMV kernel khip_spmv takes for this handle under the context's options: 1 = stream (products in nnz order through
 * LDS), 2 = strided vector (very long rows), 3 = ordered sub-wave, 4 = staged rows (one lane per row; also the one that streams
 * the dictionary codes), 5 = row templates (khip_csr_compress), 6 = wave-private windows (LDS-DMA). */
int khip_spmv_kernel_info(khip_ctx *ctx, const khip_csr *A, int *kernel);
/* Which SpMM kernel a product with 16 right-hand sides runs on this handle (csrc/spmm_tile.hip): *state = 1 when the handle
 * keeps group records for the wave-private-window kernel (built by the first khip_spmm with p = 16; 0 before that, -1 when
 * the operator lacks the locality and the other SpMM kernels are used); *window = distinct panel rows a group's LDS window
 * holds; *grid_tiles = 1 when the 32-row groups are 4 x 4 x 2 tiles of a structured grid recognised from the pattern, 0 when
 * they are 32 consecutive rows; *direct_groups of *groups go down the direct-gather path (a row longer than 32 entries, or
 * more distinct columns than the window takes); *reuse = entries per distinct column of a group.  Y is bit-identical in
 * every case; ctx option "spmm_tile" = 0 switches the kernel off.  ref: mul!(W, A, P), src/block_gmres.jl:242. */
int khip_csr_tile_info(const khip_csr *A, int *state, int *window, int *grid_tiles, int64_t *groups, int64_t *direct_groups,
                       double *reuse);
/* How a distributed handle fetches the remote part of x before a product (csrc/comm.cpp; the reference's MPI recipe,
 * docs/src/custom_workspaces.md:517-521 and :583-586): *gather_mode = 0: only the entries this rank's columns reference
 * travel (grouped Send/Recv with the owning ranks; *n_ghost entries received, *n_send sent per product); 1: every rank's
 * slice is all-gathered (*n_ghost = ranks x largest slice, *n_send = own slice).  Chosen at khip_csr_create_dist by the
 * ctx options "halo_mode" (0 = per operator: gather when some rank would fetch more than "halo_gather_pct" % of its own
 * row count, 1 = always the neighbour exchange, 2 = always the all-gather).  y is bit-identical in both modes. */
int khip_csr_halo_info(const khip_csr *A, int *gather_mode, int64_t *n_ghost, int64_t *n_send);
/* device-side generators of the benchmark operators (rows [row0, row0+m) of the global matrix,
 * global columns; ref: test/get_div_grad.jl:8-25, test/test_utils.jl:160-169).
 * kind: 0 = get_div_grad(n1,n2,n3) 7-pt Poisson, 1 = kron_unsymmetric(n1), 2 = 27-pt cfg-5 operator.
 * Outputs are device arrays owned by the caller (khip_free): rowptr int32[m+1] (local, 0-based),
 * col int32[nnz] (global, 0-based), val double[nnz]. */
int khip_gen_stencil(khip_ctx *ctx, int kind, int n1, int n2, int n3, int64_t row0, int64_t m,
                     int32_t **rowptr_dev, int32_t **col_dev, double **val_dev, int64_t *nnz);
/* Generator of the "banded + random, fixed seed" benchmark operator (csrc/gen_irregular.cpp; the stand-in for the
 * SuiteSparse matrices of the reference's benchmarks, benchmark/cg_bmark.jl:29-54, benchmark/gpu.jl:15-47, which cannot be
 * fetched offline): a symmetric band of half width half_band with 1 entry in 8 left out, `links` long-range partners per row
 * from seeded involutions of blocks of 2^20 rows (far more than 2048 distinct diagonals), off-diagonal entries in (-2, -1],
 * diagonal = 1/16 + the row's absolute sum (strictly diagonally dominant; SPD when symmetric).  flags bit 0: nonsymmetric values
 * (entries above the diagonal halved); dense_rows > 0 (nonsymmetric only): that many rows with 3000 further entries each.
 * Built on the host, rows [row0, row0 + m) of the global operator; outputs as khip_gen_stencil.  The definition is restated in
 * the file's header and, independently, by the oracle (ko_csr_banded_random). */
int khip_gen_banded_random(khip_ctx *ctx, int64_t n, int half_band, int links, uint64_t seed, int flags, int dense_rows,
                           int64_t row0, int64_t m, int32_t **rowptr_dev, int32_t **col_dev, double **val_dev, int64_t *nnz);

/* y <- A x.  ref: kmul!(y, A, x) src/krylov_utils.jl:305; sites src/cg.jl:155,196,
 * src/gmres.jl:159,222,257, src/bicgstab.jl:160,221,228.  For a distributed handle x and y are
 * the owned slices; the halo exchange happens inside. */
int khip_spmv(khip_ctx *ctx, const khip_csr *A, const double *x, double *y);
/* Y <- A X for row-major n-by-p panels (ld = p).  ref: mul!(W, A, P) src/block_gmres.jl:242.
 * The first product with an even p on an operator whose neighbouring rows share columns (banded / stencil) builds,
 * once per handle and range of p, a list of the distinct columns of every group of 256 / ceil_pow2(p / 2) rows and a
 * 16-bit slot per nonzero (~2 B per nonzero + 4 B per list entry of device memory, about the time of five products);
 * later products copy each distinct panel row once into LDS instead of gathering it per nonzero.  Y is bit-identical
 * either way (ctx option "spmm_window" = 0 keeps the direct gathers).  Like every call on a handle, this is confined
 * to the context's thread. */
int khip_spmm(khip_ctx *ctx, const khip_csr *A, const double *X, double *Y, int p);
/* algorithmic HBM bytes of one khip_spmv on this handle: 12 nnz + 4 (m+1) + 8 n + 8 m */
int khip_spmv_bytes(const khip_csr *A, int64_t *bytes);

/* Profiling hook used by bench.py's roofline leg: with khip_ctx_set_option(ctx, "profile_spmv", 1)
 * every SpMV kernel launch is bracketed by HIP events on the context's stream; this call
 * synchronises, returns the number of launches recorded since the last call and their summed
 * duration, and resets the counters. */
int khip_profile_spmv(khip_ctx *ctx, int64_t *launches, double *total_ms);
/* The same brackets by kernel family (what bench.py's cfg-3 / cfg-5 legs report): launches[t] and total_ms[t] for t < ntags of
 * 0 = SpMV, 1 = SpMM (khip_spmm: the main kernel of a launch), 2 = panel_gemm_tn (V' Q), 3 = panel_nn_tn (the fused block
 * Gram-Schmidt step Q -= V Psi ; Psi' = V' Q), 4 = panel_multi_nn (X += sum V_i Y_i), 5 = panel_gemm_nn, 6 = the panel QR's passes
 * (scale + Gram), 7 = halo pack kernel, 8 = halo transfer (grouped ncclSend / ncclRecv or the all-gather of x, bracketed on the
 * stream it runs on), 9 = a dot's 16-byte all-gather + combine kernel, 10 = the boundary rows' SpMV launch of a row-partitioned
 * product (tag 0 then is the interior launch); at most 11 families.  Synchronises the context's streams and resets every counter. */
int khip_profile_kernels(khip_ctx *ctx, int ntags, int64_t *launches, double *total_ms);

/* ------------------------------------------------ BLAS-1 shim ---------------- */
/* ref: src/krylov_utils.jl:309-349.  x and y may alias exactly (BLAS semantics). */
int khip_dot(khip_ctx *ctx, int64_t n, const double *x, const double *y, double *result_host);      /* kdot  :309-311 (kdotr :313-314) */
int khip_nrm2(khip_ctx *ctx, int64_t n, const double *x, double *result_host);                      /* knorm :316-317 */
int khip_scal(khip_ctx *ctx, int64_t n, double s, double *x);                                       /* kscal! :321-323 */
int khip_div(khip_ctx *ctx, int64_t n, double *x, double s);                                        /* kdiv!  :325-326 (= scal by 1/s) */
int khip_copy(khip_ctx *ctx, int64_t n, double *y, const double *x);                                /* kcopy! :328-329, (dest, src) */
int khip_scalcopy(khip_ctx *ctx, int64_t n, double *y, double s, const double *x);                  /* kscalcopy! :331-332 */
int khip_divcopy(khip_ctx *ctx, int64_t n, double *y, const double *x, double s);                   /* kdivcopy!  :334-335 */
int khip_axpy(khip_ctx *ctx, int64_t n, double s, const double *x, double *y);                      /* kaxpy!  :337-339 */
int khip_axpby(khip_ctx *ctx, int64_t n, double s, const double *x, double t, double *y);           /* kaxpby! :341-345 */
int khip_fill(khip_ctx *ctx, int64_t n, double *x, double val);                                     /* kfill!  :347 */
int khip_ref(khip_ctx *ctx, int64_t n, double *x, double *y, double c, double s);                   /* kref!   :349 */

/* elementwise products for diagonal operators / Jacobi preconditioning (SURVEY.md section 8f N1;
 * ref: M = Diagonal(1 ./ diag(A)) applied through mulorldiv!, src/cg.jl:160,241, test/test_gmres.jl:105-128) */
int khip_vmul(khip_ctx *ctx, int64_t n, double *w, const double *x, const double *y);   /* w = x .* y */
int khip_vdiv(khip_ctx *ctx, int64_t n, double *w, const double *x, const double *y);   /* w = x ./ y */
int khip_csr_diagonal(khip_ctx *ctx, const khip_csr *A, double *diag);                  /* diag[i] = A[i,i] */

/* ------------------------------------------------ fused accelerators --------- */
/* Each equals its unfused sequence on the device up to the last bit of the reduction
 * (elementwise results are bit-identical; reductions agree to <= 1 ulp). */
/* y <- A x ; *result_host = x . y           (src/cg.jl:196-197) */
int khip_spmv_dot(khip_ctx *ctx, const khip_csr *A, const double *x, double *y, double *result_host);
/* x <- x + a p ; r <- r - a q ; *result_host = r . r     (src/cg.jl:239-242 with M = I) */
int khip_axpy2_dot(khip_ctx *ctx, int64_t n, double a, const double *p, const double *q,
                   double *x, double *r, double *result_host);
/* y <- y + a x ; *result_host = y . y                      (src/cg.jl:240,242 with M = I: r <- r - alpha Ap ; r.r) */
int khip_axpy_sqnorm(khip_ctx *ctx, int64_t n, double a, const double *x, double *y, double *result_host);
/* The set-up of cg! with M = I and no warm start (src/cg.jl:153,158,161,162: kfill!(x, 0); kcopy!(r, b); kcopy!(p, r); gamma =
 * kdotr(r, r)) in ONE pass: 8n bytes read, 24n written instead of 48n; x, r, p and gamma hold what the four primitives leave (the
 * all-reduce of gamma included on row-partitioned vectors).  The four vectors must be distinct. */
int khip_cg_setup(khip_ctx *ctx, int64_t n, const double *b, double *x, double *r, double *p, double *result_host);
/* x <- x + a p ; p <- r + b p in one pass over p          (src/cg.jl:239 and :259 = kaxpy!(n, a, p, x) ; kaxpby!(n, 1, r, b, p)).
 * Legal because x is not read between the two reference lines; 40n bytes instead of 48n. */
int khip_cg_update(khip_ctx *ctx, int64_t n, double a, double b, const double *r, double *p, double *x);
/* y <- A x ; *result_host = w . y               (src/bicgstab.jl:221-223: v = A p ; c . v in one pass) */
int khip_spmv_dotw(khip_ctx *ctx, const khip_csr *A, const double *x, double *y, const double *w, double *result_host);
/* y <- A x ; result_host[0] = x . y ; result_host[1] = y . y        (src/bicgstab.jl:228-230: t = A s ; t.s ; t.t) */
int khip_spmv_dot2(khip_ctx *ctx, const khip_csr *A, const double *x, double *y, double *result_host);
/* s <- r - alpha v ; x <- x + alpha y                                (src/bicgstab.jl:224-226) */
int khip_bicgstab_sx(khip_ctx *ctx, int64_t n, double alpha, const double *r, const double *v, const double *y,
                     double *s, double *x);
/* x <- x + omega z ; r <- s - omega t ; result_host = {c . r, r . r}  (src/bicgstab.jl:231-234,240; z may alias s) */
int khip_bicgstab_xr(khip_ctx *ctx, int64_t n, double omega, const double *s, const double *t, const double *z,
                     const double *c, double *x, double *r, double *result_host);
/* p <- r + beta (p - omega v)                                         (src/bicgstab.jl:236-237) */
int khip_bicgstab_p(khip_ctx *ctx, int64_t n, double omega, double beta, const double *v, const double *r, double *p);
/* w <- x + b y  written to w (w may alias x or y): copy + axpy in one pass (src/bicgstab.jl:224-225,232-233) */
int khip_waxpy(khip_ctx *ctx, int64_t n, double *w, const double *x, double b, const double *y);
/* result_host[0] = x . y ; result_host[1] = x . x      (src/bicgstab.jl:230) */
int khip_dot2(khip_ctx *ctx, int64_t n, const double *x, const double *y, double *result_host);
/* Modified Gram-Schmidt cascade against k basis vectors V[0..k) (device pointers in a HOST array),
 * in the reference's order (src/gmres.jl:259-262): for i: h_i = V_i . q ; q <- q - h_i V_i.
 * Runs k+1 dependent kernels with device-resident scalars and ONE host sync; h_host gets the k
 * coefficients, *nrm_host = ||q|| afterwards (src/gmres.jl:274) when nrm_host != NULL.
 * accumulate != 0 adds the coefficients into h_host instead (reorthogonalisation pass :265-271). */
int khip_mgs(khip_ctx *ctx, int64_t n, int k, const double *const *V_host, double *q,
             double *h_host, double *nrm_host, int accumulate);
/* x <- x + sum_i y_i V_i, applied per element in the order i = 0..k-1 (bit-identical to k
 * kaxpy! calls, src/gmres.jl:348-350) */
int khip_multi_axpy(khip_ctx *ctx, int64_t n, int k, const double *y_host,
                    const double *const *V_host, double *x);

/* ------------------------------------------------ panel (block-GMRES) -------- */
/* Panels are n-by-p ROW-MAJOR in HBM (p contiguous; DESIGN.md "panel layout").
 * ref: mul!(R, V', Q) / mul!(Q, V, R, -1, 1) src/block_gmres.jl:244-247, householder!
 * src/block_krylov_utils.jl:201-208, X += V*Y src/block_gmres.jl:324-326. */
/* a panel holds n_pad = n rounded up to 16 rows; the padding rows must be (and stay) zero */
int khip_panel_rows(int64_t n, int64_t *n_pad);
int khip_panel_from_colmajor(khip_ctx *ctx, int64_t n, int p, const double *X_colmajor, double *P);
int khip_panel_to_colmajor(khip_ctx *ctx, int64_t n, int p, const double *P, double *X_colmajor);
/* Psi_host (p-by-p, column-major, HOST) <- V^T Q */
int khip_panel_gemm_tn(khip_ctx *ctx, int64_t n, int p, const double *V, const double *Q, double *Psi_host);
/* Q <- beta*Q + alpha * V * Psi  (Psi p-by-p column-major HOST) */
int khip_panel_gemm_nn(khip_ctx *ctx, int64_t n, int p, double alpha, const double *V,
                       const double *Psi_host, double beta, double *Q);
/* Block Gram-Schmidt sweep of Q against the k panels V[0..k) (device pointers in a HOST array) in the reference's
 * order (src/block_gmres.jl:244-247): for i: Psi_i = V_i^T Q ; Q <- Q - V_i Psi_i.  Psi_host receives the k blocks
 * (p x p column-major each, block i at Psi_host + i p p); accumulate != 0 adds them instead (the reorthogonalisation
 * pass :250-256).  On one GPU Psi_{i+1} is formed by the kernel that applies Psi_i (four panel passes per step
 * instead of five, one host synchronisation per sweep); same bits as the khip_panel_gemm_tn / _nn sequence. */
int khip_panel_mgs(khip_ctx *ctx, int64_t n, int p, int k, const double *const *V_host, double *Q, double *Psi_host,
                   int accumulate);
/* X <- beta X + sum_{i < k} V_i Y_i, the products applied in the order i = 0 .. k-1: the k calls mul!(Xr, V[i], Y[i], 1, 1) of
 * the solution update src/block_gmres.jl:324-326 in one pass over X (k + 2 panel passes instead of 3 k).  V_host: k device
 * panel pointers in a HOST array; Y_host: k blocks of p x p, column-major, HOST.  Bit-identical to the k khip_panel_gemm_nn
 * calls.  ctx option "panel_multi_tiles" (default 2; 0 = one tile per wave, factors re-read per tile). */
int khip_panel_multi_nn(khip_ctx *ctx, int64_t n, int p, int k, const double *const *V_host, const double *Y_host, double beta,
                        double *X);
/* Reduced QR of the panel as householder!(Q, R, tau) = kgeqrf! + korgqr! leaves it (src/block_krylov_utils.jl:201-208,
 * :230-236, :254-262): Q overwritten by the orthonormal factor, R_host (p-by-p column-major, upper triangular) and
 * tau_host (p, may be null) with LAPACK's sign convention (R_jj = -sign(alpha_j) |x_j|, tau_j in [1, 2]).  Computed on the
 * device by CholeskyQR2 (shifted CholeskyQR3 for ill-conditioned blocks); the signs and tau come from the top p-by-p block
 * of Q (csrc/block.cpp).  ctx option "panel_signs" = 0 leaves the positive diagonal of the Cholesky factor.  A block without
 * full column rank (equal, dependent or zero columns) is factored as LAPACK factors it: A = Q R with orthonormal Q, zeros on
 * R's diagonal where a column lies in the span of the columns before it and an arbitrary unit vector in Q there; only a
 * block that is zero or not finite altogether is KHIP_ERR_NUMERIC. */
int khip_panel_qr(khip_ctx *ctx, int64_t n, int p, double *Q, double *R_host);
int khip_panel_qr_tau(khip_ctx *ctx, int64_t n, int p, double *Q, double *R_host, double *tau_host);
int khip_panel_norm(khip_ctx *ctx, int64_t n, int p, const double *Q, double *result_host);

/* ------------------------------------------------ multi-GPU ------------------- */
/* One process per GPU.  unique_id is the 128-byte ncclUniqueId created by
 * khip_comm_unique_id on rank 0 and broadcast by the launcher (torch.distributed / MPI). */
int khip_comm_unique_id(void *id128_host);
int khip_comm_init(khip_ctx *ctx, int rank, int nranks, const void *id128_host);
/* In-process backend: all `nranks` ranks are contexts of THIS process (one host thread each; any mix of
 * devices, including several ranks on one GPU), exchanging by device-to-device copies.  Same semantics as
 * the RCCL backend for everything built on top; used to test the distributed path on a single GPU. */
int khip_comm_init_local(khip_ctx *ctx, int rank, int nranks, int hub_id);
int khip_comm_rank(khip_ctx *ctx, int *rank, int *nranks);
/* What the communicator attached to ctx looks like: *rccl_ranks = ncclCommCount of the RCCL communicator (0 for the
 * in-process backend or without a communicator), *local_backend = 1 for khip_comm_init_local, *halo_comm_separate = 1
 * when the halo exchange runs on its own communicator split off the first (csrc/comm.cpp: dot all-gathers on the
 * context's stream and communicator, halo Send/Recv on the second stream and communicator).  Any pointer may be null.
 * ref: the reference's MPI recipe docs/src/custom_workspaces.md:477-586. */
int khip_comm_info(khip_ctx *ctx, int *rank, int *nranks, int *rccl_ranks, int *local_backend, int *halo_comm_separate);
int khip_comm_barrier(khip_ctx *ctx);

/* Host-only helpers of the partition / halo logic (no GPU needed; used by khip_csr_create_dist and
 * by the world_size-2 gloo tests).  ghost_columns: sorted unique columns of rows [row0, row0+m)
 * that fall outside [row0, row0+m) (rowptr is int64, local, 0-based; col global).  halo_plan: given
 * EVERY rank's sorted ghost list (concatenated, offsets ghost_off[nranks+1]) and the partition
 * row_starts[nranks+1], derive for `rank` the per-peer receive segments of its own ghost list
 * (recv_off) and the owned indices it must pack for each peer (send_idx, segments send_off). */
int khip_ghost_columns_host(const int64_t *rowptr, const int32_t *col, int64_t m, int64_t row0,
                            int32_t *out, int64_t cap, int64_t *count);
int khip_halo_plan_host(int rank, int nranks, const int64_t *row_starts, const int32_t *ghost_all,
                        const int64_t *ghost_off, int64_t *recv_off, int64_t *send_off,
                        int32_t *send_idx, int64_t send_cap);

/* ------------------------------------------------ solvers --------------------- */
/* Host control flow of the reference's in-place methods, restated in C++ above the primitives
 * (Julia is absent in this build environment; with Julia the unmodified src/cg.jl etc. drive the
 * same primitives through the k* methods).  Operators: a khip_csr, or a user callback on device
 * pointers (the mul!-based operator contract, docs/src/matrix_free.md:32-34). */
typedef int (*khip_apply_fn)(void *self, const double *x, double *y);   /* y <- Op x, 0 on success */
typedef struct {
  const khip_csr *csr;       /* used when apply == NULL */
  khip_apply_fn   apply;     /* user operator on device pointers */
  void           *self;
} khip_operator;

/* built-in Jacobi preconditioner z <- r ./ diag(A) as an operator for the M / N arguments of the solvers */
int khip_jacobi_create(khip_ctx *ctx, const khip_csr *A, khip_operator *op_out);
int khip_jacobi_destroy(khip_operator *op);

/* ILU(0) of A on its own pattern as an operator y <- U^{-1} L^{-1} x for the M / N arguments; for SPD A this is
 * the IC(0) preconditioner (U = D L^T).  Replaces the vendor ic02 / ilu02 + triangular ldiv! of the
 * reference's GPU recipes (docs/src/gpu.md:74-163, test/gpu/nvidia.jl:37-100).  A must outlive the operator,
 * its column indices must be sorted within rows, every row needs a diagonal entry (KHIP_ERR_NUMERIC
 * otherwise, as for a zero pivot).  Distributed handle: block-Jacobi ILU(0) of the owned diagonal block. */
int khip_ilu0_create(khip_ctx *ctx, const khip_csr *A, khip_operator *op_out);
int khip_ilu0_destroy(khip_operator *op);
/* introspection: number of level-scheduling levels of the two triangular solves; device pointer to the
 * factor values (strict lower part = L without its unit diagonal, upper part incl. diagonal = U) on A's pattern */
int khip_ilu0_info(const khip_operator *op, int64_t *levels_lower, int64_t *levels_upper, const double **lu_dev);
/* replay the 2 x levels launches of one application from a cached hipGraph (default 1) or enqueue them one by one (0) */
int khip_ilu0_set_graph(khip_operator *op, int enable);
/* How the triangular solves are scheduled: dims3 = the grid n1 x n2 x n3 the pattern was recognised as (natural ordering,
 * dependencies towards smaller coordinates; the solves then run block by block, one persistent launch per triangle) or
 * 0, 0, 0 (level scheduling, one launch per level); blocks per triangle; failed = 1 if a bounded wait of the block schedule
 * ever gave up (never expected).  Synchronises the context's stream.  Option "ilu_blocks" = 0 at create keeps level scheduling,
 * = 2 runs the block schedule on packed entry lists only (no row records; for tests), = 3 takes the blocks from the level-sorted
 * row sequence even where a grid is recognised (what patterns without a grid get: dims3 = 0, 0, 0 but blocks > 0). */
int khip_ilu0_block_info(const khip_operator *op, int64_t *dims3, int64_t *blocks, int *failed);
typedef int (*khip_callback_fn)(void *workspace, void *userdata);       /* callback(workspace)::Bool */

typedef struct {
  double atol, rtol;          /* NaN -> sqrt(eps)                 (src/cg.jl:104-105) */
  int    itmax;               /* 0 -> 2n (2*div(n,p) for block)   (src/cg.jl:177) */
  double timemax;             /* NaN or <= 0 -> Inf */
  int    history;             /* push residual norms             (src/cg.jl:165,245) */
  double radius;              /* cg: trust-region radius          (src/cg.jl:215-237) */
  int    linesearch;          /* cg                               (src/cg.jl:198-211) */
  int    restart;             /* gmres / block_gmres              (src/gmres.jl:219-226) */
  int    reorthogonalization; /* gmres / block_gmres              (src/gmres.jl:265-271) */
  int    fused;               /* 0 = issue primitives exactly as the reference does; 1 = fused kernels;
                               * 2 = fused kernels + scalar recurrences and stopping tests on the device (cg, bicgstab: no
                               * host round trip inside the loop; gmres: one-step look-ahead behind the single sync per inner
                               * iteration); bit-identical to 1; falls back to 1 where it does not apply */
  khip_callback_fn callback; void *callback_data;
  int    variant;             /* 0 = the reference's recurrences.  cg: 1 = single-reduction CG (Chronopoulos & Gear 1989): the two
                               * dots of an iteration are computed by ONE reduction (one all-reduce per iteration on N GPUs,
                               * 2 passes per iteration).  Different rounding: same solution to the requested tolerance, iteration
                               * counts within a few of the reference recurrence -- opt-in, own parity budget.
                               * Needs M = I, a CSR operator, no trust region / linesearch / callback (else KHIP_ERR_UNSUPPORTED).
                               * cg: 2 = pipelined CG (Ghysels & Vanroose 2014): r, w = A r and their images are recurred, the one
                               * reduction (r.w, r.r) of an iteration does not feed the product q = A w that runs beside it; on N GPUs
                               * its all-gather, the cross-rank combine and the scalar update run on the communication stream while the
                               * product runs (needs a second communicator, else in order).  104n + SpMV bytes per iteration and two more
                               * work vectors.  Opt-in, own parity budget; same requirements as variant 1.
                               * gmres: 1 = CGS2, classical Gram-Schmidt applied twice instead of the modified Gram-Schmidt cascade of
                               * src/gmres.jl:259-271: h = V_k' q as one reduction per four basis vectors, q -= V_k h in one pass, twice:
                               * three all-reduces per inner iteration on N GPUs instead of k + 1.  Opt-in, own parity budget.
                               * gmres: 2 = s-step GMRES (monomial basis, s = ctx option "gmres_sstep", 1..8, default 4): s products, one batched
                               * CGS2 against the basis and a CholeskyQR2 of the block per s inner iterations -- four reductions per s
                               * iterations; Hessenberg columns, Givens rotations and the stopping test on the host, column by column.
                               * Needs restart = true, a CSR operator, M = N = I, memory x s <= 256.  Opt-in, own parity budget. */
  int    verbose;             /* > 0: the reference's log on stdout, one row every `verbose` iterations (src/cg.jl:132,182-183,224,
                               * 267-269; src/gmres.jl:131,191-192,315,364; src/bicgstab.jl:135,193-194,255-257;
                               * src/block_gmres.jl:120,181-182,297,340; kdisplay: src/krylov_utils.jl:301).  The rows need the scalars
                               * on the host, so a verbose solve runs the host-driven loop (as fused <= 1, and with the reference's
                               * recurrence: variant != 0 has no such rows and is refused with verbose > 0); results are the same bits */
  int    log_fd;              /* where the verbose log goes -- the reference's `iostream` keyword (src/cg.jl:24,182-183; default
                               * stdout): 0 = stdout, otherwise an open file descriptor of the caller (Julia: `fd(io)`), written with
                               * dprintf, never closed by the library */
} khip_options;

typedef struct {              /* SimpleStats, src/krylov_stats.jl:24-44 */
  int    niter, solved, inconsistent, indefinite, npcCount;
  double timer;
  char   status[96];
  const double *residuals; int nres;
  char   error[160];
  double allocation_timer;    /* seconds spent allocating the workspace's vectors: at its creation plus the lazy allocations of
                               * later solves (allocate_if, src/krylov_utils.jl:281-288; stats.allocation_timer, src/krylov_stats.jl) */
} khip_stats;

khip_options khip_default_options(void);

typedef struct khip_cg_workspace          khip_cg_workspace;           /* CgWorkspace       src/krylov_workspaces.jl:236-291 */
typedef struct khip_gmres_workspace       khip_gmres_workspace;        /* GmresWorkspace    :2857-2924 */
typedef struct khip_bicgstab_workspace    khip_bicgstab_workspace;     /* BicgstabWorkspace :1568-1629 */
typedef struct khip_block_gmres_workspace khip_block_gmres_workspace;  /* BlockGmresWorkspace src/block_krylov_workspaces.jl:115-171 */

/* ---- workspaces on CALLER-OWNED vectors ("adopt") -------------------------------------------------------------------
 * The reference's workspaces own their vectors on the Julia side (`x, r, p, Ap :: S`, src/krylov_workspaces.jl:236-291) and
 * `solution(ws) === ws.x` (test/test_interface.jl:260).  A binding that specialises `cg!(ws::CgWorkspace{..,HIPVector}, A, b)`
 * therefore hands the device pointers of THOSE vectors to khip_*_workspace_adopt once, and then runs khip_*_solve -- the fused,
 * device-resident loops -- directly on them: the solution lands in the caller's x, nothing is copied, nothing the caller
 * owns is ever freed or replaced by the library.  Vectors the reference allocates lazily (z, Δx, npc_dir, p, q, yz, t;
 * allocate_if, src/krylov_utils.jl:281-299) are handed over with khip_*_workspace_adopt_vector once the caller has
 * allocated them (ptr == NULL empties the slot again); a solve that needs one the caller never handed over allocates and
 * owns it.  Histories and stats are the same bits as on a khip_*_workspace_create workspace.  After a solve only x and the
 * stats are defined: the other work vectors hold scratch (the reference promises nothing about them either), and the
 * restart-time zero fill of the basis (src/gmres.jl:211-213, src/block_gmres.jl:195-197) is not performed.
 * khip_grow_fn: restart = false lets the basis outgrow `memory`; the library then asks the caller for one more vector
 * (`push!(V, similar(x))`, src/gmres.jl:319-324; a ZEROED panel for block-GMRES, src/block_gmres.jl:300-305) -- return its
 * device pointer, or NULL to fail the solve.  Without a grow callback the library allocates (and owns) the extra vectors. */
typedef double *(*khip_grow_fn)(void *userdata);

int khip_cg_workspace_create(khip_ctx *ctx, int64_t m, int64_t n, khip_cg_workspace **out);
/* CgWorkspace on the caller's x, r, p, Ap (distinct device vectors of n doubles); adopt_vector names: "x", "r", "p", "Ap",
 * "z", "dx" (the reference's Δx: hand it over, then khip_cg_warm_start(ws, dx) only sets the flag), "npc_dir" */
int khip_cg_workspace_adopt(khip_ctx *ctx, int64_t m, int64_t n, double *x, double *r, double *p, double *Ap,
                            khip_cg_workspace **out);
int khip_cg_workspace_adopt_vector(khip_cg_workspace *ws, const char *name, double *ptr);
int khip_cg_workspace_destroy(khip_cg_workspace *ws);
int khip_cg_warm_start(khip_cg_workspace *ws, const double *x0);                   /* warm_start! src/workspace_accessors.jl:193-200 */
/* cg!(ws, A, b; M, ...)  src/cg.jl:120-291.  M == NULL means M = I; otherwise z <- M r. */
int khip_cg_solve(khip_cg_workspace *ws, const khip_operator *A, const khip_operator *M,
                  const double *b, const khip_options *opts);
double           *khip_cg_solution(khip_cg_workspace *ws);                         /* solution(ws) === ws.x */
const khip_stats *khip_cg_stats(khip_cg_workspace *ws);
/* Which loop the workspace's last solve ran: 2 = the device-resident loop (cg!, bicgstab!) / the look-ahead loop (gmres!) --
 * what the default keywords of the reference's entry points get (src/interface.jl:146-154 forwards M = I, ldiv = false,
 * callback = workspace -> false, verbose = 0: the binding maps that default callback to NULL); 1 = the host-driven loop on the
 * fused kernels (a callback, verbose > 0, a preconditioner, a user operator; same bits); 0 = one launch per primitive
 * (options.fused = 0); -1 = no solve yet.  block_gmres!: 1 (it has one loop).  A binding's test asserts on this that an entry
 * point reached the loop it was meant to reach. */
int khip_cg_last_path(khip_cg_workspace *ws);
/* named work vectors for callbacks: "x","r","p","Ap","z","dx","npc_dir" */
double           *khip_cg_vector(khip_cg_workspace *ws, const char *name);
size_t            khip_cg_workspace_bytes(khip_cg_workspace *ws);                  /* storage test, test/test_allocations.jl:41-57 */

int khip_gmres_workspace_create(khip_ctx *ctx, int64_t m, int64_t n, int memory, khip_gmres_workspace **out);
/* GmresWorkspace on the caller's x, w and basis V_host[0 .. memory) (device pointers in a HOST array; `V::Vector{S}`,
 * src/krylov_workspaces.jl:2857-2873).  adopt_vector names: "x", "w", "p", "q", "dx".  adopt_basis replaces the whole list
 * (k >= memory: the caller's V after restart = false grew it).  khip_gmres_host_state copies c, s, z (cap entries each at
 * most) and the packed R (cap (cap + 1) / 2) of the last solve out, *len = their current length, *inner_iter as the
 * reference's field; any output may be NULL. */
int khip_gmres_workspace_adopt(khip_ctx *ctx, int64_t m, int64_t n, int memory, double *x, double *w,
                               double *const *V_host, khip_gmres_workspace **out);
int khip_gmres_workspace_adopt_vector(khip_gmres_workspace *ws, const char *name, double *ptr);
int khip_gmres_workspace_adopt_basis(khip_gmres_workspace *ws, int k, double *const *V_host);
int khip_gmres_workspace_set_grow(khip_gmres_workspace *ws, khip_grow_fn grow, void *userdata);
int khip_gmres_host_state(khip_gmres_workspace *ws, int cap, double *c_host, double *s_host, double *z_host, double *R_host,
                          int *len, int *inner_iter);
int khip_gmres_workspace_destroy(khip_gmres_workspace *ws);
int khip_gmres_warm_start(khip_gmres_workspace *ws, const double *x0);
/* gmres!(ws, A, b; M, N, restart, reorthogonalization, ...)  src/gmres.jl:121-384 */
int khip_gmres_solve(khip_gmres_workspace *ws, const khip_operator *A, const khip_operator *M,
                     const khip_operator *N, const double *b, const khip_options *opts);
double           *khip_gmres_solution(khip_gmres_workspace *ws);
const khip_stats *khip_gmres_stats(khip_gmres_workspace *ws);
int khip_gmres_last_path(khip_gmres_workspace *ws);
size_t            khip_gmres_workspace_bytes(khip_gmres_workspace *ws);

int khip_bicgstab_workspace_create(khip_ctx *ctx, int64_t m, int64_t n, khip_bicgstab_workspace **out);
/* BicgstabWorkspace on the caller's x, r, p, v, s, qd (src/krylov_workspaces.jl:1568-1582); adopt_vector names: those six and
 * "yz", "t", "dx" */
int khip_bicgstab_workspace_adopt(khip_ctx *ctx, int64_t m, int64_t n, double *x, double *r, double *p, double *v, double *s,
                                  double *qd, khip_bicgstab_workspace **out);
int khip_bicgstab_workspace_adopt_vector(khip_bicgstab_workspace *ws, const char *name, double *ptr);
int khip_bicgstab_workspace_destroy(khip_bicgstab_workspace *ws);
int khip_bicgstab_warm_start(khip_bicgstab_workspace *ws, const double *x0);
/* bicgstab!(ws, A, b; c, M, N, ...)  src/bicgstab.jl:125-277 ; c == NULL -> c = b */
int khip_bicgstab_solve(khip_bicgstab_workspace *ws, const khip_operator *A, const khip_operator *M,
                        const khip_operator *N, const double *b, const double *c,
                        const khip_options *opts);
double           *khip_bicgstab_solution(khip_bicgstab_workspace *ws);
const khip_stats *khip_bicgstab_stats(khip_bicgstab_workspace *ws);
int khip_bicgstab_last_path(khip_bicgstab_workspace *ws);
size_t            khip_bicgstab_workspace_bytes(khip_bicgstab_workspace *ws);

int khip_block_gmres_workspace_create(khip_ctx *ctx, int64_t m, int64_t n, int p, int memory,
                                      khip_block_gmres_workspace **out);
/* BlockGmresWorkspace on the caller's panels X, W, V_host[0 .. memory): row-major panels of khip_panel_rows(n) x p doubles
 * with zero padding rows (the tall blocks of a BlockGmresWorkspace{..,HIPMatrix}, src/block_krylov_workspaces.jl:115-163;
 * the small blocks stay inside the library).  adopt_panel names: "X", "W", "P", "Q", "dX". */
int khip_block_gmres_workspace_adopt(khip_ctx *ctx, int64_t m, int64_t n, int p, int memory, double *X, double *W,
                                     double *const *V_host, khip_block_gmres_workspace **out);
int khip_block_gmres_workspace_adopt_panel(khip_block_gmres_workspace *ws, const char *name, double *ptr);
int khip_block_gmres_workspace_adopt_basis(khip_block_gmres_workspace *ws, int k, double *const *V_host);
int khip_block_gmres_workspace_set_grow(khip_block_gmres_workspace *ws, khip_grow_fn grow, void *userdata);
int khip_block_gmres_workspace_destroy(khip_block_gmres_workspace *ws);
int khip_block_gmres_warm_start(khip_block_gmres_workspace *ws, const double *X0_colmajor);
int khip_block_gmres_warm_start_panel(khip_block_gmres_workspace *ws, const double *X0_panel);   /* X0 already a row-major panel */
/* block_gmres!(ws, A, B; restart, reorthogonalization, ...) src/block_gmres.jl:110-358.
 * B and the solution are n-by-p COLUMN-MAJOR device arrays (the reference layout); the
 * workspace converts to row-major panels internally.  A: CSR handle (SpMM kernel) or an apply
 * callback; M / N: NULL (= I) or apply callbacks.  Callbacks of the block solver receive ROW-MAJOR
 * device panels of khip_panel_rows(n) x p doubles (padding rows zero, and they must stay zero). */
int khip_block_gmres_solve(khip_block_gmres_workspace *ws, const khip_operator *A, const khip_operator *M,
                           const khip_operator *N, const double *B_colmajor, const khip_options *opts);
/* the same with B a row-major panel (khip_panel_rows(n) x p, zero padding rows) read in place; the solution is the X panel */
int khip_block_gmres_solve_panel(khip_block_gmres_workspace *ws, const khip_operator *A, const khip_operator *M,
                                 const khip_operator *N, const double *B_panel, const khip_options *opts);
int khip_block_gmres_get_X(khip_block_gmres_workspace *ws, double *X_colmajor);
const khip_stats *khip_block_gmres_stats(khip_block_gmres_workspace *ws);
int khip_block_gmres_last_path(khip_block_gmres_workspace *ws);
/* storage test (test/test_allocations.jl:734-761): bytes of the workspace with n x p blocks at their logical size;
 * *extra_bytes (may be null) = what this implementation holds beyond the reference's formula (the row-major panel copy
 * of B and the p x p staging blocks of the fused sweeps) */
size_t            khip_block_gmres_workspace_bytes(khip_block_gmres_workspace *ws, size_t *extra_bytes);

/* ---- Krylov processes (src/krylov_processes.jl) --------------------------------------------------------------
 * The bases are dense COLUMN-MAJOR device arrays as in the reference (`M(undef, n, k+1)`): column j starts at
 * V + j*ldv, ldv >= n and even, V 16-byte aligned.  The small matrices come back on the HOST in the reference's
 * own storage.  An exact breakdown returns KHIP_ERR_NUMERIC with the reference's message in khip_last_error()
 * unless allow_breakdown != 0 (the new basis vector is then zero-filled, as the reference does).
 * Row-partitioned operators: n (and m) are the local row counts; dots and norms are all-reduced inside. */
/* V, beta1, T = hermitian_lanczos(A, b, k; allow_breakdown, reorthogonalization)     src/krylov_processes.jl:28-102
 * T_nzval_host[3k-1] = nzval of the (k+1) x k tridiagonal SparseMatrixCSC of :35-48: column 1 holds (T11, T21),
 * column i >= 2 holds (T[i-1,i], T[i,i], T[i+1,i]). */
int khip_hermitian_lanczos(khip_ctx *ctx, const khip_operator *A, int64_t n, const double *b, int k,
                           int allow_breakdown, int reorthogonalization, double *V, int64_t ldv,
                           double *beta1_host, double *T_nzval_host);
/* V, beta, H = arnoldi(A, b, k; allow_breakdown, reorthogonalization)                 src/krylov_processes.jl:250-296
 * H_host = dense (k+1) x k upper Hessenberg, column-major with leading dimension k+1. */
int khip_arnoldi(khip_ctx *ctx, const khip_operator *A, int64_t n, const double *b, int k, int allow_breakdown,
                 int reorthogonalization, double *V, int64_t ldv, double *beta_host, double *H_host);
/* V, U, beta1, L = golub_kahan(A, b, k; allow_breakdown)                              src/krylov_processes.jl:323-398
 * A is m x n, At its adjoint (khip_csr_transpose or a callback); V is n x (k+1), U is m x (k+1);
 * L_nzval_host[2k+1] = nzval of the (k+1) x (k+1) lower bidiagonal SparseMatrixCSC of :331-347:
 * (L11, L21, L22, L32, ..., L[k+1,k], L[k+1,k+1]). */
int khip_golub_kahan(khip_ctx *ctx, const khip_operator *A, const khip_operator *At, int64_t m, int64_t n,
                     const double *b, int k, int allow_breakdown, double *V, int64_t ldv, double *U, int64_t ldu,
                     double *beta1_host, double *L_nzval_host);
/* V, beta1, T, U, gamma1, Tt = nonhermitian_lanczos(A, b, c, k; allow_breakdown)   src/krylov_processes.jl:133-222
 * A square, At its adjoint; V, U are n x (k+1); T_nzval_host / Tt_nzval_host [3k-1] = nzval of T and of T^H in the
 * tridiagonal pattern above (column i: T[i-1,i] = gamma_i, T[i,i] = alpha_i, T[i+1,i] = beta_{i+1}). */
int khip_nonhermitian_lanczos(khip_ctx *ctx, const khip_operator *A, const khip_operator *At, int64_t n, const double *b,
                              const double *c, int k, int allow_breakdown, double *V, int64_t ldv, double *U, int64_t ldu,
                              double *beta1_host, double *gamma1_host, double *T_nzval_host, double *Tt_nzval_host);
/* V, beta1, T, U, gamma1, Tt = saunders_simon_yip(A, b, c, k; allow_breakdown)      src/krylov_processes.jl:431-524
 * A is m x n; V is m x (k+1), U is n x (k+1); same storage of T and T^H as above. */
int khip_saunders_simon_yip(khip_ctx *ctx, const khip_operator *A, const khip_operator *At, int64_t m, int64_t n,
                            const double *b, const double *c, int k, int allow_breakdown, double *V, int64_t ldv,
                            double *U, int64_t ldu, double *beta1_host, double *gamma1_host, double *T_nzval_host,
                            double *Tt_nzval_host);
/* V, beta, H, U, gamma, F = montoison_orban(A, B, b, c, k; allow_breakdown, reorthogonalization)
 *                                                                                   src/krylov_processes.jl:553-632
 * A is m x n, B is n x m; V is m x (k+1), U is n x (k+1); H_host, F_host dense (k+1) x k column-major. */
int khip_montoison_orban(khip_ctx *ctx, const khip_operator *A, const khip_operator *B, int64_t m, int64_t n,
                         const double *b, const double *c, int k, int allow_breakdown, int reorthogonalization,
                         double *V, int64_t ldv, double *U, int64_t ldu, double *beta_host, double *gamma_host,
                         double *H_host, double *F_host);

#ifdef __cplusplus
}
#endif
#endif /* KRYLOV_HIP_H */
