/*
 * krylov_hip_test.h -- TEST-ONLY exports of libkrylov_hip.so.
 *
 * Not part of the drop-in boundary (include/krylov_hip.h): nothing in the Julia glue of INTEGRATION.md, the reference's
 * C / Fortran interface (libkrylov_hip_capi.so) or the examples calls these.  They let tests/ check host-side pieces of the
 * product against the reference's exact known answers (test/test_aux.jl:3-117) and against LAPACK without a device
 * (tests/test_abi.py, tests/test_panel_qr_host.py, tests/test_ilu_blocks_host.py) and may change or disappear at any time.
 */
#ifndef KRYLOV_HIP_TEST_H
#define KRYLOV_HIP_TEST_H

#include "krylov_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Test-only exports of the scalar helpers of the solver loops, so that the reference's exact known answers
 * (test/test_aux.jl:3-117) are checked against the copies the product runs: sym_givens (src/krylov_utils.jl:21-51),
 * roots_quadratic (:110-152), to_boundary with M = I (:375-402; x, d device vectors, the dots run on the device). */
int khip_test_sym_givens(double a, double b, double *c, double *s, double *rho);
int khip_test_roots_quadratic(double q2, double q1, double q0, int nitref, double *root1, double *root2);
int khip_test_to_boundary(khip_ctx *ctx, int64_t n, const double *x, const double *d, double radius, int flip,
                          double *sigma1, double *sigma2);

/* test-only measurement hook (tools/slab_iteration.py, tests/test_gpu_self_halo.py): a ONE-rank RCCL communicator whose slab's
 * off-slab columns wrap onto its own rows and travel through the real exchange plan (pack kernel, grouped ncclSend / ncclRecv to
 * itself on the halo stream, interior / boundary split, 16-byte all-gather + combine).  Set before khip_comm_init / the handle's
 * creation.  Not an option of khip_ctx_set_option: it changes the operator (periodic slab). */
int khip_test_set_halo_self(khip_ctx *ctx, int enable);

/* test-only, host-only: the analysis behind the block schedule on a CSR pattern in host memory (no device; checks that
 * every row lands in exactly one block and no block depends on a later one).  mode 1 = as khip_ilu0_create, 3 = level
 * sequence.  out10 = grid dims[3], skewed basis, blocks lower / upper, largest face list, 48-byte records possible,
 * largest row, rows of the largest block rounded up to the wave. */
int khip_test_ilu_blocks_host(int64_t n, const int64_t *rowptr, const int32_t *col, int mode, int64_t *out10);

/* test-only, host-only: the rows [row0, row0 + m) of khip_gen_banded_random into host arrays (no device): rowptr_out has m + 1
 * entries; col_out / val_out may be null on a first call that only asks for the row pointers and *nnz_out. */
int khip_test_gen_banded_random_host(int64_t n, int half_band, int links, uint64_t seed, int flags, int dense_rows, int64_t row0,
                                     int64_t m, int32_t *rowptr_out, int32_t *col_out, double *val_out, int64_t *nnz_out);
/* test-only, host-only: the p x p host steps of khip_panel_qr (no device).  deflating_chol: G = Gram matrix (column-major); detect
 * != 0: columns whose Cholesky pivot is <= tol^2 max_j G_jj are left out of the factor and returned as a bit mask, else the
 * columns `preset` are; Rhat (column-major, upper) = the factor of the other columns with R_jj = 1 and a zero row for those left
 * out; *ok = 0 when a column that is not left out has a vanishing pivot.  householder_r: R (row-major, upper) of a rows x p
 * row-major matrix by unblocked Householder QR (the last TSQR level across ranks); A is overwritten.
 * householder_signs: LAPACK's column signs S and tau of the Householder QR of an n x p panel with orthonormal columns from its
 * top p x p block Q1 (row-major; overwritten). */
int khip_test_deflating_chol(int p, const double *G, double tol, int detect, unsigned preset, double *Rhat, int *ok, unsigned *mask);
int khip_test_householder_r(int rows, int p, double *A, double *R);
int khip_test_householder_signs(int p, int64_t n, double *Q1, double *S, double *tau);
/* ... and the small dense routines block_gmres! runs on the host for its 2p x p blocks (column-major, LAPACK semantics): which = 0
 * DGEQR2 (A m x n in place, tau), 1 DORG2R (first n columns of Q in place of DGEQR2's output), 2 DORM2R('L', 'T') on C (m x nc),
 * 3 the inverse of an upper triangular n x n matrix into C. */
int khip_test_small_dense(int which, int m, int n, int nc, double *A, double *tau, double *Cmat);

/* test-only: how many lazily built accelerators (coded / delta column streams, SpMM tile records) failed to build for a reason
 * OTHER than an out-of-memory device since the library was loaded -- such a failure degrades to the plain kernels (results stay
 * right) and is a defect of the builder; the GPU test session asserts 0 (tests/conftest.py). */
int khip_test_optional_build_failures(int *count);

#ifdef __cplusplus
}
#endif
#endif /* KRYLOV_HIP_TEST_H */
