/*
 * krylov_hip_ext.h -- what libkrylov_hip_capi.so adds to the reference's C interface.
 *
 * Include AFTER the reference's own header:
 *
 *     #include "krylov.h"          // from Krylov.jl's interfaces/include (not redistributed here)
 *     #include "krylov_hip_ext.h"
 *
 * libkrylov_hip_capi.so exports every function of krylov.h for the solvers of the MI355X hot path (cg, gmres,
 * bicgstab, block_gmres; Float64) -- other (solver, dtype) pairs return -2 as an unknown pair does upstream -- and
 * accepts one more device enumerator next to KRYLOV_CPU (krylov.h:44-46):
 *
 *   KRYLOV_HIP   b, c, x0, the x / y (X / Y) handed to the callbacks and the buffer of krylov_get_x /
 *                krylov_block_get_X are DEVICE pointers (column-major blocks for the block interface); nothing is
 *                copied.  Callbacks must enqueue their work on krylov_hip_stream() (or synchronise themselves).
 */
#ifndef KRYLOV_HIP_EXT_H
#define KRYLOV_HIP_EXT_H

#ifdef __cplusplus
extern "C" {
#endif

#define KRYLOV_HIP ((KrylovDeviceType)1)

/* the khip_ctx (include/krylov_hip.h) and hipStream_t this library runs on */
void *krylov_hip_context(void);
void *krylov_hip_stream(void);

/* Attach a CSR operator (m x n of the workspace) that lives in HBM from then on; with it, krylov_solve /
 * krylov_block_solve accept matvec_A == NULL and run the fused kernels and device-resident loops instead of a
 * callback per product.  rowptr: 32- or 64-bit integers (rowptr_bits), index_base 0 or 1; on_device != 0 when the
 * three arrays are already device pointers.  Returns 0 / -1 (krylov_hip_last_error()). */
int krylov_hip_set_csr(void *ws, long long nnz, const void *rowptr, int rowptr_bits, const int *col, const double *val,
                       int index_base, int on_device);

/* 0 = issue the primitives exactly as the reference does, 1 = fused kernels, 2 (default) = fused + scalars on the device */
int krylov_hip_set_fused(void *ws, int level);

const char *krylov_hip_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
