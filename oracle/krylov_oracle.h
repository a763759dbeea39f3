/*
 * krylov_oracle.h -- CPU restatement of the Krylov.jl hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under krylov.jl_amd/ may include, link
 * or call this.  Allowed users: tests/, __graft_entry__.smoke(), and the
 * cpu_baseline leg of bench.py.
 *
 * Every function cites the reference file:line (relative to the upstream
 * Krylov.jl v0.10.8 tree) whose behaviour it restates.  The arithmetic that
 * the reference delegates to third-party code (OpenBLAS through
 * libblastrampoline, SparseArrays.mul!, LAPACK xGEQRF/xORGQR/xORMQR -- none
 * of them vendored or version-pinned upstream, no Manifest.toml) is restated
 * from the published BLAS / LAPACK definitions with a FIXED, documented
 * summation order (see krylov_oracle.c header).
 *
 * Parity pinning: see oracle/README.md.  Pinned against (1) the known-answer
 * values of test/test_aux.jl, (2) the reference's own C conformance clients
 * interfaces/test/C/*.c and interfaces/examples/C/basic_cg.c compiled from
 * where they lie against oracle/ref_capi_shim.c, (3) scipy/LAPACK for the
 * dense kernels.  Residual HISTORIES at 64^3..512^3 are NOT pinned by any
 * reference artefact (Julia is absent here): "parity unpinned" for those,
 * they are golden vectors of this oracle only.
 */
#ifndef KRYLOV_ORACLE_H
#define KRYLOV_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* y = Op(x); x has the operator's column count, y its row count. */
typedef void (*ko_matvec)(const double *x, double *y, void *userdata);
/* Y = Op(X) for n-by-p column-major blocks. */
typedef void (*ko_block_matvec)(const double *X, double *Y, int p, void *userdata);
/* returns nonzero to request exit (callback(workspace)::Bool, src/cg.jl:264) */
typedef int (*ko_callback)(void *workspace, void *userdata);

/* ---- CSR container (0-based, int64 row pointers, int32 columns) ---------- */
typedef struct {
  int64_t n;        /* rows == cols */
  int64_t nnz;
  int64_t *rowptr;  /* n+1 */
  int32_t *col;     /* nnz */
  double  *val;     /* nnz */
} ko_csr;

/* generators (test/get_div_grad.jl:8-25, test/test_utils.jl:153-169) */
int  ko_csr_poisson3d(int n1, int n2, int n3, ko_csr *A);
int  ko_csr_kron_unsymmetric(int n1, ko_csr *A);
int  ko_csr_stencil27_unsym(int n1, ko_csr *A);   /* cfg-5 synthetic, documented in DESIGN.md */
/* banded + random, fixed seed (non-stencil benchmark operator; definition: krylov.jl_amd/csrc/gen_irregular.cpp header) */
int  ko_csr_banded_random(int64_t n, int half_band, int links, uint64_t seed, int unsym, int dense_rows, ko_csr *A);
int  ko_csr_tridiag(int n, double lo, double di, double up, ko_csr *A);
void ko_csr_free(ko_csr *A);
/* slice rows [r0,r1) keeping global column indices */
int  ko_csr_row_slice(const ko_csr *A, int64_t r0, int64_t r1, ko_csr *out);

/* y = A x : serial, one multiply and one add per entry in column order
 * (SparseArrays.mul! via src/krylov_utils.jl:305; docs/src/tips.md:38) */
void ko_spmv(const ko_csr *A, const double *x, double *y);
/* row-parallel OpenMP form (threaded_mul!, docs/src/tips.md:44-55) */
void ko_spmv_omp(const ko_csr *A, const double *x, double *y);
void ko_spmm(const ko_csr *A, const double *X, double *Y, int p); /* column-major n-by-p */
void ko_set_dot_mode(int mode);   /* 0 (default): the documented sequential extended-precision dots; 1: Dot2 (krylov_oracle.c) */
int  ko_get_dot_mode(void);
void ko_csr_matvec(const double *x, double *y, void *csr);        /* ko_matvec adaptor  */
void ko_csr_matvec_omp(const double *x, double *y, void *csr);
void ko_csr_block_matvec(const double *X, double *Y, int p, void *csr);
void ko_csr_block_matvec_omp(const double *X, double *Y, int p, void *csr);   /* same values, row-parallel */

/* Matrix-free get_div_grad(n1,n2,n3) (test/get_div_grad.jl:8-25): the 7-point product computed from the grid
 * indices, no CSR arrays.  Per row: the entries ko_csr_poisson3d stores, in its (ascending column) order
 * (-n1 n2, -n1, -1, 0, +1, +n1, +n1 n2 where the neighbour exists), accumulator +0.0, rounded product then
 * rounded add: every y value is bit-identical to ko_spmv on the CSR operator.  BASELINE cfg 4 (1024^3) needs
 * it: the CSR arrays alone would be 94 GB.  ud = a ko_stencil7. */
typedef struct { int n1, n2, n3; } ko_stencil7;
void ko_stencil7_matvec(const double *x, double *y, void *ud);       /* serial            */
void ko_stencil7_matvec_omp(const double *x, double *y, void *ud);   /* row-parallel: same values */

/* ---- ILU(0) / IC(0) preconditioner -------------------------
 * The reference gets these from the vendor library (ic02 / ilu02 of CUSPARSE resp. rocSPARSE,
 * docs/src/gpu.md:74-163, test/gpu/nvidia.jl:37-100) -- an un-vendored dependency.  Restated from the
 * published algorithm (Saad, Iterative Methods, alg. 10.4, IKJ variant on the pattern of A): for SPD A the
 * ILU(0) factors are L and U = D L^T, so M^{-1} = U^{-1} L^{-1} is the IC(0) preconditioner L_c L_c^T = A on
 * the pattern in exact arithmetic.  Column indices must be sorted within each row.
 * lu: nnz values (strict lower = L without its unit diagonal, upper incl. diagonal = U); diag[i] = position of
 * (i,i).  Returns 0, or -(i+1) for a missing / zero pivot in row i. */
int  ko_ilu0(const ko_csr *A, double *lu, int64_t *diag);
/* y = U^{-1} L^{-1} x: forward then backward substitution, each row accumulated in stored order with one
 * rounded multiply and one rounded subtract per entry (docs/src/gpu.md:95-99 ldiv_ic0!) */
void ko_ilu0_solve(const ko_csr *A, const double *lu, const int64_t *diag, const double *x, double *y);

/* ---- BLAS-1 shim (src/krylov_utils.jl:305-349) --------------------------- */
double ko_dot(int64_t n, const double *x, const double *y);                 /* :309-311 */
double ko_nrm2(int64_t n, const double *x);                                 /* :316-317 */
void   ko_scal(int64_t n, double s, double *x);                             /* :321-323 */
void   ko_div(int64_t n, double *x, double s);                              /* :325-326 */
void   ko_copy(int64_t n, double *y, const double *x);                      /* :328-329 (dest, src) */
void   ko_scalcopy(int64_t n, double *y, double s, const double *x);        /* :331-332 */
void   ko_divcopy(int64_t n, double *y, const double *x, double s);         /* :334-335 */
void   ko_axpy(int64_t n, double s, const double *x, double *y);            /* :337-339 */
void   ko_axpby(int64_t n, double s, const double *x, double t, double *y); /* :341-345 */
void   ko_fill(int64_t n, double *x, double val);                           /* :347 */
void   ko_ref(int64_t n, double *x, double *y, double c, double s);         /* :349 reflect! */
/* OpenMP variants used only by the all-core CPU baseline */
double ko_dot_omp(int64_t n, const double *x, const double *y);
void   ko_axpy_omp(int64_t n, double s, const double *x, double *y);
void   ko_axpby_omp(int64_t n, double s, const double *x, double t, double *y);
void   ko_set_threads(int nthreads);   /* 1 = faithful single-thread mode */
int    ko_get_threads(void);

/* ---- scalar helpers ------------------------------------------------------ */
void ko_sym_givens(double a, double b, double *c, double *s, double *rho);  /* src/krylov_utils.jl:21-51 */
int  ko_roots_quadratic(double q2, double q1, double q0, int nitref,
                        double *root1, double *root2);                      /* :110-152 */
int  ko_to_boundary(int64_t n, const double *x, const double *d, double radius,
                    int flip, double xNorm2, double dNorm2,
                    double *sigma1, double *sigma2);                        /* :375-402 (M = I) */

/* ---- options / stats (src/krylov_stats.jl:24-44, kwargs of each solver) --- */
typedef struct {
  double atol, rtol;        /* NaN -> sqrt(eps) (src/cg.jl:104-105) */
  int    itmax;             /* 0 -> 2n (src/cg.jl:177) */
  double timemax;           /* NaN/<=0 -> Inf */
  int    history;
  double radius;            /* cg only */
  int    linesearch;        /* cg only */
  int    restart;           /* gmres / block_gmres */
  int    reorthogonalization;
  int    ldiv;              /* kept for signature parity; M/N callbacks are applied as given */
  ko_callback callback; void *callback_data;
} ko_options;

typedef struct {
  int    niter, solved, inconsistent, indefinite, npcCount;
  double timer;
  char   status[96];
  double *residuals; int nres, cap;   /* filled when history != 0 */
  char   error[160];                  /* set when a solver returns -1 (Julia error(...)) */
} ko_stats;

ko_options ko_default_options(void);
void ko_stats_init(ko_stats *s);
void ko_stats_free(ko_stats *s);

/* ---- workspaces ----------------------------------------------------------- */
typedef struct {            /* src/krylov_workspaces.jl:236-291 */
  int64_t m, n;
  double *dx, *x, *r, *npc_dir, *p, *Ap, *z;
  int warm_start;
  ko_stats stats;
} ko_cg_workspace;

typedef struct {            /* src/krylov_workspaces.jl:2857-2924 */
  int64_t m, n;
  int mem;                  /* length(c) */
  int nV;                   /* length(V) (grows when restart=false) */
  double *dx, *x, *w, *p, *q;
  double **V;
  double *c, *s, *z, *R;    /* host scalars; R packed upper triangular */
  int capR, capcs, capz;
  int inner_iter;
  int warm_start;
  ko_stats stats;
} ko_gmres_workspace;

typedef struct {            /* src/krylov_workspaces.jl:1568-1629 */
  int64_t m, n;
  double *dx, *x, *r, *p, *v, *s, *qd, *yz, *t;
  int warm_start;
  ko_stats stats;
} ko_bicgstab_workspace;

typedef struct {            /* src/block_krylov_workspaces.jl:115-171 */
  int64_t m, n; int p;
  int mem, nV;
  double *dX, *X, *W, *P, *Q, *C, *D;
  double **V, **Z, **R, **H, **tau;
  int nR, nH;
  int warm_start;
  ko_stats stats;
} ko_block_gmres_workspace;

ko_cg_workspace          *ko_cg_workspace_create(int64_t m, int64_t n);
ko_gmres_workspace       *ko_gmres_workspace_create(int64_t m, int64_t n, int memory);
ko_bicgstab_workspace    *ko_bicgstab_workspace_create(int64_t m, int64_t n);
ko_block_gmres_workspace *ko_block_gmres_workspace_create(int64_t m, int64_t n, int p, int memory);
void ko_cg_workspace_free(ko_cg_workspace *ws);
void ko_gmres_workspace_free(ko_gmres_workspace *ws);
void ko_bicgstab_workspace_free(ko_bicgstab_workspace *ws);
void ko_block_gmres_workspace_free(ko_block_gmres_workspace *ws);
/* warm_start!(ws, x0)  (src/workspace_accessors.jl:193-200) */
void ko_cg_warm_start(ko_cg_workspace *ws, const double *x0);
void ko_gmres_warm_start(ko_gmres_workspace *ws, const double *x0);
void ko_bicgstab_warm_start(ko_bicgstab_workspace *ws, const double *x0);
void ko_block_gmres_warm_start(ko_block_gmres_workspace *ws, const double *X0);

/* ---- solvers: 0 ok, -1 = Julia error(...) (message in ws->stats.error) ---- */
/* cg!  src/cg.jl:120-291 ; M applies z <- M r (NULL = I) */
int ko_cg(ko_cg_workspace *ws, ko_matvec A, ko_matvec M, void *ud,
          const double *b, const ko_options *opts);
/* gmres!  src/gmres.jl:121-384 */
int ko_gmres(ko_gmres_workspace *ws, ko_matvec A, ko_matvec M, ko_matvec N, void *ud,
             const double *b, const ko_options *opts);
/* bicgstab!  src/bicgstab.jl:125-277 ; c == NULL -> c = b (:105) */
int ko_bicgstab(ko_bicgstab_workspace *ws, ko_matvec A, ko_matvec M, ko_matvec N, void *ud,
                const double *b, const double *c, const ko_options *opts);
/* block_gmres!  src/block_gmres.jl:110-358 */
int ko_block_gmres(ko_block_gmres_workspace *ws, ko_block_matvec A, ko_block_matvec M,
                   ko_block_matvec N, void *ud, const double *B, const ko_options *opts);

/* cg! (ko_cg, unchanged) on the matrix-free get_div_grad(n1,n2,n3) with b = ones: the cfg-4 oracle
 * ("matrix-free stencil oracle", 5 vectors = 43 GB at 1024^3).  Products run row-parallel when
 * ko_set_threads(>1) (values unchanged); the dots stay ko_dot's serial extended-precision sums.
 * x_idx / x_out: nsample entries of the solution copied out (the vector itself is freed).  The history is in st
 * (caller: ko_stats_init before, ko_stats_free after). */
int ko_cg_stencil7(int n1, int n2, int n3, const ko_options *opts, ko_stats *st,
                   int nsample, const int64_t *x_idx, double *x_out);

/* CPU-baseline loop for bench.py: seconds per CG iteration with `threads` OpenMP threads */
double ko_cg_bench(const ko_csr *A, int iters, int threads, double *rnorm_out);

/* ---- dense helpers restating LAPACK (src/block_krylov_utils.jl:192-301) --- */
void ko_geqrf(int m, int n, double *A, int lda, double *tau);               /* DGEQR2 */
void ko_orgqr(int m, int n, int k, double *A, int lda, const double *tau);  /* DORG2R */
void ko_ormqr_LT(int m, int n, int k, const double *A, int lda, const double *tau,
                 double *C, int ldc);                                       /* DORM2R side=L trans=T */
void ko_householder(int n, int k, double *Q, double *R, double *tau, int compact); /* :201-208 */

#ifdef __cplusplus
}
#endif
#endif
