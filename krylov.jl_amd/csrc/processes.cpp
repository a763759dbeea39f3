// Krylov processes on the device: hermitian_lanczos, nonhermitian_lanczos, arnoldi, golub_kahan,
// saunders_simon_yip, montoison_orban -- all of src/krylov_processes.jl for Float64.
//
// Reference: src/krylov_processes.jl:28-102 (hermitian_lanczos), :133-222 (nonhermitian_lanczos), :250-296 (arnoldi),
// :323-398 (golub_kahan), :431-524 (saunders_simon_yip), :553-632 (montoison_orban).
// Same order of kmul! / kdotr / kaxpy! / knorm / kdivcopy! as the reference, on the same device primitives the
// solvers use: the orthogonalisation runs through the MGS cascade (khip_mgs) whose coefficients stay on the
// device until the one host read a step needs (the breakdown test of :92 / :288 is a host branch in the reference
// too).  Bases are dense column-major n x (k+1) blocks in HBM with a leading dimension (`M(undef, n, k+1)`, :52);
// the small matrices are returned on the host in the reference's own storage: T and L as the `nzval` of the
// SparseMatrixCSC the reference builds (:35-48, :331-347), H dense (k+1) x k column-major (:260).
// Row-partitioned operators work unchanged: n is then the local row count and every dot / norm is all-reduced
// inside the primitives.
#include <cmath>
#include <cstring>
#include <vector>

#include "khip_internal.hpp"

using namespace khip;

namespace {

int apply(khip_ctx *ctx, const khip_operator *op, const double *x, double *y) {
  if (op->apply) {
    const int rc = op->apply(op->self, x, y);
    if (rc != 0) { set_error("user operator returned %d", rc); return KHIP_ERR_INVALID; }
    return KHIP_OK;
  }
  if (!op->csr) { set_error("operator has neither a CSR handle nor an apply callback"); return KHIP_ERR_INVALID; }
  return khip_spmv(ctx, op->csr, x, y);
}

bool basis_ok(const double *V, int64_t n, int64_t ld) {
  return V && ld >= n && (ld % 2 == 0) && (reinterpret_cast<uintptr_t>(V) % 16 == 0);
}

// v <- b / ||b||  or  v <- 0 on an exact breakdown (:61-68, :264-271, :355-367)
int first_vector(khip_ctx *ctx, int64_t n, const double *b, double *v, int allow_breakdown, const char *what,
                 double *beta_out) {
  double beta = 0.0;
  KHIP_TRY(khip_nrm2(ctx, n, b, &beta));
  *beta_out = beta;
  if (beta == 0.0) {
    if (!allow_breakdown) { set_error("Exact breakdown %s == 0.", what); return KHIP_ERR_NUMERIC; }
    return khip_fill(ctx, n, v, 0.0);
  }
  return khip_divcopy(ctx, n, v, b, beta);
}

// q <- q / s in place, or q <- 0 with the reference's error when s == 0 and breakdowns are not allowed
int normalise(khip_ctx *ctx, int64_t n, double *q, double s, int allow_breakdown, const char *what, int it) {
  if (s == 0.0) {
    if (!allow_breakdown) { set_error("Exact breakdown %s == 0 at iteration i = %d.", what, it); return KHIP_ERR_NUMERIC; }
    return khip_fill(ctx, n, q, 0.0);
  }
  return khip_divcopy(ctx, n, q, q, s);
}

}  // namespace

extern "C" {

int khip_hermitian_lanczos(khip_ctx *ctx, const khip_operator *A, int64_t n, const double *b, int k,
                           int allow_breakdown, int reorthogonalization, double *V, int64_t ldv,
                           double *beta1_host, double *T_nzval_host) {
  KHIP_REQUIRE(ctx && A && b && beta1_host && T_nzval_host && n >= 0 && k >= 1, "hermitian_lanczos: bad argument");
  KHIP_REQUIRE(basis_ok(V, n, ldv), "hermitian_lanczos: V must be 16-byte aligned with an even leading dimension >= n");
  double *nz = T_nzval_host;
  memset(nz, 0, sizeof(double) * (size_t)(3 * k - 1));
  int pa = 0;                                                    // position of alpha_i in nzval (:54)
  for (int i = 0; i < k; ++i) {
    double *vi = V + (int64_t)i * ldv;
    double *q = V + (int64_t)(i + 1) * ldv;
    double *vim1 = i > 0 ? V + (int64_t)(i - 1) * ldv : nullptr;
    if (i == 0) KHIP_TRY(first_vector(ctx, n, b, vi, allow_breakdown, "β₁", beta1_host));   // :60-68
    KHIP_TRY(apply(ctx, A, vi, q));                              // :70
    if (i >= 1) {                                                // :71-76
      const double beta_i = nz[pa - 2];
      nz[pa - 1] = beta_i;
      KHIP_TRY(khip_axpy(ctx, n, -beta_i, vim1, q));
    }
    double alpha = 0.0, beta_next = 0.0;
    const double *one[1] = {vi};
    if (!reorthogonalization) {
      KHIP_TRY(khip_mgs(ctx, n, 1, one, q, &alpha, &beta_next, 0));            // :77-78 and :91 in one cascade
    } else {
      KHIP_TRY(khip_mgs(ctx, n, 1, one, q, &alpha, nullptr, 0));               // :77-78
      if (i >= 1) {                                                            // :80-85 then :86-88
        const double *two[2] = {vim1, vi};
        double t[2];
        KHIP_TRY(khip_mgs(ctx, n, 2, two, q, t, &beta_next, 0));
        nz[pa - 2] += t[0];
        nz[pa - 1] += t[0];
        alpha += t[1];
      } else {
        double t;
        KHIP_TRY(khip_mgs(ctx, n, 1, one, q, &t, &beta_next, 0));              // :86-88
        alpha += t;
      }
    }
    nz[pa] = alpha;                                                            // :90
    KHIP_TRY(normalise(ctx, n, q, beta_next, allow_breakdown, "βᵢ₊₁", i + 1));   // :92-97
    nz[pa + 1] = beta_next;                                                    // :98
    pa += 3;
  }
  return KHIP_OK;
}

int khip_arnoldi(khip_ctx *ctx, const khip_operator *A, int64_t n, const double *b, int k, int allow_breakdown,
                 int reorthogonalization, double *V, int64_t ldv, double *beta_host, double *H_host) {
  KHIP_REQUIRE(ctx && A && b && beta_host && H_host && n >= 0 && k >= 1, "arnoldi: bad argument");
  KHIP_REQUIRE(basis_ok(V, n, ldv), "arnoldi: V must be 16-byte aligned with an even leading dimension >= n");
  const int ldh = k + 1;
  memset(H_host, 0, sizeof(double) * (size_t)ldh * k);           // :260
  std::vector<const double *> cols((size_t)k + 1);
  for (int j = 0; j <= k; ++j) cols[j] = V + (int64_t)j * ldv;
  for (int j = 0; j < k; ++j) {
    double *vj = V + (int64_t)j * ldv;
    double *q = V + (int64_t)(j + 1) * ldv;
    if (j == 0) KHIP_TRY(first_vector(ctx, n, b, vj, allow_breakdown, "β", beta_host));   // :265-272
    KHIP_TRY(apply(ctx, A, vj, q));                              // :274
    double *h = H_host + (size_t)j * ldh;
    double hn = 0.0;
    KHIP_TRY(khip_mgs(ctx, n, j + 1, cols.data(), q, h, reorthogonalization ? nullptr : &hn, 0));   // :275-279
    if (reorthogonalization) KHIP_TRY(khip_mgs(ctx, n, j + 1, cols.data(), q, h, &hn, 1));          // :280-286
    h[j + 1] = hn;                                               // :287
    KHIP_TRY(normalise(ctx, n, q, hn, allow_breakdown, "Hᵢ₊₁.ᵢ", j + 1));   // :288-293
  }
  return KHIP_OK;
}

int khip_golub_kahan(khip_ctx *ctx, const khip_operator *A, const khip_operator *At, int64_t m, int64_t n,
                     const double *b, int k, int allow_breakdown, double *V, int64_t ldv, double *U, int64_t ldu,
                     double *beta1_host, double *L_nzval_host) {
  KHIP_REQUIRE(ctx && A && At && b && beta1_host && L_nzval_host && m >= 0 && n >= 0 && k >= 1, "golub_kahan: bad argument");
  KHIP_REQUIRE(basis_ok(V, n, ldv) && basis_ok(U, m, ldu),
               "golub_kahan: V and U must be 16-byte aligned with even leading dimensions >= n and >= m");
  double *nz = L_nzval_host;
  memset(nz, 0, sizeof(double) * (size_t)(2 * k + 1));
  int pa = 0;
  for (int i = 0; i < k; ++i) {
    double *ui = U + (int64_t)i * ldu, *vi = V + (int64_t)i * ldv;
    double *q = U + (int64_t)(i + 1) * ldu, *p = V + (int64_t)(i + 1) * ldv;
    if (i == 0) {                                                // :359-377
      KHIP_TRY(first_vector(ctx, m, b, ui, allow_breakdown, "β₁", beta1_host));
      KHIP_TRY(apply(ctx, At, ui, vi));
      double alpha1 = 0.0;
      KHIP_TRY(khip_nrm2(ctx, n, vi, &alpha1));
      if (alpha1 == 0.0 && !allow_breakdown) { set_error("Exact breakdown α₁ == 0."); return KHIP_ERR_NUMERIC; }
      KHIP_TRY(normalise(ctx, n, vi, alpha1, 1, "", 0));
      nz[pa] = alpha1;
    }
    KHIP_TRY(apply(ctx, A, vi, q));                              // :378
    const double alpha = nz[pa];
    double sq = 0.0;
    KHIP_TRY(khip_axpy_sqnorm(ctx, m, -alpha, ui, q, &sq));      // :380-381 in one pass
    const double beta_next = std::sqrt(sq);
    KHIP_TRY(normalise(ctx, m, q, beta_next, allow_breakdown, "βᵢ₊₁", i + 1));   // :382-387
    KHIP_TRY(apply(ctx, At, q, p));                              // :388
    KHIP_TRY(khip_axpy_sqnorm(ctx, n, -beta_next, vi, p, &sq));  // :389-390
    const double alpha_next = std::sqrt(sq);
    KHIP_TRY(normalise(ctx, n, p, alpha_next, allow_breakdown, "αᵢ₊₁", i + 1));  // :391-396
    nz[pa + 1] = beta_next;                                      // :397-398
    nz[pa + 2] = alpha_next;
    pa += 2;
  }
  return KHIP_OK;
}

int khip_nonhermitian_lanczos(khip_ctx *ctx, const khip_operator *A, const khip_operator *At, int64_t n, const double *b,
                              const double *c, int k, int allow_breakdown, double *V, int64_t ldv, double *U, int64_t ldu,
                              double *beta1_host, double *gamma1_host, double *T_nzval_host, double *Tt_nzval_host) {
  KHIP_REQUIRE(ctx && A && At && b && c && beta1_host && gamma1_host && T_nzval_host && Tt_nzval_host && n >= 0 && k >= 1,
               "nonhermitian_lanczos: bad argument");
  KHIP_REQUIRE(basis_ok(V, n, ldv) && basis_ok(U, n, ldu),
               "nonhermitian_lanczos: V and U must be 16-byte aligned with even leading dimensions >= n");
  double *nt = T_nzval_host, *nh = Tt_nzval_host;
  memset(nt, 0, sizeof(double) * (size_t)(3 * k - 1));
  memset(nh, 0, sizeof(double) * (size_t)(3 * k - 1));
  *beta1_host = 0.0;
  *gamma1_host = 0.0;
  int pa = 0;
  for (int i = 0; i < k; ++i) {
    double *vi = V + (int64_t)i * ldv, *ui = U + (int64_t)i * ldu;
    double *q = V + (int64_t)(i + 1) * ldv, *p = U + (int64_t)(i + 1) * ldu;
    if (i == 0) {                                                // :173-187
      double cb = 0.0;
      KHIP_TRY(khip_dot(ctx, n, c, b, &cb));
      if (cb == 0.0) {
        if (!allow_breakdown) { set_error("Exact breakdown β₁γ₁ == 0."); return KHIP_ERR_NUMERIC; }
        // the reference zero-fills v2 / u2 here and leaves v1 / u1 as allocated (undef); zeros make that deterministic
        KHIP_TRY(khip_fill(ctx, n, vi, 0.0));
        KHIP_TRY(khip_fill(ctx, n, ui, 0.0));
      } else {
        const double beta1 = std::sqrt(std::fabs(cb));
        const double gamma1 = cb / beta1;
        KHIP_TRY(khip_divcopy(ctx, n, vi, b, beta1));
        KHIP_TRY(khip_divcopy(ctx, n, ui, c, gamma1));
        *beta1_host = beta1;
        *gamma1_host = gamma1;
      }
    }
    KHIP_TRY(apply(ctx, A, vi, q));                              // :188
    KHIP_TRY(apply(ctx, At, ui, p));                             // :189
    if (i >= 1) {                                                // :190-197
      const double beta_i = nt[pa - 2], gamma_i = nt[pa - 1];
      KHIP_TRY(khip_axpy(ctx, n, -gamma_i, V + (int64_t)(i - 1) * ldv, q));
      KHIP_TRY(khip_axpy(ctx, n, -beta_i, U + (int64_t)(i - 1) * ldu, p));
    }
    double alpha = 0.0, pq = 0.0;
    KHIP_TRY(khip_dot(ctx, n, ui, q, &alpha));                   // :198
    nt[pa] = alpha;
    nh[pa] = alpha;
    KHIP_TRY(khip_axpy(ctx, n, -alpha, vi, q));                  // :201
    KHIP_TRY(khip_axpy(ctx, n, -alpha, ui, p));                  // :202
    KHIP_TRY(khip_dot(ctx, n, p, q, &pq));                       // :203
    double beta_next = 0.0, gamma_next = 0.0;
    if (pq == 0.0) {                                             // :204-209
      if (!allow_breakdown) { set_error("Exact breakdown βᵢ₊₁γᵢ₊₁ == 0 at iteration i = %d.", i + 1); return KHIP_ERR_NUMERIC; }
      KHIP_TRY(khip_fill(ctx, n, q, 0.0));
      KHIP_TRY(khip_fill(ctx, n, p, 0.0));
    } else {                                                     // :210-215
      beta_next = std::sqrt(std::fabs(pq));
      gamma_next = pq / beta_next;
      KHIP_TRY(khip_divcopy(ctx, n, q, q, beta_next));
      KHIP_TRY(khip_divcopy(ctx, n, p, p, gamma_next));
    }
    nt[pa + 1] = beta_next;                                      // :216-221
    nh[pa + 1] = gamma_next;
    if (i + 1 <= k - 1) {
      nt[pa + 2] = gamma_next;
      nh[pa + 2] = beta_next;
    }
    pa += 3;
  }
  return KHIP_OK;
}

int khip_saunders_simon_yip(khip_ctx *ctx, const khip_operator *A, const khip_operator *At, int64_t m, int64_t n,
                            const double *b, const double *c, int k, int allow_breakdown, double *V, int64_t ldv,
                            double *U, int64_t ldu, double *beta1_host, double *gamma1_host, double *T_nzval_host,
                            double *Tt_nzval_host) {
  KHIP_REQUIRE(ctx && A && At && b && c && beta1_host && gamma1_host && T_nzval_host && Tt_nzval_host && m >= 0 && n >= 0 &&
                   k >= 1, "saunders_simon_yip: bad argument");
  KHIP_REQUIRE(basis_ok(V, m, ldv) && basis_ok(U, n, ldu),
               "saunders_simon_yip: V and U must be 16-byte aligned with even leading dimensions >= m and >= n");
  double *nt = T_nzval_host, *nh = Tt_nzval_host;
  memset(nt, 0, sizeof(double) * (size_t)(3 * k - 1));
  memset(nh, 0, sizeof(double) * (size_t)(3 * k - 1));
  int pa = 0;
  for (int i = 0; i < k; ++i) {
    double *vi = V + (int64_t)i * ldv, *ui = U + (int64_t)i * ldu;
    double *q = V + (int64_t)(i + 1) * ldv, *p = U + (int64_t)(i + 1) * ldu;
    if (i == 0) {                                                // :470-485
      KHIP_TRY(first_vector(ctx, m, b, vi, allow_breakdown, "β₁", beta1_host));
      KHIP_TRY(first_vector(ctx, n, c, ui, allow_breakdown, "γ₁ᴴ", gamma1_host));
    }
    KHIP_TRY(apply(ctx, A, ui, q));                              // :486
    KHIP_TRY(apply(ctx, At, vi, p));                             // :487
    if (i >= 1) {                                                // :488-495
      const double beta_i = nt[pa - 2], gamma_i = nt[pa - 1];
      KHIP_TRY(khip_axpy(ctx, m, -gamma_i, V + (int64_t)(i - 1) * ldv, q));
      KHIP_TRY(khip_axpy(ctx, n, -beta_i, U + (int64_t)(i - 1) * ldu, p));
    }
    double alpha = 0.0, beta_next = 0.0, sq = 0.0;
    const double *one[1] = {vi};
    KHIP_TRY(khip_mgs(ctx, m, 1, one, q, &alpha, &beta_next, 0));            // :496, :499, :501 in one cascade
    nt[pa] = alpha;
    nh[pa] = alpha;
    KHIP_TRY(khip_axpy_sqnorm(ctx, n, -alpha, ui, p, &sq));                  // :500 and :508 in one pass
    const double gamma_next = std::sqrt(sq);
    KHIP_TRY(normalise(ctx, m, q, beta_next, allow_breakdown, "βᵢ₊₁", i + 1));   // :502-507
    KHIP_TRY(normalise(ctx, n, p, gamma_next, allow_breakdown, "γᵢ₊₁", i + 1));  // :509-514
    nt[pa + 1] = beta_next;                                      // :515-520
    nh[pa + 1] = gamma_next;
    if (i + 1 <= k - 1) {
      nt[pa + 2] = gamma_next;
      nh[pa + 2] = beta_next;
    }
    pa += 3;
  }
  return KHIP_OK;
}

int khip_montoison_orban(khip_ctx *ctx, const khip_operator *A, const khip_operator *B, int64_t m, int64_t n,
                         const double *b, const double *c, int k, int allow_breakdown, int reorthogonalization,
                         double *V, int64_t ldv, double *U, int64_t ldu, double *beta_host, double *gamma_host,
                         double *H_host, double *F_host) {
  KHIP_REQUIRE(ctx && A && B && b && c && beta_host && gamma_host && H_host && F_host && m >= 0 && n >= 0 && k >= 1,
               "montoison_orban: bad argument");
  KHIP_REQUIRE(basis_ok(V, m, ldv) && basis_ok(U, n, ldu),
               "montoison_orban: V and U must be 16-byte aligned with even leading dimensions >= m and >= n");
  const int ldh = k + 1;
  memset(H_host, 0, sizeof(double) * (size_t)ldh * k);
  memset(F_host, 0, sizeof(double) * (size_t)ldh * k);
  std::vector<const double *> vc((size_t)k + 1), uc((size_t)k + 1);
  for (int j = 0; j <= k; ++j) { vc[j] = V + (int64_t)j * ldv; uc[j] = U + (int64_t)j * ldu; }
  for (int j = 0; j < k; ++j) {
    double *vj = V + (int64_t)j * ldv, *uj = U + (int64_t)j * ldu;
    double *q = V + (int64_t)(j + 1) * ldv, *p = U + (int64_t)(j + 1) * ldu;
    if (j == 0) {                                                // :571-586
      KHIP_TRY(first_vector(ctx, m, b, vj, allow_breakdown, "β", beta_host));
      KHIP_TRY(first_vector(ctx, n, c, uj, allow_breakdown, "γ", gamma_host));
    }
    KHIP_TRY(apply(ctx, A, uj, q));                              // :587
    KHIP_TRY(apply(ctx, B, vj, p));                              // :588
    double *h = H_host + (size_t)j * ldh, *f = F_host + (size_t)j * ldh;
    double hn = 0.0, fn = 0.0;
    // the reference interleaves the two Gram-Schmidt sweeps (:589-607); they touch disjoint vectors, so each runs
    // as its own cascade with the same values in the same order
    KHIP_TRY(khip_mgs(ctx, m, j + 1, vc.data(), q, h, reorthogonalization ? nullptr : &hn, 0));
    KHIP_TRY(khip_mgs(ctx, n, j + 1, uc.data(), p, f, reorthogonalization ? nullptr : &fn, 0));
    if (reorthogonalization) {
      KHIP_TRY(khip_mgs(ctx, m, j + 1, vc.data(), q, h, &hn, 1));
      KHIP_TRY(khip_mgs(ctx, n, j + 1, uc.data(), p, f, &fn, 1));
    }
    h[j + 1] = hn;                                               // :608
    KHIP_TRY(normalise(ctx, m, q, hn, allow_breakdown, "Hᵢ₊₁.ᵢ", j + 1));   // :609-614
    f[j + 1] = fn;                                               // :615
    KHIP_TRY(normalise(ctx, n, p, fn, allow_breakdown, "Fᵢ₊₁.ᵢ", j + 1));   // :616-621
  }
  return KHIP_OK;
}

}  // extern "C"
