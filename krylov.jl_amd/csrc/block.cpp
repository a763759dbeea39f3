// block.cpp -- block_gmres! (src/block_gmres.jl:110-358) above the panel kernels of panel.hip.
//
// Device side: the n x p blocks X, W, V[1..mem] live in HBM as row-major panels (panel.hip); SpMM,
// V^T Q, Q -= V Psi, X += V Y and the panel QR run there.  Host side: the p x p / 2p x p algebra of
// the block Hessenberg QR (LAPACK-style Householder on 2p x p blocks, block back-substitution) --
// a few kflop per iteration, exactly as the reference keeps it in small dense matrices.
//
// Panel QR: the reference calls householder!(Q, C, tau) = geqrf + orgqr on the n x p block
// (src/block_krylov_utils.jl:201-208): p BLAS-2 passes over the panel.  Here: CholeskyQR2 --
// G = Q^T Q (one MFMA pass), R = chol(G), Q <- Q R^-1 (one pass), repeated once -- 4 passes, all
// at HBM speed.  Cholesky leaves R = R2 R1 with a positive diagonal; LAPACK's R has Householder signs.  The factors
// differ by a diagonal S of +-1 (Q_h = Q S, R_h = S R), and S is recovered WITHOUT touching the panel again: Householder
// QR of an orthonormal Q is an LU factorisation without pivoting of Q - [S; 0] whose pivots -- hence the signs
// S_j = -sign(pivot_j) and the scalars tau_j = 1 + |pivot_j| -- only involve the TOP p x p block of Q (Ballard, Demmel,
// Grigori, Jacquelin, Nguyen, Solomonik: "Reconstructing Householder vectors from TSQR", IPDPS 2014).  That block is
// fetched (2 KB) before the last in-place scaling Q <- Q R^-1, S is folded into R^-1, and the panel comes out as
// geqrf + orgqr would leave it: same Q, same R, same tau (up to rounding), so the Psi / C / H blocks of block_gmres! are
// the reference's, not merely equivalent ones.  An ill-conditioned block (cond(Q)^2 > 1/eps)
// first takes a SHIFTED Cholesky pass (shifted CholeskyQR3) -- still entirely on the device (an exactly
// rank-deficient block, which the reference does not support either, docs/src/interfaces/reference.md:236,
// comes out like LAPACK's: A = QR with a tiny diagonal entry).  Only p x p matrices are ever handled on the host.
#include <chrono>
#include <cmath>
#include <limits>

#include "khip_internal.hpp"

using namespace khip;

extern "C" int khip_panel_rows(int64_t n, int64_t *n_pad);

namespace {

constexpr double kEps = std::numeric_limits<double>::epsilon();
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- small dense helpers, column-major (LAPACK DGEQR2 / DORG2R / DORM2R semantics) ----------
double nrm2_h(int n, const double *x) {
  long double s = 0;
  for (int i = 0; i < n; ++i) s += (long double)x[i] * x[i];
  return (double)std::sqrt((double)s);
}
void larfg(int n, double &alpha, double *x, double &tau) {
  if (n <= 1) { tau = 0; return; }
  const double xnorm = nrm2_h(n - 1, x);
  if (xnorm == 0) { tau = 0; return; }
  const double beta = -std::copysign(std::hypot(alpha, xnorm), alpha);
  tau = (beta - alpha) / beta;
  const double sc = 1.0 / (alpha - beta);
  for (int i = 0; i < n - 1; ++i) x[i] *= sc;
  alpha = beta;
}
void larf_left(int m, int n, const double *v, double tau, double *C, int ldc) {   // v[0] = 1 implicit
  if (tau == 0) return;
  for (int j = 0; j < n; ++j) {
    double *c = C + (size_t)j * ldc;
    double w = c[0];
    for (int i = 1; i < m; ++i) w += v[i] * c[i];
    const double tw = tau * w;
    c[0] -= tw;
    for (int i = 1; i < m; ++i) c[i] -= v[i] * tw;
  }
}
void geqr2(int m, int n, double *A, int lda, double *tau) {
  const int k = m < n ? m : n;
  for (int i = 0; i < k; ++i) {
    double *aii = A + (size_t)i * lda + i;
    larfg(m - i, aii[0], aii + (m - i > 1 ? 1 : 0), tau[i]);
    if (i < n - 1) larf_left(m - i, n - i - 1, aii, tau[i], A + (size_t)(i + 1) * lda + i, lda);
  }
}
void org2r(int m, int n, int k, double *A, int lda, const double *tau) {
  for (int j = k; j < n; ++j) {
    for (int l = 0; l < m; ++l) A[(size_t)j * lda + l] = 0;
    A[(size_t)j * lda + j] = 1;
  }
  for (int i = k - 1; i >= 0; --i) {
    double *aii = A + (size_t)i * lda + i;
    if (i < n - 1) larf_left(m - i, n - i - 1, aii, tau[i], A + (size_t)(i + 1) * lda + i, lda);
    for (int l = 1; l < m - i; ++l) aii[l] *= -tau[i];
    aii[0] = 1.0 - tau[i];
    for (int l = 0; l < i; ++l) A[(size_t)i * lda + l] = 0;
  }
}
void orm2r_LT(int m, int n, int k, const double *A, int lda, const double *tau, double *C, int ldc) {
  for (int i = 0; i < k; ++i) larf_left(m - i, n, A + (size_t)i * lda + i, tau[i], C + i, ldc);
}
// upper Cholesky G = R^T R (column-major p x p); false on breakdown
bool chol_upper(int p, const double *G, double *R) {
  for (int j = 0; j < p; ++j) {
    for (int i = 0; i <= j; ++i) {
      double s = G[(size_t)j * p + i];
      for (int l = 0; l < i; ++l) s -= R[(size_t)i * p + l] * R[(size_t)j * p + l];
      if (i == j) {
        if (!(s > 0) || !std::isfinite(s)) return false;
        R[(size_t)j * p + j] = std::sqrt(s);
      } else {
        R[(size_t)j * p + i] = s / R[(size_t)i * p + i];
      }
    }
    for (int i = j + 1; i < p; ++i) R[(size_t)j * p + i] = 0;
  }
  return true;
}
// Cholesky of a Gram matrix G (column-major p x p) that leaves out the columns lying, to working accuracy, in the span of the
// columns before them.  detect: column j joins the set D when its pivot -- its squared distance from that span -- is at most
// tol^2 times the largest diagonal entry; otherwise D = preset and a vanishing pivot elsewhere is a failure (*ok = false).
// Rhat (upper, column-major) is the factor of the other columns, with R_jj = 1 and a zero row j for j in D and, above the
// diagonal of such a column, its coefficients along the columns kept: Q <- Q Rhat^-1 orthonormalises the kept columns (as far
// as one Cholesky round does) and turns column j in D into its own remainder outside their span -- rounding dust.
unsigned deflating_chol(int p, const double *G, double tol, bool detect, unsigned preset, double *Rhat, bool *ok) {
  double gmax = 0.0;
  for (int j = 0; j < p; ++j) gmax = std::fmax(gmax, G[(size_t)j * p + j]);
  unsigned mask = detect ? 0u : preset;
  *ok = std::isfinite(gmax) && gmax > 0;
  for (int j = 0; j < p; ++j) {
    for (int i = 0; i < p; ++i) Rhat[(size_t)j * p + i] = 0.0;
    for (int i = 0; i <= j; ++i) {
      if (i < j && ((mask >> i) & 1u)) continue;                       // row of a column left out: zero
      double s = G[(size_t)j * p + i];
      for (int l = 0; l < i; ++l)
        if (!((mask >> l) & 1u)) s -= Rhat[(size_t)i * p + l] * Rhat[(size_t)j * p + l];
      if (i < j) { Rhat[(size_t)j * p + i] = s / Rhat[(size_t)i * p + i]; continue; }
      const bool vanishing = !(s > tol * tol * gmax) || !std::isfinite(s);
      if ((mask >> j) & 1u) Rhat[(size_t)j * p + j] = 1.0;
      else if (vanishing && detect) { mask |= 1u << j; Rhat[(size_t)j * p + j] = 1.0; }
      else if (vanishing) { *ok = false; Rhat[(size_t)j * p + j] = 1.0; }
      else Rhat[(size_t)j * p + j] = std::sqrt(s);
    }
  }
  return mask;
}
void inv_upper(int p, const double *R, double *Ri) {   // Ri = R^-1, both upper, column-major
  for (int j = 0; j < p; ++j) {
    for (int i = 0; i < p; ++i) Ri[(size_t)j * p + i] = 0;
    Ri[(size_t)j * p + j] = 1.0 / R[(size_t)j * p + j];
    for (int i = j - 1; i >= 0; --i) {
      double s = 0;
      for (int l = i + 1; l <= j; ++l) s += R[(size_t)l * p + i] * Ri[(size_t)j * p + l];
      Ri[(size_t)j * p + i] = -s / R[(size_t)i * p + i];
    }
  }
}
void matmul_pp(int p, const double *A, const double *B, double *C) {   // C = A B
  for (int j = 0; j < p; ++j)
    for (int i = 0; i < p; ++i) {
      double s = 0;
      for (int l = 0; l < p; ++l) s += A[(size_t)l * p + i] * B[(size_t)j * p + l];
      C[(size_t)j * p + i] = s;
    }
}

// R factor (p x p, row-major, arbitrary diagonal signs) of a small rows x p row-major matrix by unblocked Householder QR on the
// host: the last level of TSQR when several ranks contribute a triangle each.  Overwrites A.
void householder_r_host(int rows, int p, double *A, double *R) {
  for (int j = 0; j < p && j < rows; ++j) {
    double sigma = 0.0;
    for (int i = j + 1; i < rows; ++i) sigma += A[(size_t)i * p + j] * A[(size_t)i * p + j];
    const double alpha = A[(size_t)j * p + j];
    if (sigma == 0.0) continue;
    const double nrm = std::sqrt(alpha * alpha + sigma);
    const double beta = alpha >= 0.0 ? -nrm : nrm, tau = (beta - alpha) / beta, scale = 1.0 / (alpha - beta);
    for (int c = j + 1; c < p; ++c) {
      double w = A[(size_t)j * p + c];
      for (int i = j + 1; i < rows; ++i) w += A[(size_t)i * p + j] * scale * A[(size_t)i * p + c];
      A[(size_t)j * p + c] -= tau * w;
      for (int i = j + 1; i < rows; ++i) A[(size_t)i * p + c] -= tau * w * A[(size_t)i * p + j] * scale;
    }
    A[(size_t)j * p + j] = beta;
  }
  for (int i = 0; i < p; ++i)
    for (int c = 0; c < p; ++c) R[(size_t)i * p + c] = (c >= i && i < rows) ? A[(size_t)i * p + c] : 0.0;
}

// Signs and tau of LAPACK's Householder QR from the top p x p block Q1 (row-major, row i at Q1 + i p) of a panel with
// orthonormal columns: LU without pivoting of Q1 - S, the sign of each pivot chosen as DLARFG chooses beta
// (beta = -sign(alpha) |x|, sign(+0) = +).  Overwrites Q1.
void householder_signs(int p, int64_t n, double *Q1, double *S, double *tau) {
  for (int j = 0; j < p; ++j) {
    const double a = Q1[(size_t)j * p + j];
    if ((int64_t)j >= n - 1) {                         // DLARFG on a single entry: H = I, tau = 0, the entry keeps its sign
      S[j] = std::signbit(a) ? -1.0 : 1.0;
      tau[j] = 0.0;
      continue;
    }
    // DLARFG with an exactly zero sub-column (an identity-like or already triangular panel: the reduced column is +-e_j):
    // H = I, tau = 0, beta = alpha -- the entry keeps its sign.  Seen here as |pivot| == 1 with zeros below it in the top
    // block (rows past p cannot hold more than rounding dust then, the column has unit norm).
    bool unit_column = std::fabs(a) == 1.0;
    for (int i = j + 1; i < p && unit_column; ++i) unit_column = Q1[(size_t)i * p + j] == 0.0;
    if (unit_column) {
      S[j] = std::signbit(a) ? -1.0 : 1.0;
      tau[j] = 0.0;
      continue;                                        // nothing below the pivot to eliminate
    }
    S[j] = std::signbit(a) ? 1.0 : -1.0;
    tau[j] = 1.0 + std::fabs(a);
    const double piv = a - S[j];                       // |piv| >= 1
    Q1[(size_t)j * p + j] = piv;
    for (int i = j + 1; i < p; ++i) {
      const double l = Q1[(size_t)i * p + j] / piv;
      for (int k = j + 1; k < p; ++k) Q1[(size_t)i * p + k] -= l * Q1[(size_t)j * p + k];
    }
  }
}

struct StatsBoxB {
  khip_stats st;
  std::vector<double> residuals;
  int path = -1;       // khip_block_gmres_last_path: 1 once the library's loop (tile SpMM + fused panel sweeps) has run a solve
  StatsBoxB() { memset(&st, 0, sizeof(st)); snprintf(st.status, sizeof(st.status), "unknown"); }
  void reset() { residuals.clear(); st.residuals = nullptr; st.nres = 0; st.indefinite = 0; st.npcCount = 0; st.error[0] = 0; }
  void publish() { st.residuals = residuals.empty() ? nullptr : residuals.data(); st.nres = (int)residuals.size(); }
  int fail(int code, const char *msg) { snprintf(st.error, sizeof(st.error), "%s", msg); set_error("%s", msg); publish(); return code; }
  int fail_rc(int rc) { snprintf(st.error, sizeof(st.error), "%s", khip_last_error()); publish(); return rc; }
};

}  // namespace

extern "C" int khip_test_deflating_chol(int p, const double *G, double tol, int detect, unsigned preset, double *Rhat, int *ok, unsigned *mask) {
  KHIP_REQUIRE(p >= 1 && p <= 32 && G && Rhat && ok && mask, "test_deflating_chol: bad argument");
  bool good = false;
  *mask = deflating_chol(p, G, tol, detect != 0, preset, Rhat, &good);
  *ok = good ? 1 : 0;
  return KHIP_OK;
}
// which: 0 = geqr2 (A m x n column-major in place, tau), 1 = org2r (the first n columns of Q from geqr2's output, k = n),
// 2 = orm2r_LT (C <- Q^T C with Q = the k = n reflectors in A; C is m x nc), 3 = inv_upper (A n x n upper -> C = A^-1)
extern "C" int khip_test_small_dense(int which, int m, int n, int nc, double *A, double *tau, double *Cmat) {
  KHIP_REQUIRE(m >= 1 && n >= 1 && A, "test_small_dense: bad argument");
  switch (which) {
    case 0: KHIP_REQUIRE(tau, "test_small_dense: tau"); geqr2(m, n, A, m, tau); return KHIP_OK;
    case 1: KHIP_REQUIRE(tau && m >= n, "test_small_dense: tau"); org2r(m, n, n, A, m, tau); return KHIP_OK;
    case 2: KHIP_REQUIRE(tau && Cmat && nc >= 1, "test_small_dense: C"); orm2r_LT(m, nc, n < m ? n : m, A, m, tau, Cmat, m); return KHIP_OK;
    case 3: KHIP_REQUIRE(Cmat && m == n, "test_small_dense: C"); inv_upper(n, A, Cmat); return KHIP_OK;
    default: set_error("test_small_dense: which = 0 .. 3"); return KHIP_ERR_INVALID;
  }
}
extern "C" int khip_test_householder_r(int rows, int p, double *A, double *R) {
  KHIP_REQUIRE(rows >= 1 && p >= 1 && A && R, "test_householder_r: bad argument");
  householder_r_host(rows, p, A, R);
  return KHIP_OK;
}
extern "C" int khip_test_householder_signs(int p, int64_t n, double *Q1, double *S, double *tau) {
  KHIP_REQUIRE(p >= 1 && n >= 1 && Q1 && S && tau, "test_householder_signs: bad argument");
  householder_signs(p, n, Q1, S, tau);
  return KHIP_OK;
}

extern "C" int khip_panel_qr(khip_ctx *ctx, int64_t n, int p, double *Q, double *R_host) {
  return khip_panel_qr_tau(ctx, n, p, Q, R_host, nullptr);
}

static int panel_qr_tau_impl(khip_ctx *ctx, int64_t n, int p, double *Q, double *R_host, double *tau_host, const double *G0);
extern "C" int khip_panel_qr_tau(khip_ctx *ctx, int64_t n, int p, double *Q, double *R_host, double *tau_host) {
  return panel_qr_tau_impl(ctx, n, p, Q, R_host, tau_host, nullptr);
}
// G0: the Gram matrix Q^T Q of the panel as it comes in, when the caller already has it (khip::panel_mgs_gram), else null
static int panel_qr_tau_impl(khip_ctx *ctx, int64_t n, int p, double *Q, double *R_host, double *tau_host, const double *G0) {
  KHIP_REQUIRE(ctx && Q && R_host && p >= 1 && p <= 32, "panel_qr: bad argument (1 <= p <= 32)");
  const size_t pp = (size_t)p * p;
  std::vector<double> S((size_t)p, 1.0), tauv((size_t)p, 0.0);
  std::vector<double> G(pp), R(pp), Ri(pp), Racc(pp, 0.0), tmp(pp);
  for (int i = 0; i < p; ++i) Racc[(size_t)i * p + i] = 1.0;             // accumulated R = R_k ... R_1 (R_0)
  // CholeskyQR2: two rounds of G = Q'Q (FP64 MFMA), host Cholesky of the p x p Gram matrix, Q <- Q R^-1 in place.
  // Safe while cond(Q)^2 < 1/eps.  When the first Cholesky says otherwise (pivot ratio below 1e-7, or a
  // breakdown), a SHIFTED pass comes first (shifted CholeskyQR3, Fukaya-Kannan-Nakatsukasa-Yamamoto-Yanagisawa,
  // SISC 2020): chol(G + s I) with s = 11 (n p + p (p + 1)) eps ||Q||^2 always exists and brings cond(Q) down to
  // ~eps^-1/2, after which the two ordinary rounds converge.  Everything touching the n x p panel stays on the
  // device; only p x p matrices visit the host, as in the reference (src/block_gmres.jl:250-283).
  // rows of the whole panel: the shift of the shifted pass and DLARFG's single-entry case depend on it, and every rank of a
  // row-partitioned panel must apply the SAME factors to its rows
  int64_t n_global_rows = n, row_offset = 0;          // row_offset: global number of this rank's first panel row
  if (comm_nranks(ctx) > 1) {
    std::vector<double> counts((size_t)comm_nranks(ctx), 0.0);
    counts[(size_t)comm_rank_of(ctx)] = (double)n;
    KHIP_TRY(comm_allreduce_sum_host(ctx, counts.data(), (int)counts.size()));
    n_global_rows = 0;
    for (int r = 0; r < comm_nranks(ctx); ++r) {
      if (r == comm_rank_of(ctx)) row_offset = n_global_rows;
      n_global_rows += (int64_t)counts[(size_t)r];
    }
  }
  bool shifted_done = false, deflated = false;
  unsigned pending = 0;                   // columns to be replaced by stand-in directions after this round's scaling
  double fill_scale = 0.0;
  bool have_G = false;                    // G of this round came out of the kernel that applied the previous R^-1 (or of the caller's sweep)
  const bool tsqr = ctx->tune.panel_qr_tsqr != 0;
  if (G0 && !tsqr) { memcpy(G.data(), G0, sizeof(double) * pp); have_G = true; }
  for (int pass = 0; pass < 2; ++pass) {
    if (tsqr) {
      // R by TSQR (panel.hip: block Householder QRs out of LDS, tree of triangles) instead of chol(Q'Q): backward
      // stable for any conditioning, so no conditioning test and no shifted pass.  Row-partitioned panels: every rank reduces
      // its rows to one triangle, the triangles are gathered (a sum of zero-padded slots) and the host finishes the tree.
      // The diagonal is made positive, which makes R the Cholesky factor's equal in exact arithmetic; from there on the
      // round is the same: Q <- Q R^-1, LAPACK's signs and tau from the top block in the last round.
      std::vector<double> Rrow(pp);
      KHIP_TRY(panel_tsqr_r(ctx, n, p, Q, Rrow.data()));
      const int G_ = comm_nranks(ctx);
      if (G_ > 1) {
        std::vector<double> all((size_t)G_ * pp, 0.0);
        std::copy(Rrow.begin(), Rrow.end(), all.begin() + (size_t)comm_rank_of(ctx) * pp);
        KHIP_TRY(comm_allreduce_sum_host(ctx, all.data(), (int)all.size()));
        householder_r_host(G_ * p, p, all.data(), Rrow.data());           // QR of the stacked triangles
      }
      double dmax = 0, dmin = std::numeric_limits<double>::infinity();
      for (int i = 0; i < p; ++i) {
        const double d = std::fabs(Rrow[(size_t)i * p + i]);
        dmax = std::fmax(dmax, d); dmin = std::fmin(dmin, d);
      }
      const double thr = 16.0 * (double)p * std::numeric_limits<double>::epsilon() * dmax;
      if (!(dmin > thr) || !std::isfinite(dmax)) {
        // |R_jj| is the distance of column j from the span of the columns before it: those at rounding level are deflated as in
        // the Cholesky path below (the factor that isolates their remainders comes from the Gram matrix of the other columns)
        unsigned D = 0;
        if (!deflated && std::isfinite(dmax) && dmax > 0)
          for (int i = 0; i < p; ++i)
            if (!(std::fabs(Rrow[(size_t)i * p + i]) > thr)) D |= 1u << i;
        bool fac_ok = false;
        if (D != 0u) {
          KHIP_TRY(khip_panel_gemm_tn(ctx, n, p, Q, Q, G.data()));
          (void)deflating_chol(p, G.data(), 0.0, false, D, R.data(), &fac_ok);
        }
        if (D == 0u || !fac_ok) {
          set_error("panel_qr: the block is numerically rank deficient (block_gmres! needs full column rank)");
          return KHIP_ERR_NUMERIC;
        }
        pending = D;
        fill_scale = dmax / std::sqrt((double)(n_global_rows > 0 ? n_global_rows : 1));
      }
      if (pending == 0u) {
        for (int i = 0; i < p; ++i) {                                      // row-major -> upper, column-major, positive diagonal
          const double sgn = Rrow[(size_t)i * p + i] < 0 ? -1.0 : 1.0;
          for (int j = 0; j < p; ++j) R[(size_t)j * p + i] = j >= i ? sgn * Rrow[(size_t)i * p + j] : 0.0;
        }
      }
      have_G = false;
    }
    if (!tsqr && !have_G) KHIP_TRY(khip_panel_gemm_tn(ctx, n, p, Q, Q, G.data()));
    if (!tsqr) have_G = false;
    bool ok = tsqr ? true : chol_upper(p, G.data(), R.data());
    if (!tsqr && ok && pass == 0 && !shifted_done) {
      double dmax = 0, dmin = std::numeric_limits<double>::infinity();
      for (int i = 0; i < p; ++i) { dmax = std::fmax(dmax, R[(size_t)i * p + i]); dmin = std::fmin(dmin, R[(size_t)i * p + i]); }
      if (!(dmin > 1e-7 * dmax)) ok = false;                             // cond(Q)^2 would exceed 1/eps
    }
    if (!ok && (shifted_done || pass != 0)) {
      // Still no factor after the shifted pass: some columns lie in the span of the columns before them (equal or dependent
      // right-hand sides, a zero column, a block Krylov space that is exhausted in some directions).  The reference's
      // Householder QR goes on with a zero on R's diagonal and whatever unit vectors its reflectors complete the basis with
      // (src/block_gmres.jl:250-283 calls householder! on such blocks like on any other).  Same here: the columns D whose
      // Cholesky pivots vanish are left out of the factor; this round's scaling Q <- Q Rhat^-1 turns them into their own
      // remainders outside the span of the others (rounding dust), which are then replaced by fixed pseudo-random directions
      // while rows D of the accumulated R are zeroed -- W = Q' (Z Racc) up to that dust -- and the rounds start over on Q'.
      bool fac_ok = false;
      const unsigned D = deflated ? 0u : deflating_chol(p, G.data(), 1e-7, true, 0u, R.data(), &fac_ok);
      double gmax = 0.0;
      for (int i = 0; i < p; ++i) gmax = std::fmax(gmax, G[(size_t)i * p + i]);
      if (D == 0u || !fac_ok) {
        set_error("panel_qr: the block is numerically rank deficient (block_gmres! needs full column rank)");
        return KHIP_ERR_NUMERIC;
      }
      pending = D;
      fill_scale = std::sqrt(gmax / (double)(n_global_rows > 0 ? n_global_rows : 1));
      ok = true;
    }
    if (!ok) {
      double tr = 0;
      for (int i = 0; i < p; ++i) tr += G[(size_t)i * p + i];             // ||Q||_F^2 >= ||Q||_2^2
      if (!(tr > 0) || !std::isfinite(tr)) {
        set_error("panel_qr: the block is zero or not finite");
        return KHIP_ERR_NUMERIC;
      }
      const double shift = 11.0 * ((double)n_global_rows * p + (double)p * (p + 1)) * std::numeric_limits<double>::epsilon() * tr;
      for (int i = 0; i < p; ++i) G[(size_t)i * p + i] += shift;
      if (!chol_upper(p, G.data(), R.data())) {
        set_error("panel_qr: shifted Cholesky broke down");
        return KHIP_ERR_NUMERIC;
      }
      shifted_done = true;
      pass = -1;                                                          // two ordinary rounds follow
    }
    inv_upper(p, R.data(), Ri.data());
    if (pass == 1 && ctx->tune.panel_signs != 0 && pending == 0u) {
      // last scaling: LAPACK's column signs from the top block of the result, Q1 = (top block of Q) R^-1, folded into R^-1.
      // Row-partitioned panels: every rank contributes the rows of the top block it owns (normally all of them are rank 0's;
      // a rank with fewer than p rows leaves the rest to its successors), zeros elsewhere, and the block is the rank sum.
      std::vector<double> top(pp, 0.0), Q1(pp, 0.0);
      const int64_t first = row_offset < p ? row_offset : p;                              // global rows [first, first + have) of the top block
      const int64_t have = row_offset < p ? std::min<int64_t>(n, p - row_offset) : 0;
      if (have > 0) {
        KHIP_CHECK_HIP(hipMemcpyAsync(top.data() + (size_t)first * p, Q, sizeof(double) * (size_t)have * p, hipMemcpyDeviceToHost, ctx->stream));
        KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
      }
      if (comm_nranks(ctx) > 1) KHIP_TRY(comm_allreduce_sum_host(ctx, top.data(), (int)pp));
      for (int i = 0; i < p; ++i)
        for (int j = 0; j < p; ++j) {
          double acc = 0.0;
          for (int l = 0; l <= j; ++l) acc += top[(size_t)i * p + l] * Ri[(size_t)j * p + l];   // Ri upper, column-major
          Q1[(size_t)i * p + j] = acc;
        }
      householder_signs(p, n_global_rows, Q1.data(), S.data(), tauv.data());
      for (int j = 0; j < p; ++j)
        if (S[j] < 0)
          for (int i = 0; i < p; ++i) Ri[(size_t)j * p + i] = -Ri[(size_t)j * p + i];
    }
    if (pass == 0 && ctx->tune.panel_fuse != 0 && !tsqr && pending == 0u) {
      // first ordinary round: Q <- Q R^-1 and the Gram matrix of the second round in one pass (same bits)
      KHIP_TRY(panel_scale_gram(ctx, n, p, Q, Ri.data(), G.data()));
      have_G = true;
    } else {
      KHIP_TRY(khip_panel_gemm_nn(ctx, n, p, 1.0, Q, Ri.data(), 0.0, Q));   // in place: Q <- Q R^-1
    }
    matmul_pp(p, R.data(), Racc.data(), tmp.data());                      // Racc <- R * Racc
    Racc = tmp;
    if (pending != 0u) {
      KHIP_TRY(panel_fill_columns(ctx, n, p, Q, pending, fill_scale, 0x6b68697051520000ull + (unsigned long long)comm_rank_of(ctx)));
      for (int i = 0; i < p; ++i)
        if ((pending >> i) & 1u)
          for (int j = 0; j < p; ++j) Racc[(size_t)j * p + i] = 0.0;      // row i (column-major)
      pending = 0u;
      deflated = true;
      shifted_done = false;
      have_G = false;
      pass = -1;                                                          // the rounds start over on the completed panel
    }
  }
  memcpy(R_host, Racc.data(), sizeof(double) * pp);
  for (int j = 0; j < p; ++j)
    for (int i = 0; i < p; ++i) {
      if (i > j) R_host[(size_t)j * p + i] = 0;
      else if (S[i] < 0) R_host[(size_t)j * p + i] = -R_host[(size_t)j * p + i];      // R_h = S R
    }
  if (tau_host) memcpy(tau_host, tauv.data(), sizeof(double) * (size_t)p);
  return KHIP_OK;
}

// ================================================================== block-GMRES =====
struct khip_block_gmres_workspace {
  khip_ctx *ctx;
  int64_t m, n, np;
  int p, mem;
  double *dX = nullptr, *X = nullptr, *W = nullptr;            // panels (row-major, np x p)
  double *Pn = nullptr, *Qm = nullptr;                         // only with N / M preconditioners (src/block_gmres.jl:146-147)
  double *Bp = nullptr;                                        // panel copy of B
  std::vector<double *> V;
  std::vector<std::vector<double>> Z, R, H, tau;               // host p x p, p x p, 2p x p, p
  std::vector<double> C, D;                                    // host p x p, 2p x p (src/block_krylov_workspaces.jl:126-127)
  std::vector<double> gram;                                    // p x p: Gram matrix of the swept panel, from the sweep's last pass
  std::vector<double> sweep, Yall, tmp;                        // staging of the fused sweeps: mem p x p each (grown with the basis), p x p
  std::vector<const double *> Vp;
  bool warm_start = false;
  std::vector<const double *> borrowed;                        // panels of a caller's BlockGmresWorkspace (khip_block_gmres_workspace_adopt)
  khip_grow_fn grow = nullptr;                                 // the caller's push!(V, SM(undef, n, p)) (src/block_gmres.jl:300-305)
  void *grow_data = nullptr;
  StatsBoxB box;
  bool is_borrowed(const double *q) const { for (const double *b : borrowed) if (b == q) return true; return false; }
  void borrow(const double *q) { if (q && !is_borrowed(q)) borrowed.push_back(q); }
  void unborrow(const double *q) { for (size_t i = 0; i < borrowed.size(); ++i) if (borrowed[i] == q) { borrowed.erase(borrowed.begin() + (long)i); return; } }
};

#define KB(expr)                                        \
  do {                                                  \
    int rc_k = (expr);                                  \
    if (rc_k != KHIP_OK) return ws->box.fail_rc(rc_k);  \
  } while (0)

static thread_local double g_alloc_seconds = 0.0;              // stats.allocation_timer (allocate_if, src/krylov_utils.jl:290-297)
static int alloc_panel(khip_ctx *ctx, int64_t np, int p, double **out) {
  const double t_alloc = now_s();
  const int rc_alloc = khip_malloc(ctx, sizeof(double) * (size_t)np * p, reinterpret_cast<void **>(out));
  g_alloc_seconds += now_s() - t_alloc;
  KHIP_TRY(rc_alloc);
  return khip_fill(ctx, np * p, *out, 0.0);
}

static void block_host_arrays(khip_block_gmres_workspace *ws, int memory) {
  const int p = ws->p;
  const size_t pp = (size_t)p * p;
  ws->Z.assign(memory, std::vector<double>(pp, 0.0));
  ws->R.assign((size_t)memory * (memory + 1) / 2, std::vector<double>(pp, 0.0));
  ws->H.assign(memory, std::vector<double>(2 * pp, 0.0));
  ws->tau.assign(memory, std::vector<double>(p, 0.0));
  ws->C.assign(pp, 0.0);
  ws->D.assign(2 * pp, 0.0);
  ws->sweep.assign((size_t)memory * pp, 0.0);
  ws->Yall.assign((size_t)memory * pp, 0.0);
  ws->tmp.assign(pp, 0.0);
  ws->Vp.assign((size_t)memory, nullptr);
}

extern "C" {

int khip_block_gmres_workspace_create(khip_ctx *ctx, int64_t m, int64_t n, int p, int memory,
                                      khip_block_gmres_workspace **out) {
  KHIP_REQUIRE(ctx && out && m >= 0 && n >= 0 && p >= 1 && p <= 32, "block_gmres_workspace_create: bad argument (1 <= p <= 32)");
  if (memory <= 0) memory = 5;
  if ((int64_t)memory > n / p) memory = (int)(n / p);              // memory = min(div(n,p), memory)
  if (memory < 1) memory = 1;
  khip_block_gmres_workspace *ws = new khip_block_gmres_workspace();
  ws->ctx = ctx; ws->m = m; ws->n = n; ws->p = p; ws->mem = memory;
  khip_panel_rows(n, &ws->np);
  const double t_create = now_s();                                 // start_allocation_time, src/block_krylov_workspaces.jl:137
  int rc = alloc_panel(ctx, ws->np, p, &ws->X);
  if (!rc) rc = alloc_panel(ctx, ws->np, p, &ws->W);
  if (!rc) rc = alloc_panel(ctx, ws->np, p, &ws->Bp);
  for (int i = 0; i < memory && !rc; ++i) {
    double *v = nullptr;
    rc = alloc_panel(ctx, ws->np, p, &v);
    if (!rc) ws->V.push_back(v);
  }
  if (rc) { khip_block_gmres_workspace_destroy(ws); return rc; }
  block_host_arrays(ws, memory);
  g_alloc_seconds = 0.0;
  ws->box.st.allocation_timer = now_s() - t_create;                // workspace.stats.allocation_timer, :161
  *out = ws;
  return KHIP_OK;
}

// BlockGmresWorkspace on the CALLER's panels: X, W and the `memory` basis panels V_host[0 .. memory) (device pointers in a host
// array), each a row-major panel of khip_panel_rows(n) x p doubles whose padding rows are zero -- the tall blocks of a Julia
// BlockGmresWorkspace{Float64,Float64,Vector{Float64},HIPMatrix} (src/block_krylov_workspaces.jl:115-163; the small blocks
// Z, C, D, R, H, tau of the reference stay inside the library).  Solve with khip_block_gmres_solve_panel.
int khip_block_gmres_workspace_adopt(khip_ctx *ctx, int64_t m, int64_t n, int p, int memory, double *X, double *W,
                                     double *const *V_host, khip_block_gmres_workspace **out) {
  KHIP_REQUIRE(ctx && out && m >= 0 && n >= 0 && p >= 1 && p <= 32, "block_gmres_workspace_adopt: bad argument (1 <= p <= 32)");
  KHIP_REQUIRE(memory >= 1 && V_host && X && W && X != W, "block_gmres_workspace_adopt: X, W and `memory` >= 1 basis panels are needed");
  for (int i = 0; i < memory; ++i) KHIP_REQUIRE(V_host[i] != nullptr, "block_gmres_workspace_adopt: null basis panel");
  khip_block_gmres_workspace *ws = new khip_block_gmres_workspace();
  ws->ctx = ctx; ws->m = m; ws->n = n; ws->p = p; ws->mem = memory;
  khip_panel_rows(n, &ws->np);
  ws->X = X; ws->W = W;
  ws->borrow(X); ws->borrow(W);
  for (int i = 0; i < memory; ++i) { ws->V.push_back(V_host[i]); ws->borrow(V_host[i]); }
  block_host_arrays(ws, memory);
  g_alloc_seconds = 0.0;
  ws->box.st.allocation_timer = 0.0;
  *out = ws;
  return KHIP_OK;
}

int khip_block_gmres_workspace_adopt_panel(khip_block_gmres_workspace *ws, const char *name, double *ptr) {
  KHIP_REQUIRE(ws && name, "block_gmres_workspace_adopt_panel: null argument");
  struct { const char *k; double **slot; bool required; } tab[] = {
      {"X", &ws->X, true}, {"W", &ws->W, true}, {"P", &ws->Pn, false}, {"Q", &ws->Qm, false}, {"dX", &ws->dX, false}};
  for (auto &e : tab)
    if (strcmp(e.k, name) == 0) {
      KHIP_REQUIRE(ptr || !e.required, "block_gmres_workspace_adopt_panel: X and W cannot be emptied");
      if (*e.slot == ptr) return KHIP_OK;                       // nothing changes, in particular not who owns the panel
      if (ptr) {                                                // one panel, one slot (`borrowed` is a set of pointers)
        for (auto &o : tab) KHIP_REQUIRE(o.slot == e.slot || *o.slot != ptr, "block_gmres_workspace_adopt_panel: the pointer already is another panel of the workspace");
        for (const double *v : ws->V) KHIP_REQUIRE(v != ptr, "block_gmres_workspace_adopt_panel: the pointer already is a basis panel of the workspace");
      }
      if (*e.slot) { if (ws->is_borrowed(*e.slot)) ws->unborrow(*e.slot); else khip_free(ws->ctx, *e.slot); }
      *e.slot = ptr;
      ws->borrow(ptr);
      return KHIP_OK;
    }
  set_error("block_gmres_workspace_adopt_panel: unknown panel '%s' (X, W, P, Q, dX)", name);
  return KHIP_ERR_INVALID;
}

int khip_block_gmres_workspace_adopt_basis(khip_block_gmres_workspace *ws, int k, double *const *V_host) {
  KHIP_REQUIRE(ws && k >= 1 && V_host, "block_gmres_workspace_adopt_basis: bad argument");
  for (double *v : ws->V) KHIP_REQUIRE(ws->is_borrowed(v), "block_gmres_workspace_adopt_basis: this workspace owns its basis");
  KHIP_REQUIRE(k >= ws->mem, "block_gmres_workspace_adopt_basis: fewer panels than the workspace's memory");
  for (int i = 0; i < k; ++i) {
    KHIP_REQUIRE(V_host[i] != nullptr, "block_gmres_workspace_adopt_basis: null basis panel");
    for (const double *named : {ws->dX, ws->X, ws->W, ws->Pn, ws->Qm})
      KHIP_REQUIRE(V_host[i] != named, "block_gmres_workspace_adopt_basis: a basis panel is also one of X, W, P, Q, dX");
    for (int j = 0; j < i; ++j) KHIP_REQUIRE(V_host[i] != V_host[j], "block_gmres_workspace_adopt_basis: the same panel twice");
  }
  for (double *v : ws->V) ws->unborrow(v);
  ws->V.assign(V_host, V_host + k);
  for (double *v : ws->V) ws->borrow(v);
  return KHIP_OK;
}

int khip_block_gmres_workspace_set_grow(khip_block_gmres_workspace *ws, khip_grow_fn grow, void *userdata) {
  KHIP_REQUIRE(ws, "block_gmres_workspace_set_grow: null workspace");
  ws->grow = grow; ws->grow_data = userdata;
  return KHIP_OK;
}

int khip_block_gmres_workspace_destroy(khip_block_gmres_workspace *ws) {
  if (!ws) return KHIP_OK;
  for (double *v : {ws->dX, ws->X, ws->W, ws->Bp, ws->Pn, ws->Qm}) if (v && !ws->is_borrowed(v)) khip_free(ws->ctx, v);
  for (double *v : ws->V) if (v && !ws->is_borrowed(v)) khip_free(ws->ctx, v);
  delete ws;
  return KHIP_OK;
}

const khip_stats *khip_block_gmres_stats(khip_block_gmres_workspace *ws) { return ws ? &ws->box.st : nullptr; }
int khip_block_gmres_last_path(khip_block_gmres_workspace *ws) { return ws ? ws->box.path : -1; }

// Storage as test/test_allocations.jl:734-761 counts it (n x p blocks at their logical size, without the <= 15 padding
// rows of a panel): X, W, V[1..mem] (+ dX / P / Q when allocated) on the device, C, D, tau, Z, R, H on the host -- plus what
// this implementation adds: the row-major panel copy of B (the boundary takes the reference's column-major B) and the
// staging of the fused sweeps (2 mem + 1 blocks of p x p).  *extra_bytes (may be null) receives that addition.
size_t khip_block_gmres_workspace_bytes(khip_block_gmres_workspace *ws, size_t *extra_bytes) {
  if (!ws) return 0;
  const size_t block = sizeof(double) * (size_t)ws->n * ws->p;
  size_t panels = ws->V.size();
  for (double *v : {ws->dX, ws->X, ws->W, ws->Pn, ws->Qm}) panels += v ? 1 : 0;
  size_t host = ws->C.size() + ws->D.size();
  for (auto *grp : {&ws->Z, &ws->R, &ws->H, &ws->tau})
    for (const auto &blk : *grp) host += blk.size();
  const size_t extra = (ws->Bp ? block : 0) + sizeof(double) * (ws->sweep.size() + ws->Yall.size() + ws->tmp.size());
  if (extra_bytes) *extra_bytes = extra;
  return panels * block + sizeof(double) * host + extra;
}

int khip_block_gmres_get_X(khip_block_gmres_workspace *ws, double *X_colmajor) {
  KHIP_REQUIRE(ws && X_colmajor, "block_gmres_get_X: null argument");
  return khip_panel_to_colmajor(ws->ctx, ws->n, ws->p, ws->X, X_colmajor);
}

int khip_block_gmres_warm_start(khip_block_gmres_workspace *ws, const double *X0_colmajor) {
  KHIP_REQUIRE(ws && X0_colmajor, "block_gmres_warm_start: null argument");
  if (!ws->dX) KHIP_TRY(alloc_panel(ws->ctx, ws->np, ws->p, &ws->dX));
  KHIP_TRY(khip_panel_from_colmajor(ws->ctx, ws->n, ws->p, X0_colmajor, ws->dX));
  ws->warm_start = true;
  return KHIP_OK;
}

// warm_start!(workspace, X0) with X0 already a row-major panel (the caller's ΔX, src/workspace_accessors.jl:193-200);
// X0_panel == the adopted "dX" panel only sets the flag
int khip_block_gmres_warm_start_panel(khip_block_gmres_workspace *ws, const double *X0_panel) {
  KHIP_REQUIRE(ws && X0_panel, "block_gmres_warm_start_panel: null argument");
  if (!ws->dX) KHIP_TRY(alloc_panel(ws->ctx, ws->np, ws->p, &ws->dX));
  if (X0_panel != ws->dX) KHIP_TRY(khip_copy(ws->ctx, ws->np * ws->p, ws->dX, X0_panel));
  ws->warm_start = true;
  return KHIP_OK;
}

// Y <- Op X for row-major panels: CSR handle -> SpMM kernel; callback -> user code on the device panels
static int apply_block_op(khip_ctx *ctx, const khip_operator *op, const double *X, double *Y, int p) {
  if (op->apply) {
    const int rc = op->apply(op->self, X, Y);
    if (rc != 0) { set_error("user block operator returned %d", rc); return KHIP_ERR_INVALID; }
    return KHIP_OK;
  }
  if (!op->csr) { set_error("block operator has neither a CSR handle nor an apply callback"); return KHIP_ERR_INVALID; }
  return khip_spmm(ctx, op->csr, X, Y, p);
}

}  // extern "C"

// B_is_panel: B is a row-major panel of the caller (khip_block_gmres_solve_panel), read in place; otherwise the reference's
// column-major n x p array, converted into the workspace's own panel copy first.
static int block_gmres_solve_impl(khip_block_gmres_workspace *ws, const khip_operator *A, const khip_operator *M,
                                  const khip_operator *N, const double *B_in, bool B_is_panel, const khip_options *opts_in) {
  KHIP_REQUIRE(ws && A && B_in, "block_gmres_solve: null argument");
  khip_ctx *ctx = ws->ctx;
  khip_options o = opts_in ? *opts_in : khip_default_options();
  g_alloc_seconds = 0.0;
  const double t0 = now_s();
  const double timemax = (std::isnan(o.timemax) || o.timemax <= 0) ? std::numeric_limits<double>::infinity() : o.timemax;
  const int64_t n = ws->n, np = ws->np;
  const int p = ws->p;
  const int64_t len = np * p;                       // panel length in doubles (padding rows stay zero)
  const size_t pp = (size_t)p * p;
  khip_stats *st = &ws->box.st;
  const double atol = std::isnan(o.atol) ? std::sqrt(kEps) : o.atol, rtol = std::isnan(o.rtol) ? std::sqrt(kEps) : o.rtol;
  const bool restart = o.restart != 0, reorth = o.reorthogonalization != 0;
  if (ws->m != ws->n) return ws->box.fail(KHIP_ERR_INVALID, "System must be square");
  if (o.verbose > 0) klogf(o.log_fd, "BLOCK-GMRES: system of size %lld with %d right-hand sides\n", (long long)ws->n, ws->p);   // src/block_gmres.jl:120
  if (o.variant != 0) return ws->box.fail(KHIP_ERR_INVALID, "block_gmres: options.variant must be 0 (there is no other recurrence)");

  if (restart && !ws->dX) KB(alloc_panel(ctx, np, p, &ws->dX));
  if (!B_is_panel && !ws->Bp) KB(alloc_panel(ctx, np, p, &ws->Bp));                // adopted workspaces have no panel copy of B until a column-major B arrives
  double *dX = ws->dX, *X = ws->X, *W = ws->W;
  const double *Bp = B_is_panel ? B_in : ws->Bp;
  std::vector<double *> &V = ws->V;
  auto &Z = ws->Z; auto &R = ws->R; auto &H = ws->H; auto &tau = ws->tau;
  std::vector<double> &C = ws->C, &D = ws->D, &sweep = ws->sweep, &gram = ws->gram;      // in-place solve: no allocation per call (test/test_allocations.jl:752)
  bool have_gram = false;
  const bool warm_start = ws->warm_start;
  ws->box.reset();
  ws->box.path = 1;
  const bool MisI = (M == nullptr), NisI = (N == nullptr);
  if (!MisI && !ws->Qm) KB(alloc_panel(ctx, np, p, &ws->Qm));                      // :146
  if (!NisI && !ws->Pn) KB(alloc_panel(ctx, np, p, &ws->Pn));                      // :147
  double *Q = MisI ? W : ws->Qm, *R0 = MisI ? W : ws->Qm;
  double *Xr = restart ? dX : X;
  bool xr_zeroed = true;            // Xr holds what the update of :324-326 accumulates into (X itself without restart)

  if (!B_is_panel) KB(khip_panel_from_colmajor(ctx, n, p, B_in, ws->Bp));
  KB(khip_fill(ctx, len, X, 0.0));                                                 // src/block_gmres.jl:155
  if (warm_start) {
    KB(apply_block_op(ctx, A, dX, W, p));
    KB(khip_axpby(ctx, len, 1.0, Bp, -1.0, W));                                    // W .= B .- W
    if (restart) KB(khip_axpy(ctx, len, 1.0, dX, X));
  } else {
    KB(khip_copy(ctx, len, W, Bp));
  }
  if (!MisI) KB(apply_block_op(ctx, M, W, R0, p));                                 // R0 = M (B - A X0)  :165
  double RNorm;
  KB(khip_panel_norm(ctx, n, p, R0, &RNorm));                                      // :166
  if (o.history) ws->box.residuals.push_back(RNorm);
  const double eps_tol = atol + rtol * RNorm;

  const int mem = (int)V.size();
  int npass = 0;
  int64_t iter = 0;
  int inner_iter = 0;
  const int64_t itmax = o.itmax == 0 ? 2 * (global_rows(ctx, A, n) / p) : o.itmax;
  int64_t inner_itmax = itmax;
  const int verbose = o.verbose;                                                   // :181-182   pass  k  ‖Rₖ‖  timer
  if (verbose > 0) klogf(o.log_fd, " pass      k     \xe2\x80\x96R\xe2\x82\x96\xe2\x80\x96  timer\n");
  if (verbose > 0 && iter % verbose == 0) klogf(o.log_fd, "%5d  %5lld  %7.1e  %.2fs\n", npass, (long long)iter, RNorm, now_s() - t0);

  bool solved = RNorm <= eps_tol;
  bool tired = iter >= itmax;
  bool inner_tired = inner_iter >= inner_itmax;
  bool user_requested_exit = false, overtimed = false;
  const char *status = "unknown";

  while (!(solved || tired || user_requested_exit || overtimed)) {
    int nr = 0;
    // :195-197 zero-fills the mem basis panels here.  Every panel is written (copy of R0 / of the orthonormalised Q)
    // before anything reads it and this ABI has no accessor for V, so the fill is unobservable and its mem panel
    // writes (6.5 GB per restart at cfg 5) are skipped.
    for (auto &blk : R) std::fill(blk.begin(), blk.end(), 0.0);
    for (auto &blk : Z) std::fill(blk.begin(), blk.end(), 0.0);

    if (restart) {
      // :198 zero-fills Xr here and :324-326 accumulate into it.  Nothing reads Xr in between, so the update below starts
      // from beta = 0 instead (the panel kernels do not read X then: same bits as 1 * 0 + sum) and the fill + one panel read
      // per cycle go away -- unless a callback could look at the workspace's dX during the cycle.
      xr_zeroed = o.callback != nullptr;
      if (xr_zeroed) KB(khip_fill(ctx, len, Xr, 0.0));
      if (npass >= 1) {
        // the residual block of a restart goes straight into V[1] (the copy of :211 is the only reader of R0)
        double *Wr = MisI ? V[0] : W;
        KB(apply_block_op(ctx, A, X, Wr, p));
        KB(khip_axpby(ctx, len, 1.0, Bp, -1.0, Wr));
        if (!MisI) KB(apply_block_op(ctx, M, W, V[0], p));
      }
    }

    if (!(restart && npass >= 1)) KB(khip_copy(ctx, len, V[0], R0));               // :211
    KB(khip_panel_qr(ctx, n, p, V[0], Z[0].data()));                               // :212 householder!(V[1], Z[1], ..)

    npass = npass + 1;
    inner_iter = 0;
    inner_tired = false;

    while (!(solved || inner_tired || user_requested_exit || overtimed)) {
      inner_iter = inner_iter + 1;

      if (!restart && (inner_iter > mem)) {                                        // :224-232
        for (int i = 0; i < inner_iter; ++i) R.emplace_back(pp, 0.0);
        H.emplace_back(2 * pp, 0.0);
        tau.emplace_back(p, 0.0);
      }

      double *Pk = NisI ? V[inner_iter - 1] : ws->Pn;
      if (!NisI) KB(apply_block_op(ctx, N, V[inner_iter - 1], Pk, p));             // :241  P <- N V_k
      // Q of this iteration lives in the next basis panel when that exists: the copy of :307 disappears
      double *const Qsave = Q;
      if ((int)V.size() > inner_iter) Q = V[inner_iter];
      double *Wk = MisI ? Q : W;
      KB(apply_block_op(ctx, A, Pk, Wk, p));                                       // :242  W <- A N V_k
      if (!MisI) KB(apply_block_op(ctx, M, W, Q, p));                              // :243  Q <- M A N V_k
      // :244-247 (Psi_i = V_i^T Q ; Q -= V_i Psi_i for i = 1..k) and the reorthogonalisation pass :250-256, each as
      // one sweep whose blocks stay on the device between the steps
      {
        std::vector<const double *> &Vp = ws->Vp;
        if (Vp.size() < (size_t)inner_iter) Vp.resize((size_t)inner_iter);                // only when the basis grew (restart = false)
        for (int i = 0; i < inner_iter; ++i) Vp[i] = V[i];
        if (sweep.size() < (size_t)inner_iter * pp) sweep.resize((size_t)inner_iter * pp);
        std::fill(sweep.begin(), sweep.begin() + (size_t)inner_iter * pp, 0.0);
        // the LAST sweep also returns the Gram matrix of the swept panel (same pass as its last update): the QR's first round
        const bool gram_wanted = ctx->tune.panel_qr_tsqr == 0;
        if (gram.size() < pp) gram.resize(pp);
        KB(khip::panel_mgs_gram(ctx, n, p, inner_iter, Vp.data(), Q, sweep.data(), 0, (gram_wanted && !reorth) ? gram.data() : nullptr, &have_gram));
        for (int i = 0; i < inner_iter; ++i) std::copy(sweep.begin() + (size_t)i * pp, sweep.begin() + (size_t)(i + 1) * pp, R[nr + i].begin());
        if (reorth) {
          KB(khip::panel_mgs_gram(ctx, n, p, inner_iter, Vp.data(), Q, sweep.data(), 0, gram_wanted ? gram.data() : nullptr, &have_gram));
          for (int i = 0; i < inner_iter; ++i)
            for (size_t l = 0; l < pp; ++l) R[nr + i][l] += sweep[(size_t)i * pp + l];
        }
      }

      KB(panel_qr_tau_impl(ctx, n, p, Q, C.data(), nullptr, have_gram ? gram.data() : nullptr));   // :259 householder!(Q, C, ..)

      for (int i = 0; i < inner_iter - 1; ++i) {                                   // :263-269
        for (int j = 0; j < p; ++j)
          for (int l = 0; l < p; ++l) {
            D[(size_t)j * 2 * p + l] = R[nr + i][(size_t)j * p + l];
            D[(size_t)j * 2 * p + p + l] = R[nr + i + 1][(size_t)j * p + l];
          }
        orm2r_LT(2 * p, p, p, H[i].data(), 2 * p, tau[i].data(), D.data(), 2 * p);
        for (int j = 0; j < p; ++j)
          for (int l = 0; l < p; ++l) {
            R[nr + i][(size_t)j * p + l] = D[(size_t)j * 2 * p + l];
            R[nr + i + 1][(size_t)j * p + l] = D[(size_t)j * 2 * p + p + l];
          }
      }

      std::vector<double> &Hk = H[inner_iter - 1];                                 // :272-274
      std::vector<double> &tauk = tau[inner_iter - 1];
      for (int j = 0; j < p; ++j)
        for (int l = 0; l < p; ++l) {
          Hk[(size_t)j * 2 * p + l] = R[nr + inner_iter - 1][(size_t)j * p + l];
          Hk[(size_t)j * 2 * p + p + l] = C[(size_t)j * p + l];
        }
      geqr2(2 * p, p, Hk.data(), 2 * p, tauk.data());                              // householder!(H_k, R, tau, compact=true)
      std::vector<double> &Rkk = R[nr + inner_iter - 1];
      std::fill(Rkk.begin(), Rkk.end(), 0.0);
      for (int j = 0; j < p; ++j)
        for (int i = 0; i <= j; ++i) Rkk[(size_t)j * p + i] = Hk[(size_t)j * 2 * p + i];

      std::vector<double> &Zk = Z[inner_iter - 1];                                 // :277-280
      for (int j = 0; j < p; ++j)
        for (int l = 0; l < p; ++l) {
          D[(size_t)j * 2 * p + l] = Zk[(size_t)j * p + l];
          D[(size_t)j * 2 * p + p + l] = 0.0;
        }
      orm2r_LT(2 * p, p, p, Hk.data(), 2 * p, tauk.data(), D.data(), 2 * p);
      for (int j = 0; j < p; ++j)
        for (int l = 0; l < p; ++l) {
          Zk[(size_t)j * p + l] = D[(size_t)j * 2 * p + l];
          C[(size_t)j * p + l] = D[(size_t)j * 2 * p + p + l];                      // C .= D2 (:284)
        }
      RNorm = nrm2_h((int)pp, C.data());                                           // :285
      if (o.history) ws->box.residuals.push_back(RNorm);
      nr = nr + inner_iter;

      if (o.callback) { ws->box.publish(); user_requested_exit = o.callback(ws, o.callback_data) != 0; }
      solved = RNorm <= eps_tol;
      if (restart) {
        const int64_t lim = (int64_t)mem < inner_itmax ? (int64_t)mem : inner_itmax;
        inner_tired = inner_iter >= lim;
      } else {
        inner_tired = inner_iter >= inner_itmax;
      }
      overtimed = time_limit_reached(ctx, now_s() - t0, timemax);
      if (verbose > 0 && (iter + inner_iter) % verbose == 0)                       // :297
        klogf(o.log_fd, "%5d  %5lld  %7.1e  %.2fs\n", npass, (long long)(iter + inner_iter), RNorm, now_s() - t0);

      if (!(solved || inner_tired || user_requested_exit || overtimed)) {
        if (!restart && (inner_iter >= mem)) {                                     // :300-305
          while ((int)V.size() <= inner_iter) {
            double *v = nullptr;
            if (ws->grow) {                                                        // the caller's push!(V, SM(undef, n, p)): a zeroed panel
              const double t_grow = now_s();
              v = ws->grow(ws->grow_data);
              g_alloc_seconds += now_s() - t_grow;
              if (!v) return ws->box.fail(KHIP_ERR_INVALID, "block_gmres: the workspace's grow callback returned no panel");
              ws->borrow(v);
            } else {
              KB(alloc_panel(ctx, np, p, &v));
            }
            V.push_back(v);
          }
          while ((int)Z.size() <= inner_iter) Z.emplace_back(pp, 0.0);
        }
        if (Q != V[inner_iter]) KB(khip_copy(ctx, len, V[inner_iter], Q));         // :307
        for (int j = 0; j < p; ++j)
          for (int l = 0; l < p; ++l) Z[inner_iter][(size_t)j * p + l] = D[(size_t)j * 2 * p + p + l];
      }
      Q = Qsave;
    }

    // block back-substitution (:313-321), Y aliases Z
    auto &Y = Z;
    std::vector<double> &tmp = ws->tmp;
    for (int i = inner_iter; i >= 1; --i) {
      int pos = nr + i - inner_iter;
      for (int j = inner_iter; j >= i + 1; --j) {
        const std::vector<double> &Rm = R[pos - 1];
        for (int cc = 0; cc < p; ++cc)
          for (int rr = 0; rr < p; ++rr) {
            double acc = 0.0;
            for (int l = 0; l < p; ++l) acc += Rm[(size_t)l * p + rr] * Y[j - 1][(size_t)cc * p + l];
            tmp[(size_t)cc * p + rr] = acc;
          }
        for (size_t l = 0; l < pp; ++l) Y[i - 1][l] -= tmp[l];
        pos = pos - j + 1;
      }
      const std::vector<double> &U = R[pos - 1];
      for (int cc = 0; cc < p; ++cc) {
        double *ycol = Y[i - 1].data() + (size_t)cc * p;
        for (int rr = p - 1; rr >= 0; --rr) {
          double acc = ycol[rr];
          for (int l = rr + 1; l < p; ++l) acc -= U[(size_t)l * p + rr] * ycol[l];
          ycol[rr] = acc / U[(size_t)rr * p + rr];
        }
      }
    }

    {                                                                              // :324-326, the k products in one pass
      std::vector<const double *> &Vp = ws->Vp;
      std::vector<double> &Yall = ws->Yall;
      if (Vp.size() < (size_t)inner_iter) Vp.resize((size_t)inner_iter);
      if (Yall.size() < (size_t)inner_iter * pp) Yall.resize((size_t)inner_iter * pp);
      for (int i = 0; i < inner_iter; ++i) {
        Vp[i] = V[i];
        std::copy(Y[i].begin(), Y[i].begin() + pp, Yall.begin() + (size_t)i * pp);
      }
      KB(panel_multi_nn(ctx, n, p, inner_iter, Vp.data(), Yall.data(), xr_zeroed ? 1.0 : 0.0, Xr));
    }
    if (!NisI) {                                                                   // :327-330
      KB(khip_copy(ctx, len, ws->Pn, Xr));
      KB(apply_block_op(ctx, N, ws->Pn, Xr, p));
    }
    if (restart) KB(khip_axpy(ctx, len, 1.0, Xr, X));

    inner_itmax = inner_itmax - inner_iter;
    iter = iter + inner_iter;
    tired = iter >= itmax;
    overtimed = time_limit_reached(ctx, now_s() - t0, timemax);
  }

  if (verbose > 0) { klogf(o.log_fd, "\n"); klog_flush(o.log_fd); }                              // :340
  if (tired) status = "maximum number of iterations exceeded";
  if (solved) status = "solution good enough given atol and rtol";
  if (overtimed) status = "time limit exceeded";
  if (user_requested_exit) status = "user-requested exit";

  if (warm_start && !restart) KB(khip_axpy(ctx, len, 1.0, dX, X));
  ws->warm_start = false;
  KB(khip_ctx_sync(ctx));

  st->niter = (int)iter;
  st->solved = solved;
  st->timer = now_s() - t0;
  st->allocation_timer += g_alloc_seconds;                      // lazy allocations of this solve
  g_alloc_seconds = 0.0;
  snprintf(st->status, sizeof(st->status), "%s", status);
  ws->box.publish();
  return KHIP_OK;
}

extern "C" {

int khip_block_gmres_solve(khip_block_gmres_workspace *ws, const khip_operator *A, const khip_operator *M,
                           const khip_operator *N, const double *B_colmajor, const khip_options *opts_in) {
  return block_gmres_solve_impl(ws, A, M, N, B_colmajor, false, opts_in);
}

// block_gmres!(ws, A, B) with B a row-major panel of khip_panel_rows(n) x p doubles (padding rows zero), read in place: what a
// binding whose matrix type already is a panel passes (no column-major round trip, no panel copy of B).  The solution is the
// workspace's X panel (the caller's, for an adopted workspace).
int khip_block_gmres_solve_panel(khip_block_gmres_workspace *ws, const khip_operator *A, const khip_operator *M,
                                 const khip_operator *N, const double *B_panel, const khip_options *opts_in) {
  return block_gmres_solve_impl(ws, A, M, N, B_panel, true, opts_in);
}

}  // extern "C"
