// solver_device.hpp -- device-resident scalar state of the solver loops ("fused = 2").
//
// The reference's cg! (src/cg.jl:195-268) computes its scalars (alpha, beta, pNorm^2, the stopping
// tests) on the host between kernels: two host round trips per iteration.  Here the same scalar
// recurrences run as an EPILOGUE of the reduction that produces their input (last thread of the finish
// kernel, or of the cross-rank combine kernel), in the same IEEE double operations and order as the
// host code, and the vector kernels read alpha / beta from this struct.  The host only enqueues
// iterations ahead and polls a snapshot; once a stopping test fires the epilogue lowers `stop_seq` and
// every later kernel of the queue (each carries its own sequence number) returns immediately, so the
// vectors end in exactly the state the reference's loop leaves them in.
#pragma once

#include <hip/hip_runtime.h>

namespace khip {

enum Epilogue { EPI_NONE = 0, EPI_CG_STEP1 = 1, EPI_CG_STEP2 = 2, EPI_BICG_A = 3, EPI_BICG_B = 4, EPI_BICG_C = 5, EPI_CGCG = 6 };

struct CgDevState {
  double gamma;        // r.z of the current iterate              (src/cg.jl:162, 257)
  double pAp;          //                                          (:197)
  double alpha;        // gamma / pAp                              (:213)
  double beta;         // gamma_next / gamma                       (:256)
  double pNorm2;       //                                          (:257)
  double rNorm;        // sqrt(gamma_next)                         (:243)
  double eps_tol;      // atol + rtol * rNorm0                     (:186)
  double keps;         // eps(Float64) in the curvature test       (:198)
  long long stop_seq;  // kernels whose sequence number is >= stop_seq do nothing
  long long iter;      // completed iterations
  long long hist_base; // hist[k - 1 - hist_base] = rNorm after iteration k
  long long hist_cap;
  double *hist;        // device history window (null: no history)
  int solved, zero_curvature, inconsistent, not_spd;
};

// Single-reduction CG (Chronopoulos & Gear 1989): per iteration ONE reduction delivers
// gamma' = r.r and delta = (A r).r, from which beta = gamma'/gamma and alpha = gamma' / (delta - beta gamma'/alpha).
struct CgcgDevState {
  double gamma, alpha, beta, rNorm, eps_tol;
  long long stop_seq, iter, hist_base, hist_cap;
  double *hist;
  int solved, breakdown;      // breakdown: the alpha denominator is not positive (operator not SPD / loss of accuracy)
};

// bicgstab! (src/bicgstab.jl:213-253) with M = N = I: the scalars of one iteration
struct BicgDevState {
  double rho;          // c.r of the current iterate                 (:215)
  double alpha;        // rho / c.v                                   (:223)
  double omega;        // t.s / t.t                                   (:230)
  double beta;         // (next_rho / rho) (alpha / omega)            (:235)
  double rNorm;        // ||r||                                        (:240)
  double eps_tol;      // atol + rtol * rNorm0                        (:189)
  long long stop_seq;
  long long iter;
  long long hist_base;
  long long hist_cap;
  double *hist;
  int solved, breakdown;
};

constexpr long long kSeqNever = 0x7fffffffffffffffLL;

__device__ __forceinline__ bool seq_skip(const long long *stop_seq, long long seq) {
  return stop_seq != nullptr && seq >= *stop_seq;
}

// v = the finished reduction result(s); seq = sequence number of the kernel that produced it
__device__ inline void solver_epilogue(int epi, void *state, const double *v, long long seq) {
  if (epi == EPI_CG_STEP1) {                       // v[0] = p.Ap          src/cg.jl:197-213
    CgDevState *st = static_cast<CgDevState *>(state);
    const double pAp = v[0];
    st->pAp = pAp;
    if (pAp <= st->keps * st->pNorm2) {            // radius == 0, linesearch == false in this mode
      if (fabs(pAp) <= st->keps * st->pNorm2) {
        st->zero_curvature = 1;
        st->inconsistent = 1;
        st->stop_seq = seq + 1;                    // the rest of this iteration and everything after: no-ops
        return;
      }
    }
    st->alpha = st->gamma / pAp;
  } else if (epi == EPI_CG_STEP2) {                // v[0] = r.r after r -= alpha Ap     src/cg.jl:242-262
    CgDevState *st = static_cast<CgDevState *>(state);
    const double gamma_next = v[0];
    if (!(gamma_next >= 0)) {
      st->not_spd = 1;
      st->stop_seq = seq + 1;
      return;
    }
    const double rNorm = sqrt(gamma_next);
    st->rNorm = rNorm;
    const long long k = st->iter + 1;
    if (st->hist) {
      const long long idx = k - 1 - st->hist_base;
      if (idx >= 0 && idx < st->hist_cap) st->hist[idx] = rNorm;
    }
    const bool solved = (rNorm <= st->eps_tol) || (rNorm + 1.0 <= 1.0);
    if (!solved) {
      const double beta = gamma_next / st->gamma;
      st->beta = beta;
      st->pNorm2 = gamma_next + beta * beta * st->pNorm2;
      st->gamma = gamma_next;
    }
    st->solved = solved ? 1 : 0;
    st->iter = k;
    if (solved) st->stop_seq = seq + 2;            // the x update of this iteration (seq + 1) still runs
  } else if (epi == EPI_CGCG) {                    // v = (r.w, r.r) with w = A r
    CgcgDevState *st = static_cast<CgcgDevState *>(state);
    const double delta = v[0], gamma_next = v[1];
    const double rNorm = sqrt(gamma_next);
    st->rNorm = rNorm;
    const long long k = st->iter + 1;
    if (st->hist) {
      const long long idx = k - 1 - st->hist_base;
      if (idx >= 0 && idx < st->hist_cap) st->hist[idx] = rNorm;
    }
    const bool solved = (rNorm <= st->eps_tol) || (rNorm + 1.0 <= 1.0);
    const double beta = gamma_next / st->gamma;
    const double denom = delta - beta * gamma_next / st->alpha;
    const bool breakdown = !solved && !(denom > 0.0);
    st->solved = solved ? 1 : 0;
    st->breakdown = breakdown ? 1 : 0;
    st->iter = k;
    if (solved || breakdown) { st->stop_seq = seq + 1; return; }     // x, r already hold iterate k
    st->beta = beta;
    st->alpha = gamma_next / denom;
    st->gamma = gamma_next;
  } else if (epi == EPI_BICG_A) {                  // v[0] = c.v                 src/bicgstab.jl:223
    BicgDevState *st = static_cast<BicgDevState *>(state);
    st->alpha = st->rho / v[0];
  } else if (epi == EPI_BICG_B) {                  // v = (t.s, t.t)             :230
    BicgDevState *st = static_cast<BicgDevState *>(state);
    st->omega = v[0] / v[1];
  } else if (epi == EPI_BICG_C) {                  // v = (c.r, r.r)             :234-252
    BicgDevState *st = static_cast<BicgDevState *>(state);
    const double next_rho = v[0];
    st->beta = (next_rho / st->rho) * (st->alpha / st->omega);
    const double rNorm = sqrt(v[1]);
    st->rNorm = rNorm;
    const long long k = st->iter + 1;
    if (st->hist) {
      const long long idx = k - 1 - st->hist_base;
      if (idx >= 0 && idx < st->hist_cap) st->hist[idx] = rNorm;
    }
    const bool solved = (rNorm <= st->eps_tol) || (rNorm + 1.0 <= 1.0);
    const bool breakdown = (st->alpha == 0.0) || (st->alpha != st->alpha);
    st->solved = solved ? 1 : 0;
    st->breakdown = breakdown ? 1 : 0;
    st->rho = next_rho;
    st->iter = k;
    if (solved || breakdown) st->stop_seq = seq + 2;   // the p update of this iteration (seq + 1) still runs (:236-237)
  }
}

}  // namespace khip
