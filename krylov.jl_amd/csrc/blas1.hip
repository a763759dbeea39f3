// blas1.hip -- the BLAS-1 shim of Krylov.jl (src/krylov_utils.jl:309-349) as gfx950 kernels.
//
// Every kernel here is HBM-bound.  Layout: plain contiguous f64 vectors.  Launch shape (measured,
// profiles/r01_membench*.log): LOOP-FREE -- a workgroup of 256 lanes owns one contiguous tile of
// 256*U 16-byte vectors (U = 1 for the streaming maps, U = 4 for the reductions), every lane issues
// its U independent 16-byte accesses up front; vectors beyond `nt_min_elems` use non-temporal
// loads/stores (they cannot stay in the 256 MiB Infinity Cache anyway).  On MI355X this reaches
// copy 6.7, axpy 6.5, dot 7.1 TB/s versus 4.2-4.5 TB/s for 2048-workgroup grid-stride loops.
// Algorithmic bytes per element: dot 16 (8 if x === y), nrm2 8, axpy/axpby 24, copy/divcopy/
// scalcopy 16, fill 8, scal 16, reflect 32; fused: axpy2_dot 48, axpy_dev_dot 24(+8), waxpy 24.
#include "device_reduce.hpp"

namespace khip {

typedef double dbl2 __attribute__((ext_vector_type(2)));

enum MapOp {
  OP_COPY = 0,      // y = x
  OP_FILL = 1,      // y = a
  OP_SCAL = 2,      // y = a * y
  OP_SCALCOPY = 3,  // y = a * x
  OP_DIVCOPY = 4,   // y = x / a
  OP_AXPY = 5,      // y = fma(a, x, y)
  OP_AXPBY = 6,     // y = fma(a, x, b * y)
  OP_REF = 7,       // (x, y) = (a x + b y, b x - a y)      [a = c, b = s]
  OP_WAXPY = 8,     // w = fma(b, y, x)
  OP_VMUL = 9,      // w = x * y   (diagonal operator)
  OP_VDIV = 10,     // w = x / y   (Jacobi: z = r ./ diag(A))
};

template <int VEC> struct VecT;
template <> struct VecT<1> { using type = double; };
template <> struct VecT<2> { using type = dbl2; };

__device__ __forceinline__ double vget(const double &v, int) { return v; }
__device__ __forceinline__ double vget(const dbl2 &v, int i) { return i == 0 ? v.x : v.y; }
__device__ __forceinline__ void vset(double &v, int, double s) { v = s; }
__device__ __forceinline__ void vset(dbl2 &v, int i, double s) { if (i == 0) v.x = s; else v.y = s; }

template <bool NT, typename T> __device__ __forceinline__ T ldg(const T *p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}
template <bool NT, typename T> __device__ __forceinline__ void stg(T v, T *p) {
  if (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

template <int OP> __host__ __device__ constexpr bool reads_x() {
  return OP == OP_COPY || OP == OP_SCALCOPY || OP == OP_DIVCOPY || OP == OP_AXPY || OP == OP_AXPBY ||
         OP == OP_REF || OP == OP_WAXPY || OP == OP_VMUL || OP == OP_VDIV;
}
template <int OP> __host__ __device__ constexpr bool reads_y() {
  return OP == OP_SCAL || OP == OP_AXPY || OP == OP_AXPBY || OP == OP_REF || OP == OP_WAXPY || OP == OP_VMUL || OP == OP_VDIV;
}

template <int OP>
__device__ __forceinline__ void map_scalar(double a, double b, double xv, double yv, double &ox, double &oy) {
  ox = xv;
  if (OP == OP_COPY) oy = xv;
  else if (OP == OP_FILL) oy = a;
  else if (OP == OP_SCAL) oy = a * yv;
  else if (OP == OP_SCALCOPY) oy = a * xv;
  else if (OP == OP_DIVCOPY) oy = xv / a;
  else if (OP == OP_AXPY) oy = fma(a, xv, yv);
  else if (OP == OP_AXPBY) oy = fma(a, xv, b * yv);
  else if (OP == OP_REF) { ox = a * xv + b * yv; oy = b * xv - a * yv; }
  else if (OP == OP_WAXPY) oy = fma(b, yv, xv);
  else if (OP == OP_VMUL) oy = xv * yv;
  else if (OP == OP_VDIV) oy = xv / yv;
}

// x, y, w deliberately NOT __restrict__: exact aliasing is legal (BLAS semantics, src/bicgstab.jl:153-157).
template <int OP, int VEC, bool NT, int U>
__global__ __launch_bounds__(kBlock) void map_kernel(int64_t n, double a, double b, const double *x, double *y,
                                                     double *w) {
  using T = typename VecT<VEC>::type;
  const int64_t nvec = n / VEC;
  const int64_t base = (int64_t)blockIdx.x * (kBlock * U) + threadIdx.x;
  const T *X = reinterpret_cast<const T *>(x);
  T *Y = reinterpret_cast<T *>(y);
  T *W = reinterpret_cast<T *>((OP == OP_WAXPY || OP == OP_VMUL || OP == OP_VDIV) ? w : y);
  T *XO = reinterpret_cast<T *>(const_cast<double *>(x));
  T xv[U] = {}, yv[U] = {};
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t i = base + u * kBlock;
    if (i < nvec) {
      if (reads_x<OP>()) xv[u] = ldg<NT>(X + i);
      if (reads_y<OP>()) yv[u] = ldg<NT>(Y + i);
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t i = base + u * kBlock;
    if (i < nvec) {
      T ox, oy;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        double sx, sy;
        map_scalar<OP>(a, b, reads_x<OP>() ? vget(xv[u], e) : 0.0, reads_y<OP>() ? vget(yv[u], e) : 0.0, sx, sy);
        vset(ox, e, sx);
        vset(oy, e, sy);
      }
      stg<NT>(oy, W + i);
      if (OP == OP_REF) stg<NT>(ox, XO + i);
    }
  }
  if (VEC == 2 && (n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {   // odd tail element
    const int64_t t = n - 1;
    double sx, sy;
    map_scalar<OP>(a, b, reads_x<OP>() ? x[t] : 0.0, reads_y<OP>() ? y[t] : 0.0, sx, sy);
    ((OP == OP_WAXPY || OP == OP_VMUL || OP == OP_VDIV) ? w : y)[t] = sy;
    if (OP == OP_REF) const_cast<double *>(x)[t] = sx;
  }
}

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static inline bool use_nt(khip_ctx *ctx, int64_t n) { return n >= (int64_t)ctx->tune.nt_min_elems; }

static inline int64_t tiles_for(int64_t nvec, int u) {
  int64_t t = (nvec + (int64_t)kBlock * u - 1) / ((int64_t)kBlock * u);
  return t < 1 ? 1 : t;
}

template <int OP>
static int launch_map_op(khip_ctx *ctx, int64_t n, double a, double b, const double *x, double *y, double *w) {
  if (n <= 0) return KHIP_OK;
  constexpr bool uses_w = (OP == OP_WAXPY || OP == OP_VMUL || OP == OP_VDIV);
  const bool v2 = (!reads_x<OP>() || aligned16(x)) && aligned16(y) && (!uses_w || aligned16(w)) && n >= 2;
  const bool nt = use_nt(ctx, n);
  const int64_t nvec = v2 ? n / 2 : n;
  const int64_t g = tiles_for(nvec, 1);
  if (g > 0x7fffffffLL) { set_error("vector too long for one launch"); return KHIP_ERR_INVALID; }
#define KHIP_MAP(VEC, NT) \
  hipLaunchKernelGGL((map_kernel<OP, VEC, NT, 1>), dim3((unsigned)g), dim3(kBlock), 0, ctx->stream, n, a, b, x, y, w)
  if (v2) { if (nt) KHIP_MAP(2, true); else KHIP_MAP(2, false); }
  else    { if (nt) KHIP_MAP(1, true); else KHIP_MAP(1, false); }
#undef KHIP_MAP
  KHIP_CHECK_HIP(hipGetLastError());
  return KHIP_OK;
}

int launch_map(khip_ctx *ctx, int op, int64_t n, double a, double b, const double *x, double *y, double *w) {
  switch (op) {
    case OP_COPY: return launch_map_op<OP_COPY>(ctx, n, a, b, x, y, w);
    case OP_FILL: return launch_map_op<OP_FILL>(ctx, n, a, b, x, y, w);
    case OP_SCAL: return launch_map_op<OP_SCAL>(ctx, n, a, b, x, y, w);
    case OP_SCALCOPY: return launch_map_op<OP_SCALCOPY>(ctx, n, a, b, x, y, w);
    case OP_DIVCOPY: return launch_map_op<OP_DIVCOPY>(ctx, n, a, b, x, y, w);
    case OP_AXPY: return launch_map_op<OP_AXPY>(ctx, n, a, b, x, y, w);
    case OP_AXPBY: return launch_map_op<OP_AXPBY>(ctx, n, a, b, x, y, w);
    case OP_REF: return launch_map_op<OP_REF>(ctx, n, a, b, x, y, w);
    case OP_WAXPY: return launch_map_op<OP_WAXPY>(ctx, n, a, b, x, y, w);
    case OP_VMUL: return launch_map_op<OP_VMUL>(ctx, n, a, b, x, y, w);
    case OP_VDIV: return launch_map_op<OP_VDIV>(ctx, n, a, b, x, y, w);
    default: set_error("launch_map: unknown op %d", op); return KHIP_ERR_INVALID;
  }
}

// ---------------------------------------------------------------- CG direction update ----
// x <- fma(a, p, x) ; p <- fma(1, r, b * p): kaxpy!(n, a, p, x) (src/cg.jl:239) and kaxpby!(n, 1, r, b, p)
// (src/cg.jl:259) in ONE pass over p -- 40n bytes instead of 24n + 24n.  Same expressions as
// OP_AXPY / OP_AXPBY, so both outputs are bit-identical to the two separate kernels.
template <int VEC, bool NT>
__global__ __launch_bounds__(kBlock) void cg_update_kernel(int64_t n, double a, double b, const double *r, double *p,
                                                           double *x) {
  using T = typename VecT<VEC>::type;
  const int64_t nvec = n / VEC;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < nvec) {
    const T rv = ldg<NT>(reinterpret_cast<const T *>(r) + i);
    const T pv = ldg<NT>(reinterpret_cast<T *>(p) + i);
    const T xv = ldg<NT>(reinterpret_cast<T *>(x) + i);
    T xo, po;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      vset(xo, e, fma(a, vget(pv, e), vget(xv, e)));
      vset(po, e, fma(1.0, vget(rv, e), b * vget(pv, e)));
    }
    stg<NT>(xo, reinterpret_cast<T *>(x) + i);
    stg<NT>(po, reinterpret_cast<T *>(p) + i);
  }
  if (VEC == 2 && (n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t t = n - 1;
    const double pv = p[t];
    x[t] = fma(a, pv, x[t]);
    p[t] = fma(1.0, r[t], b * pv);
  }
}

int launch_cg_update(khip_ctx *ctx, int64_t n, double a, double b, const double *r, double *p, double *x) {
  if (n <= 0) return KHIP_OK;
  const bool v2 = n >= 2 && aligned16(r) && aligned16(p) && aligned16(x);
  const bool nt = use_nt(ctx, n);
  const int64_t g = tiles_for(v2 ? n / 2 : n, 1);
  if (g > 0x7fffffffLL) { set_error("vector too long for one launch"); return KHIP_ERR_INVALID; }
#define KHIP_CGU(VEC, NT) \
  hipLaunchKernelGGL((cg_update_kernel<VEC, NT>), dim3((unsigned)g), dim3(kBlock), 0, ctx->stream, n, a, b, r, p, x)
  if (v2) { if (nt) KHIP_CGU(2, true); else KHIP_CGU(2, false); }
  else    { if (nt) KHIP_CGU(1, true); else KHIP_CGU(1, false); }
#undef KHIP_CGU
  KHIP_CHECK_HIP(hipGetLastError());
  return KHIP_OK;
}

// ---------------------------------------------------------------- single-reduction CG update ----
// p <- r + beta p ; s <- w + beta s (s = A p without a product) ; x <- x + alpha p ; r <- r - alpha s
// (Chronopoulos & Gear 1989): all the vector work of one iteration in one pass, 72n bytes; alpha, beta from the
// device state the reduction epilogue maintains.
template <int VEC, bool NT>
__global__ __launch_bounds__(kBlock) void cgcg_update_kernel(int64_t n, const CgcgDevState *st, long long seq, const double *w,
                                                             double *r, double *p, double *s, double *x) {
  using T = typename VecT<VEC>::type;
  if (seq >= st->stop_seq) return;
  const double a = st->alpha, b = st->beta;
  const int64_t nvec = n / VEC;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < nvec) {
    const T wv = ldg<NT>(reinterpret_cast<const T *>(w) + i), rv = ldg<NT>(reinterpret_cast<T *>(r) + i);
    const T pv = ldg<NT>(reinterpret_cast<T *>(p) + i), sv = ldg<NT>(reinterpret_cast<T *>(s) + i);
    const T xv = ldg<NT>(reinterpret_cast<T *>(x) + i);
    T po, so, xo, ro;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const double pn = fma(b, vget(pv, e), vget(rv, e));
      const double sn = fma(b, vget(sv, e), vget(wv, e));
      vset(po, e, pn);
      vset(so, e, sn);
      vset(xo, e, fma(a, pn, vget(xv, e)));
      vset(ro, e, fma(-a, sn, vget(rv, e)));
    }
    stg<NT>(po, reinterpret_cast<T *>(p) + i);
    stg<NT>(so, reinterpret_cast<T *>(s) + i);
    stg<NT>(xo, reinterpret_cast<T *>(x) + i);
    stg<NT>(ro, reinterpret_cast<T *>(r) + i);
  }
  if (VEC == 2 && (n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t t = n - 1;
    const double pn = fma(b, p[t], r[t]), sn = fma(b, s[t], w[t]);
    p[t] = pn; s[t] = sn;
    x[t] = fma(a, pn, x[t]);
    r[t] = fma(-a, sn, r[t]);
  }
}

int launch_cgcg_update(khip_ctx *ctx, int64_t n, const void *st_dev, long long seq, const double *w, double *r, double *p,
                       double *s, double *x) {
  if (n <= 0) return KHIP_OK;
  const CgcgDevState *st = static_cast<const CgcgDevState *>(st_dev);
  const bool v2 = n >= 2 && aligned16(w) && aligned16(r) && aligned16(p) && aligned16(s) && aligned16(x);
  const bool nt = use_nt(ctx, n);
  const int64_t g = tiles_for(v2 ? n / 2 : n, 1);
  if (g > 0x7fffffffLL) { set_error("vector too long for one launch"); return KHIP_ERR_INVALID; }
#define KHIP_L(VEC, NT) \
  hipLaunchKernelGGL((cgcg_update_kernel<VEC, NT>), dim3((unsigned)g), dim3(kBlock), 0, ctx->stream, n, st, seq, w, r, p, s, x)
  if (v2) { if (nt) KHIP_L(2, true); else KHIP_L(2, false); }
  else    { if (nt) KHIP_L(1, true); else KHIP_L(1, false); }
#undef KHIP_L
  KHIP_CHECK_HIP(hipGetLastError());
  return KHIP_OK;
}

// ---------------------------------------------------------------- pipelined CG update ----
// Ghysels & Vanroose (2014): z <- q + beta z (z = A s) ; s <- w + beta s (s = A p) ; p <- r + beta p ; x <- x + alpha p ;
// r <- r - alpha s ; w <- w - alpha z (w = A r), with q = A w from the product that ran beside the reduction.  One pass,
// 7 reads + 6 writes = 104n bytes; alpha, beta from the device state the reduction epilogue maintains.
template <int VEC, bool NT>
__global__ __launch_bounds__(kBlock) void pcg_update_kernel(int64_t n, const CgcgDevState *st, long long seq, const double *q,
                                                            double *z, double *s, double *p, double *x, double *r, double *w) {
  using T = typename VecT<VEC>::type;
  if (seq >= st->stop_seq) return;
  const double a = st->alpha, b = st->beta;
  const int64_t nvec = n / VEC;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < nvec) {
    const T qv = ldg<NT>(reinterpret_cast<const T *>(q) + i), zv = ldg<NT>(reinterpret_cast<T *>(z) + i);
    const T sv = ldg<NT>(reinterpret_cast<T *>(s) + i), pv = ldg<NT>(reinterpret_cast<T *>(p) + i);
    const T xv = ldg<NT>(reinterpret_cast<T *>(x) + i), rv = ldg<NT>(reinterpret_cast<T *>(r) + i);
    const T wv = ldg<NT>(reinterpret_cast<T *>(w) + i);
    T zo, so, po, xo, ro, wo;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const double zn = fma(b, vget(zv, e), vget(qv, e));
      const double sn = fma(b, vget(sv, e), vget(wv, e));
      const double pn = fma(b, vget(pv, e), vget(rv, e));
      vset(zo, e, zn);
      vset(so, e, sn);
      vset(po, e, pn);
      vset(xo, e, fma(a, pn, vget(xv, e)));
      vset(ro, e, fma(-a, sn, vget(rv, e)));
      vset(wo, e, fma(-a, zn, vget(wv, e)));
    }
    stg<NT>(zo, reinterpret_cast<T *>(z) + i);
    stg<NT>(so, reinterpret_cast<T *>(s) + i);
    stg<NT>(po, reinterpret_cast<T *>(p) + i);
    stg<NT>(xo, reinterpret_cast<T *>(x) + i);
    stg<NT>(ro, reinterpret_cast<T *>(r) + i);
    stg<NT>(wo, reinterpret_cast<T *>(w) + i);
  }
  if (VEC == 2 && (n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t t = n - 1;
    const double zn = fma(b, z[t], q[t]), sn = fma(b, s[t], w[t]), pn = fma(b, p[t], r[t]);
    z[t] = zn; s[t] = sn; p[t] = pn;
    x[t] = fma(a, pn, x[t]);
    r[t] = fma(-a, sn, r[t]);
    w[t] = fma(-a, zn, w[t]);
  }
}

int launch_pcg_update(khip_ctx *ctx, int64_t n, const void *st_dev, long long seq, const double *q, double *z, double *s, double *p,
                      double *x, double *r, double *w) {
  if (n <= 0) return KHIP_OK;
  const CgcgDevState *st = static_cast<const CgcgDevState *>(st_dev);
  const bool v2 = n >= 2 && aligned16(q) && aligned16(z) && aligned16(s) && aligned16(p) && aligned16(x) && aligned16(r) && aligned16(w);
  const bool nt = use_nt(ctx, n);
  const int64_t g = tiles_for(v2 ? n / 2 : n, 1);
  if (g > 0x7fffffffLL) { set_error("vector too long for one launch"); return KHIP_ERR_INVALID; }
#define KHIP_L(VEC, NT) \
  hipLaunchKernelGGL((pcg_update_kernel<VEC, NT>), dim3((unsigned)g), dim3(kBlock), 0, ctx->stream, n, st, seq, q, z, s, p, x, r, w)
  if (v2) { if (nt) KHIP_L(2, true); else KHIP_L(2, false); }
  else    { if (nt) KHIP_L(1, true); else KHIP_L(1, false); }
#undef KHIP_L
  KHIP_CHECK_HIP(hipGetLastError());
  return KHIP_OK;
}

// ---------------------------------------------------------------- BiCGSTAB fused updates ----
// The elementwise work of one bicgstab! iteration (src/bicgstab.jl:224-237) in three passes instead of
// nine; every expression is the one the separate kernels use (OP_WAXPY, OP_AXPY, OP_AXPBY), so all vectors
// are bit-identical to the unfused sequence.
//   sx:  s = r - alpha v  (:224-225) ; x += alpha y  (:226)                                  48n bytes
//   xr:  x += omega z (:231) ; r = s - omega t (:232-233) ; c.r (:234) ; r.r (:240)          48n (40n if z === s)
//   p :  p = r + beta (p - omega v)  (:236-237)                                              32n
template <int VEC, bool NT>
__global__ __launch_bounds__(kBlock) void bicg_sx_kernel(int64_t n, double alpha, const BicgDevState *st, long long seq,
                                                         const double *r, const double *v, const double *y, double *s,
                                                         double *x) {
  using T = typename VecT<VEC>::type;
  if (st) {                                          // device-resident loop: scalar and stop word from the state
    if (seq >= st->stop_seq) return;
    alpha = st->alpha;
  }
  const int64_t nvec = n / VEC;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < nvec) {
    const T rv = ldg<NT>(reinterpret_cast<const T *>(r) + i), vv = ldg<NT>(reinterpret_cast<const T *>(v) + i);
    const T yv = ldg<NT>(reinterpret_cast<const T *>(y) + i), xv = ldg<NT>(reinterpret_cast<T *>(x) + i);
    T so, xo;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      vset(so, e, fma(-alpha, vget(vv, e), vget(rv, e)));
      vset(xo, e, fma(alpha, vget(yv, e), vget(xv, e)));
    }
    stg<NT>(so, reinterpret_cast<T *>(s) + i);
    stg<NT>(xo, reinterpret_cast<T *>(x) + i);
  }
  if (VEC == 2 && (n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t t = n - 1;
    const double rt = r[t], vt = v[t], yt = y[t];
    s[t] = fma(-alpha, vt, rt);
    x[t] = fma(alpha, yt, x[t]);
  }
}

template <int VEC, bool NT>
__global__ __launch_bounds__(kBlock) void bicg_p_kernel(int64_t n, double omega, double beta, const BicgDevState *st,
                                                        long long seq, const double *v, const double *r, double *p) {
  using T = typename VecT<VEC>::type;
  if (st) {
    if (seq >= st->stop_seq) return;
    omega = st->omega;
    beta = st->beta;
  }
  const int64_t nvec = n / VEC;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < nvec) {
    const T vv = ldg<NT>(reinterpret_cast<const T *>(v) + i), rv = ldg<NT>(reinterpret_cast<const T *>(r) + i);
    const T pv = ldg<NT>(reinterpret_cast<T *>(p) + i);
    T po;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const double p1 = fma(-omega, vget(vv, e), vget(pv, e));      // kaxpy!(n, -omega, v, p)
      vset(po, e, fma(1.0, vget(rv, e), beta * p1));               // kaxpby!(n, 1, r, beta, p)
    }
    stg<NT>(po, reinterpret_cast<T *>(p) + i);
  }
  if (VEC == 2 && (n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t t = n - 1;
    const double p1 = fma(-omega, v[t], p[t]);
    p[t] = fma(1.0, r[t], beta * p1);
  }
}

template <bool COMP, int VEC, bool NT>
__global__ __launch_bounds__(kBlock) void bicg_xr_kernel(int64_t n, double omega, const BicgDevState *st, const double *s,
                                                         const double *t, const double *z, const double *c, double *x,
                                                         double *r, RedArgs ra) {
  using T = typename VecT<VEC>::type;
  if (seq_skip(ra.stop_seq, ra.seq)) return;
  if (st) omega = st->omega;
  dd acc[2] = {dd{0.0, 0.0}, dd{0.0, 0.0}};
  const int64_t nvec = n / VEC;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < nvec) {
    const T sv = ldg<NT>(reinterpret_cast<const T *>(s) + i), tv = ldg<NT>(reinterpret_cast<const T *>(t) + i);
    const T zv = (z == s) ? sv : ldg<NT>(reinterpret_cast<const T *>(z) + i);
    const T cv = ldg<NT>(reinterpret_cast<const T *>(c) + i), xv = ldg<NT>(reinterpret_cast<T *>(x) + i);
    T xo, ro;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      vset(xo, e, fma(omega, vget(zv, e), vget(xv, e)));
      const double rn = fma(-omega, vget(tv, e), vget(sv, e));
      vset(ro, e, rn);
      acc_prod<COMP>(acc[0], vget(cv, e), rn);
      acc_prod<COMP>(acc[1], rn, rn);
    }
    stg<NT>(xo, reinterpret_cast<T *>(x) + i);
    stg<NT>(ro, reinterpret_cast<T *>(r) + i);
  }
  if (VEC == 2 && (n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t k = n - 1;
    const double sk = s[k], tk = t[k], zk = z[k], ck = c[k];
    x[k] = fma(omega, zk, x[k]);
    const double rn = fma(-omega, tk, sk);
    r[k] = rn;
    acc_prod<COMP>(acc[0], ck, rn);
    acc_prod<COMP>(acc[1], rn, rn);
  }
  wave_publish<2>(acc, ra);
}

int launch_bicg_sx(khip_ctx *ctx, int64_t n, double alpha, const double *r, const double *v, const double *y, double *s,
                   double *x, const void *st_dev, long long seq) {
  const BicgDevState *st = static_cast<const BicgDevState *>(st_dev);
  if (n <= 0) return KHIP_OK;
  const bool v2 = n >= 2 && aligned16(r) && aligned16(v) && aligned16(y) && aligned16(s) && aligned16(x);
  const bool nt = use_nt(ctx, n);
  const int64_t g = tiles_for(v2 ? n / 2 : n, 1);
  if (g > 0x7fffffffLL) { set_error("vector too long for one launch"); return KHIP_ERR_INVALID; }
#define KHIP_L(VEC, NT) \
  hipLaunchKernelGGL((bicg_sx_kernel<VEC, NT>), dim3((unsigned)g), dim3(kBlock), 0, ctx->stream, n, alpha, st, seq, r, v, y, s, x)
  if (v2) { if (nt) KHIP_L(2, true); else KHIP_L(2, false); }
  else    { if (nt) KHIP_L(1, true); else KHIP_L(1, false); }
#undef KHIP_L
  KHIP_CHECK_HIP(hipGetLastError());
  return KHIP_OK;
}

int launch_bicg_p(khip_ctx *ctx, int64_t n, double omega, double beta, const double *v, const double *r, double *p,
                  const void *st_dev, long long seq) {
  const BicgDevState *st = static_cast<const BicgDevState *>(st_dev);
  if (n <= 0) return KHIP_OK;
  const bool v2 = n >= 2 && aligned16(r) && aligned16(v) && aligned16(p);
  const bool nt = use_nt(ctx, n);
  const int64_t g = tiles_for(v2 ? n / 2 : n, 1);
  if (g > 0x7fffffffLL) { set_error("vector too long for one launch"); return KHIP_ERR_INVALID; }
#define KHIP_L(VEC, NT) \
  hipLaunchKernelGGL((bicg_p_kernel<VEC, NT>), dim3((unsigned)g), dim3(kBlock), 0, ctx->stream, n, omega, beta, st, seq, v, r, p)
  if (v2) { if (nt) KHIP_L(2, true); else KHIP_L(2, false); }
  else    { if (nt) KHIP_L(1, true); else KHIP_L(1, false); }
#undef KHIP_L
  KHIP_CHECK_HIP(hipGetLastError());
  return KHIP_OK;
}

int launch_bicg_xr(khip_ctx *ctx, int64_t n, double omega, const double *s, const double *t, const double *z,
                   const double *c, double *x, double *r, int slot, const void *st_dev) {
  const BicgDevState *st = static_cast<const BicgDevState *>(st_dev);
  if (n < 0) { set_error("negative length"); return KHIP_ERR_INVALID; }
  const bool v2 = n >= 2 && aligned16(s) && aligned16(t) && aligned16(z) && aligned16(c) && aligned16(x) && aligned16(r);
  const bool comp = ctx->tune.compensated != 0;
  const bool nt = use_nt(ctx, n);
  const int64_t g = tiles_for(v2 ? n / 2 : n, 1);
  if (g > 0x7fffffffLL) { set_error("vector too long for one launch"); return KHIP_ERR_INVALID; }
  KHIP_TRY(ensure_reduction_scratch(ctx, g * kWavesPerBlock, 2));
  RedArgs ra = make_red_args(ctx, slot);
#define KHIP_L(COMP, VEC, NT) \
  hipLaunchKernelGGL((bicg_xr_kernel<COMP, VEC, NT>), dim3((unsigned)g), dim3(kBlock), 0, ctx->stream, n, omega, st, s, t, z, c, x, r, ra)
#define KHIP_L2(COMP) do { if (v2) { if (nt) KHIP_L(COMP, 2, true); else KHIP_L(COMP, 2, false); } \
                           else    { if (nt) KHIP_L(COMP, 1, true); else KHIP_L(COMP, 1, false); } } while (0)
  if (comp) KHIP_L2(true); else KHIP_L2(false);
#undef KHIP_L2
#undef KHIP_L
  KHIP_CHECK_HIP(hipGetLastError());
  return launch_finish(ctx, g * kWavesPerBlock, 2, slot);
}

// Device-scalar variant for the device-resident CG loop: alpha, beta and the `solved` flag come from
// the CgDevState the epilogues maintain; when the stopping test has fired only x is updated
// (src/cg.jl:255-260 skips the direction update once solved).
template <int VEC, bool NT>
__global__ __launch_bounds__(kBlock) void cg_update_dev_kernel(int64_t n, const CgDevState *st, long long seq,
                                                               const double *r, double *p, double *x) {
  using T = typename VecT<VEC>::type;
  if (seq >= st->stop_seq) return;
  const double a = st->alpha, b = st->beta;
  const bool solved = st->solved != 0;
  const int64_t nvec = n / VEC;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < nvec) {
    const T pv = ldg<NT>(reinterpret_cast<T *>(p) + i);
    const T xv = ldg<NT>(reinterpret_cast<T *>(x) + i);
    T rv = {};
    if (!solved) rv = ldg<NT>(reinterpret_cast<const T *>(r) + i);
    T xo, po;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      vset(xo, e, fma(a, vget(pv, e), vget(xv, e)));
      vset(po, e, fma(1.0, vget(rv, e), b * vget(pv, e)));
    }
    stg<NT>(xo, reinterpret_cast<T *>(x) + i);
    if (!solved) stg<NT>(po, reinterpret_cast<T *>(p) + i);   // (a cacheable store of p for the next SpMV's gather was measured: slower)
  }
  if (VEC == 2 && (n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t t = n - 1;
    const double pv = p[t];
    x[t] = fma(a, pv, x[t]);
    if (!solved) p[t] = fma(1.0, r[t], b * pv);
  }
}

int launch_cg_update_dev(khip_ctx *ctx, int64_t n, const void *cg_state_dev, long long seq, const double *r, double *p,
                         double *x) {
  if (n <= 0) return KHIP_OK;
  const CgDevState *st = static_cast<const CgDevState *>(cg_state_dev);
  const bool v2 = n >= 2 && aligned16(r) && aligned16(p) && aligned16(x);
  const bool nt = use_nt(ctx, n);
  const int64_t g = tiles_for(v2 ? n / 2 : n, 1);
  if (g > 0x7fffffffLL) { set_error("vector too long for one launch"); return KHIP_ERR_INVALID; }
#define KHIP_CGU(VEC, NT) \
  hipLaunchKernelGGL((cg_update_dev_kernel<VEC, NT>), dim3((unsigned)g), dim3(kBlock), 0, ctx->stream, n, st, seq, r, p, x)
  if (v2) { if (nt) KHIP_CGU(2, true); else KHIP_CGU(2, false); }
  else    { if (nt) KHIP_CGU(1, true); else KHIP_CGU(1, false); }
#undef KHIP_CGU
  KHIP_CHECK_HIP(hipGetLastError());
  return KHIP_OK;
}

// Scalar epilogue on its own (the reduction value was written to results[slot] by other means).
__global__ void epilogue_kernel(RedArgs ra) {
  if (seq_skip(ra.stop_seq, ra.seq)) return;
  if (ra.epi) solver_epilogue(ra.epi, ra.epi_state, ra.results + ra.slot, ra.seq);
}

int launch_epilogue_only(khip_ctx *ctx, int slot) {
  RedArgs ra = make_red_args(ctx, slot);
  hipLaunchKernelGGL(epilogue_kernel, dim3(1), dim3(1), 0, ctx->stream, ra);
  KHIP_CHECK_HIP(hipGetLastError());
  return KHIP_OK;
}

// Cross-rank fold of all-gathered (hi, lo) partials, gathered[rank][count], in rank order with TwoSum --
// the same operations as the host loop of comm_allreduce_dd, so every rank gets the bit-identical scalar.
__global__ void combine_kernel(const dd *gathered, int nranks, int count, RedArgs ra) {
  if (seq_skip(ra.stop_seq, ra.seq)) return;
  const int i = threadIdx.x;
  if (i < count) {
    double hi = 0.0, lo = 0.0;
    for (int r = 0; r < nranks; ++r) {
      const dd v = gathered[(size_t)r * count + i];
      double s, e;
      two_sum(hi, v.hi, s, e);
      hi = s;
      lo += v.lo + e;
    }
    ra.results[ra.slot + i] = hi + lo;
  }
  __syncthreads();
  if (threadIdx.x == 0 && ra.epi) solver_epilogue(ra.epi, ra.epi_state, ra.results + ra.slot, ra.seq);
}

int launch_combine(khip_ctx *ctx, const dd *gathered_dev, int nranks, int count, int slot, hipStream_t stream) {
  if (count > 64) { set_error("combine: too many scalars"); return KHIP_ERR_INVALID; }
  RedArgs ra = make_red_args(ctx, slot);
  hipLaunchKernelGGL(combine_kernel, dim3(1), dim3(64), 0, stream ? stream : ctx->stream, gathered_dev, nranks, count, ra);
  KHIP_CHECK_HIP(hipGetLastError());
  return KHIP_OK;
}

// ---------------------------------------------------------------- reductions ----
enum RedOp {
  RED_DOT = 0,        // out0 = x . y
  RED_SQ = 1,         // out0 = x . x                 (x read once)
  RED_DOT2 = 2,       // out0 = x . y ; out1 = x . x
  RED_AXPY2 = 3,      // X += a p ; R -= a q ; out0 = R . R
  RED_AXPYDEV = 4,    // Y -= (*coef) x ; out0 = z . Y   (z == Y -> ||Y||^2)
  RED_AXPYSQ = 5,     // Y += a x ; out0 = Y . Y
  RED_CGSETUP = 6,    // U = x ; V = x ; W = 0 ; out0 = x . x      (cg! setup in one pass: r = b, p = b, x = 0, gamma = b . b)
};

template <int ROP> struct RedOut { static constexpr int n = (ROP == RED_DOT2) ? 2 : 1; };

struct RedPtrs {
  const double *x;      // DOT: x | SQ: x | DOT2: x | AXPY2: p | AXPYDEV: x
  const double *y;      // DOT: y |       | DOT2: y | AXPY2: q | AXPYDEV: z
  double *u;            //                           AXPY2: X | AXPYDEV: Y | AXPYSQ: Y
  double *v;            //                           AXPY2: R
  const double *coef;   // AXPYDEV: device scalar
  double a;             // AXPY2
  double *w;            // CGSETUP: W
};

// KEEP: the y / u streams use ordinary (cacheable) accesses even when NT is set for x -- the MGS cascade
// re-reads q (u) and the basis vector it just dotted (y becomes the next step's x) in the very next kernel.
template <int ROP, bool COMP, int VEC, bool NT, int U, bool KEEP = false>
__global__ __launch_bounds__(kBlock) void reduce_kernel(int64_t n, RedPtrs p, RedArgs ra) {
  constexpr bool NTK = NT && !KEEP;
  using T = typename VecT<VEC>::type;
  constexpr int NOUT = RedOut<ROP>::n;
  if (seq_skip(ra.stop_seq, ra.seq)) return;
  dd acc[NOUT];
#pragma unroll
  for (int o = 0; o < NOUT; ++o) acc[o] = dd{0.0, 0.0};
  const int64_t nvec = n / VEC;
  const int64_t base = (int64_t)blockIdx.x * (kBlock * U) + threadIdx.x;
  const T *X = reinterpret_cast<const T *>(p.x);
  const T *Y = reinterpret_cast<const T *>(p.y);
  T *Uv = reinterpret_cast<T *>(p.u);
  T *Vv = reinterpret_cast<T *>(p.v);
  double a = p.a;
  if (ROP == RED_AXPYDEV) a = -(*p.coef);
  const bool z_is_y = (ROP == RED_AXPYSQ) || ((ROP == RED_AXPYDEV) && (p.y == p.u));
  constexpr bool rd_y = (ROP == RED_DOT || ROP == RED_DOT2 || ROP == RED_AXPY2 || ROP == RED_AXPYDEV);
  constexpr bool rd_u = (ROP == RED_AXPY2 || ROP == RED_AXPYDEV || ROP == RED_AXPYSQ);
  constexpr bool rd_v = (ROP == RED_AXPY2);
  T *Wv = reinterpret_cast<T *>(p.w);

  T xv[U] = {}, yv[U] = {}, uv[U] = {}, vv[U] = {};
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t j = base + u * kBlock;
    if (j < nvec) {
      xv[u] = ldg<NT>(X + j);
      if (rd_y && !(ROP == RED_AXPYDEV && z_is_y)) yv[u] = ldg<NTK>(Y + j);
      if (rd_u) uv[u] = ldg<NTK>(Uv + j);
      if (rd_v) vv[u] = ldg<NT>(Vv + j);
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t j = base + u * kBlock;
    if (j < nvec) {
      if (ROP == RED_AXPY2) {
        T un, vn;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          vset(un, e, fma(a, vget(xv[u], e), vget(uv[u], e)));
          double rn = fma(-a, vget(yv[u], e), vget(vv[u], e));
          vset(vn, e, rn);
          acc_prod<COMP>(acc[0], rn, rn);
        }
        stg<NTK>(un, Uv + j);
        stg<NT>(vn, Vv + j);
      } else if (ROP == RED_AXPYDEV || ROP == RED_AXPYSQ) {
        T un;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          double yn = fma(a, vget(xv[u], e), vget(uv[u], e));
          vset(un, e, yn);
          acc_prod<COMP>(acc[0], z_is_y ? yn : vget(yv[u], e), yn);
        }
        stg<NTK>(un, Uv + j);
      } else if (ROP == RED_CGSETUP) {
        T zero;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const double xe = vget(xv[u], e);
          vset(zero, e, 0.0);
          acc_prod<COMP>(acc[0], xe, xe);
        }
        stg<NT>(xv[u], Uv + j);
        stg<NT>(xv[u], Vv + j);
        stg<NT>(zero, Wv + j);
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          double xe = vget(xv[u], e);
          if (ROP == RED_DOT) acc_prod<COMP>(acc[0], xe, vget(yv[u], e));
          if (ROP == RED_SQ) acc_prod<COMP>(acc[0], xe, xe);
          if (ROP == RED_DOT2) {
            acc_prod<COMP>(acc[0], xe, vget(yv[u], e));
            acc_prod<COMP>(acc[NOUT - 1], xe, xe);
          }
        }
      }
    }
  }
  if (VEC == 2 && (n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {   // odd tail element, scalar
    const int64_t t = n - 1;
    double xe = p.x[t];
    if (ROP == RED_DOT) acc_prod<COMP>(acc[0], xe, p.y[t]);
    if (ROP == RED_SQ) acc_prod<COMP>(acc[0], xe, xe);
    if (ROP == RED_DOT2) { acc_prod<COMP>(acc[0], xe, p.y[t]); acc_prod<COMP>(acc[NOUT - 1], xe, xe); }
    if (ROP == RED_AXPY2) {
      p.u[t] = fma(a, xe, p.u[t]);
      double rn = fma(-a, p.y[t], p.v[t]);
      p.v[t] = rn;
      acc_prod<COMP>(acc[0], rn, rn);
    }
    if (ROP == RED_CGSETUP) { p.u[t] = xe; p.v[t] = xe; p.w[t] = 0.0; acc_prod<COMP>(acc[0], xe, xe); }
    if (ROP == RED_AXPYDEV || ROP == RED_AXPYSQ) {
      double yn = fma(a, xe, p.u[t]);
      double ze = z_is_y ? yn : p.y[t];
      p.u[t] = yn;
      acc_prod<COMP>(acc[0], ze, yn);
    }
  }
  wave_publish<NOUT>(acc, ra);
}

template <int ROP>
static int launch_reduce(khip_ctx *ctx, int64_t n, const RedPtrs &p, int slot) {
  if (n < 0) { set_error("negative length"); return KHIP_ERR_INVALID; }
  const bool v2 = n >= 2 && aligned16(p.x) && (p.y == nullptr || aligned16(p.y)) && (p.u == nullptr || aligned16(p.u)) &&
                  (p.v == nullptr || aligned16(p.v)) && (p.w == nullptr || aligned16(p.w));
  const bool comp = ctx->tune.compensated != 0;
  // MGS cascade (AXPYDEV): q and the basis vector just dotted are read again by the very next kernel; when both
  // fit in the 256 MiB Infinity Cache, leaving them cacheable beats streaming them (GMRES(30) at 256^3:
  // 1.91 -> 1.71 ms per inner iteration; at 384^3 it costs 2-5 %: profiles/r01h_mgs_keep.log)
  int keep = ctx->tune.mgs_keep;
  if (keep < 0) keep = (ROP == RED_AXPYDEV && (size_t)n * sizeof(double) <= (size_t)144 << 20) ? 1 : 0;
  const bool nt = use_nt(ctx, n) && !(ROP == RED_AXPYDEV && keep == 2);
  const int64_t nvec = v2 ? n / 2 : n;
  // 16-byte accesses per lane: 4 for the read-only reductions of long vectors (dot 6.1 vs 5.8 TB/s, nrm2 6.1 vs 4.3),
  // 1 for the ones that also write (r -= a Ap ; r.r: 5.96 vs 5.72 TB/s) -- measured at n = 512^3, tools/archive/sweep_reduce_width.py
  constexpr bool writes = (ROP == RED_AXPY2 || ROP == RED_AXPYDEV || ROP == RED_AXPYSQ || ROP == RED_CGSETUP);
  const bool u4 = ctx->tune.red_u == 0 ? (!writes && nvec >= (int64_t)kBlock * 4 * 1024) : ctx->tune.red_u == 4;
  const int64_t g = tiles_for(nvec, u4 ? 4 : 1);
  if (g > 0x7fffffffLL) { set_error("vector too long for one launch"); return KHIP_ERR_INVALID; }
  KHIP_TRY(ensure_reduction_scratch(ctx, g * kWavesPerBlock, RedOut<ROP>::n));
  RedArgs ra = make_red_args(ctx, slot);
#define KHIP_RED(COMP, VEC, NT, U) \
  hipLaunchKernelGGL((reduce_kernel<ROP, COMP, VEC, NT, U>), dim3((unsigned)g), dim3(kBlock), 0, ctx->stream, n, p, ra)
#define KHIP_RED_U(COMP, VEC, NT) do { if (u4) KHIP_RED(COMP, VEC, NT, 4); else KHIP_RED(COMP, VEC, NT, 1); } while (0)
#define KHIP_RED_NT(COMP, VEC) do { if (nt) KHIP_RED_U(COMP, VEC, true); else KHIP_RED_U(COMP, VEC, false); } while (0)
  if (ROP == RED_AXPYDEV && nt && v2 && comp && keep == 1) {
    if (u4) hipLaunchKernelGGL((reduce_kernel<ROP, true, 2, true, 4, true>), dim3((unsigned)g), dim3(kBlock), 0, ctx->stream, n, p, ra);
    else hipLaunchKernelGGL((reduce_kernel<ROP, true, 2, true, 1, true>), dim3((unsigned)g), dim3(kBlock), 0, ctx->stream, n, p, ra);
  } else if (comp) { if (v2) KHIP_RED_NT(true, 2); else KHIP_RED_NT(true, 1); }
  else             { if (v2) KHIP_RED_NT(false, 2); else KHIP_RED_NT(false, 1); }
#undef KHIP_RED_NT
#undef KHIP_RED_U
#undef KHIP_RED
  KHIP_CHECK_HIP(hipGetLastError());
  return launch_finish(ctx, g * kWavesPerBlock, RedOut<ROP>::n, slot);
}

int launch_dot(khip_ctx *ctx, int64_t n, const double *x, const double *y, int slot) {
  if (x == y) return launch_nrm2sq(ctx, n, x, slot);     // aliased dot reads the vector once (src/cg.jl:242, z === r)
  RedPtrs p{x, y, nullptr, nullptr, nullptr, 0.0};
  return launch_reduce<RED_DOT>(ctx, n, p, slot);
}
int launch_nrm2sq(khip_ctx *ctx, int64_t n, const double *x, int slot) {
  RedPtrs p{x, nullptr, nullptr, nullptr, nullptr, 0.0};
  return launch_reduce<RED_SQ>(ctx, n, p, slot);
}
int launch_dot2(khip_ctx *ctx, int64_t n, const double *x, const double *y, int slot) {
  RedPtrs p{x, y, nullptr, nullptr, nullptr, 0.0};
  return launch_reduce<RED_DOT2>(ctx, n, p, slot);
}
int launch_axpy2_dot(khip_ctx *ctx, int64_t n, double a, const double *pv, const double *q, double *x, double *r,
                     int slot) {
  RedPtrs p{pv, q, x, r, nullptr, a};
  return launch_reduce<RED_AXPY2>(ctx, n, p, slot);
}
int launch_axpy_sqnorm(khip_ctx *ctx, int64_t n, double a, const double *x, double *y, int slot) {
  RedPtrs p{x, nullptr, y, nullptr, nullptr, a};
  return launch_reduce<RED_AXPYSQ>(ctx, n, p, slot);
}
// cg! setup (src/cg.jl:153-162 with M = I and no warm start) in ONE pass: x = 0, r = b, p = b, gamma = b . b -- reads 8n and
// writes 24n bytes instead of the 48n of kfill! + 2 kcopy! + kdotr; every value is what the four primitives leave.
int launch_cg_setup(khip_ctx *ctx, int64_t n, const double *b, double *x, double *r, double *pvec, int slot) {
  RedPtrs p{b, nullptr, r, pvec, nullptr, 0.0, x};
  return launch_reduce<RED_CGSETUP>(ctx, n, p, slot);
}
int launch_axpy_dev_dot(khip_ctx *ctx, int64_t n, const double *coef_dev, const double *x, double *y, const double *z,
                        int slot) {
  RedPtrs p{x, z, y, nullptr, coef_dev, 0.0};
  return launch_reduce<RED_AXPYDEV>(ctx, n, p, slot);
}

// ------------------------------------------------------------ multi-axpy -------
constexpr int kMultiMax = 32;
struct MultiArgs {
  const double *v[kMultiMax];
  double c[kMultiMax];
};

template <int VEC, bool NT>
__global__ __launch_bounds__(kBlock) void multi_axpy_kernel(int64_t n, int k, MultiArgs ma, double *x) {
  using T = typename VecT<VEC>::type;
  const int64_t nvec = n / VEC;
  T *X = reinterpret_cast<T *>(x);
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < nvec) {
    T xv = ldg<NT>(X + i);
    int j = 0;
    for (; j + 4 <= k; j += 4) {
      T v0 = ldg<NT>(reinterpret_cast<const T *>(ma.v[j]) + i);
      T v1 = ldg<NT>(reinterpret_cast<const T *>(ma.v[j + 1]) + i);
      T v2 = ldg<NT>(reinterpret_cast<const T *>(ma.v[j + 2]) + i);
      T v3 = ldg<NT>(reinterpret_cast<const T *>(ma.v[j + 3]) + i);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        double s = vget(xv, e);
        s = fma(ma.c[j], vget(v0, e), s);
        s = fma(ma.c[j + 1], vget(v1, e), s);
        s = fma(ma.c[j + 2], vget(v2, e), s);
        s = fma(ma.c[j + 3], vget(v3, e), s);
        vset(xv, e, s);
      }
    }
    for (; j < k; ++j) {
      T v0 = ldg<NT>(reinterpret_cast<const T *>(ma.v[j]) + i);
#pragma unroll
      for (int e = 0; e < VEC; ++e) vset(xv, e, fma(ma.c[j], vget(v0, e), vget(xv, e)));
    }
    stg<NT>(xv, X + i);
  }
  if (VEC == 2 && (n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t t = n - 1;
    double s = x[t];
    for (int j = 0; j < k; ++j) s = fma(ma.c[j], ma.v[j][t], s);
    x[t] = s;
  }
}

// ------------------------------------------------------------ block (classical) Gram-Schmidt pieces -------
// gmres! options.variant = 1 (CGS2, solvers.cpp): h = V_k' q as ONE reduction pass per four basis vectors and
// q -= V_k h as one pass with the coefficients read from device memory -- two all-reduces of k scalars per pass on N GPUs
// instead of the k one-scalar all-reduces of the modified Gram-Schmidt cascade (src/gmres.jl:259-271).
struct Dot4Ptrs { const double *x; const double *y[4]; int cnt; };

template <bool COMP, int VEC, bool NT>
__global__ __launch_bounds__(kBlock) void multi_dot4_kernel(int64_t n, Dot4Ptrs p, RedArgs ra) {
  using T = typename VecT<VEC>::type;
  if (seq_skip(ra.stop_seq, ra.seq)) return;
  dd acc[4];
#pragma unroll
  for (int o = 0; o < 4; ++o) acc[o] = dd{0.0, 0.0};
  const int64_t nvec = n / VEC;
  const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (j < nvec) {
    const T xv = ldg<NT>(reinterpret_cast<const T *>(p.x) + j);
    T yv[4];
#pragma unroll
    for (int o = 0; o < 4; ++o)
      if (o < p.cnt) yv[o] = ldg<NT>(reinterpret_cast<const T *>(p.y[o]) + j);
#pragma unroll
    for (int o = 0; o < 4; ++o)
      if (o < p.cnt) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc_prod<COMP>(acc[o], vget(yv[o], e), vget(xv, e));
      }
  }
  if (VEC == 2 && (n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t t = n - 1;
    for (int o = 0; o < p.cnt; ++o) acc_prod<COMP>(acc[o], p.y[o][t], p.x[t]);
  }
  wave_publish<4>(acc, ra);
}

// results[slot + i] = V_i . q for i < k (k <= kResultSlots), four per launch; the caller all-reduces and reads them
int launch_multi_dot(khip_ctx *ctx, int64_t n, int k, const double *const *V_host, const double *q, int slot) {
  if (k <= 0) return KHIP_OK;
  const bool comp = ctx->tune.compensated != 0;
  const bool nt = use_nt(ctx, n);
  for (int base = 0; base < k; base += 4) {
    Dot4Ptrs p;
    p.x = q;
    p.cnt = k - base < 4 ? k - base : 4;
    bool v2 = n >= 2 && aligned16(q);
    for (int o = 0; o < 4; ++o) {
      p.y[o] = o < p.cnt ? V_host[base + o] : q;
      if (!aligned16(p.y[o])) v2 = false;
    }
    const int64_t g = tiles_for(v2 ? n / 2 : n, 1);
    KHIP_TRY(ensure_reduction_scratch(ctx, g * kWavesPerBlock, 4));
    RedArgs ra = make_red_args(ctx, slot + base);
#define KHIP_MD(COMP, VEC, NT) hipLaunchKernelGGL((multi_dot4_kernel<COMP, VEC, NT>), dim3((unsigned)g), dim3(kBlock), 0, ctx->stream, n, p, ra)
    if (comp) { if (v2) { if (nt) KHIP_MD(true, 2, true); else KHIP_MD(true, 2, false); } else { if (nt) KHIP_MD(true, 1, true); else KHIP_MD(true, 1, false); } }
    else      { if (v2) { if (nt) KHIP_MD(false, 2, true); else KHIP_MD(false, 2, false); } else { if (nt) KHIP_MD(false, 1, true); else KHIP_MD(false, 1, false); } }
#undef KHIP_MD
    KHIP_CHECK_HIP(hipGetLastError());
    // a finish of 4 outputs writes slot + base .. slot + base + 3; the unused ones of the last group are scratch slots
    KHIP_TRY(launch_finish(ctx, g * kWavesPerBlock, 4, slot + base));
  }
  return KHIP_OK;
}

struct MultiPtrs { const double *v[kMultiMax]; };

// x <- x - sum_j coef[j] V_j with the coefficients in device memory (results ring), applied in the order j = 0..k-1
template <int VEC, bool NT>
__global__ __launch_bounds__(kBlock) void multi_axpy_dev_kernel(int64_t n, int k, MultiPtrs mv, const double *coef, double *x) {
  using T = typename VecT<VEC>::type;
  const int64_t nvec = n / VEC;
  T *X = reinterpret_cast<T *>(x);
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < nvec) {
    T xv = ldg<NT>(X + i);
    int j = 0;
    for (; j + 4 <= k; j += 4) {                       // four basis vectors in flight; applied in the order j, j+1, j+2, j+3
      const T v0 = ldg<NT>(reinterpret_cast<const T *>(mv.v[j]) + i);
      const T v1 = ldg<NT>(reinterpret_cast<const T *>(mv.v[j + 1]) + i);
      const T v2 = ldg<NT>(reinterpret_cast<const T *>(mv.v[j + 2]) + i);
      const T v3 = ldg<NT>(reinterpret_cast<const T *>(mv.v[j + 3]) + i);
      const double c0 = -coef[j], c1 = -coef[j + 1], c2 = -coef[j + 2], c3 = -coef[j + 3];
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        double sacc = vget(xv, e);
        sacc = fma(c0, vget(v0, e), sacc);
        sacc = fma(c1, vget(v1, e), sacc);
        sacc = fma(c2, vget(v2, e), sacc);
        sacc = fma(c3, vget(v3, e), sacc);
        vset(xv, e, sacc);
      }
    }
    for (; j < k; ++j) {
      const double cj = -coef[j];
      const T v0 = ldg<NT>(reinterpret_cast<const T *>(mv.v[j]) + i);
#pragma unroll
      for (int e = 0; e < VEC; ++e) vset(xv, e, fma(cj, vget(v0, e), vget(xv, e)));
    }
    stg<NT>(xv, X + i);
  }
  if (VEC == 2 && (n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t t = n - 1;
    double sacc = x[t];
    for (int j = 0; j < k; ++j) sacc = fma(-coef[j], mv.v[j][t], sacc);
    x[t] = sacc;
  }
}

int launch_multi_axpy_dev(khip_ctx *ctx, int64_t n, int k, const double *coef_dev, const double *const *V_host, double *x) {
  if (n <= 0 || k <= 0) return KHIP_OK;
  const bool nt = use_nt(ctx, n);
  for (int base = 0; base < k; base += kMultiMax) {
    const int kk = k - base < kMultiMax ? k - base : kMultiMax;
    MultiPtrs mv;
    bool v2 = n >= 2 && aligned16(x);
    for (int j = 0; j < kMultiMax; ++j) {
      mv.v[j] = j < kk ? V_host[base + j] : nullptr;
      if (j < kk && !aligned16(mv.v[j])) v2 = false;
    }
    const int64_t g = tiles_for(v2 ? n / 2 : n, 1);
    if (v2) {
      if (nt) hipLaunchKernelGGL((multi_axpy_dev_kernel<2, true>), dim3((unsigned)g), dim3(kBlock), 0, ctx->stream, n, kk, mv, coef_dev + base, x);
      else hipLaunchKernelGGL((multi_axpy_dev_kernel<2, false>), dim3((unsigned)g), dim3(kBlock), 0, ctx->stream, n, kk, mv, coef_dev + base, x);
    } else {
      if (nt) hipLaunchKernelGGL((multi_axpy_dev_kernel<1, true>), dim3((unsigned)g), dim3(kBlock), 0, ctx->stream, n, kk, mv, coef_dev + base, x);
      else hipLaunchKernelGGL((multi_axpy_dev_kernel<1, false>), dim3((unsigned)g), dim3(kBlock), 0, ctx->stream, n, kk, mv, coef_dev + base, x);
    }
    KHIP_CHECK_HIP(hipGetLastError());
  }
  return KHIP_OK;
}

int launch_multi_axpy(khip_ctx *ctx, int64_t n, int k, const double *coef_host, const double *const *V_host,
                      double *x) {
  if (n <= 0 || k <= 0) return KHIP_OK;
  const bool nt = use_nt(ctx, n);
  for (int base = 0; base < k; base += kMultiMax) {
    const int kk = k - base < kMultiMax ? k - base : kMultiMax;
    MultiArgs ma;
    bool v2 = n >= 2 && aligned16(x);
    for (int j = 0; j < kMultiMax; ++j) {
      ma.v[j] = j < kk ? V_host[base + j] : nullptr;
      ma.c[j] = j < kk ? coef_host[base + j] : 0.0;
      if (j < kk && !aligned16(ma.v[j])) v2 = false;
    }
    const int64_t g = tiles_for(v2 ? n / 2 : n, 1);
    if (v2) {
      if (nt) hipLaunchKernelGGL((multi_axpy_kernel<2, true>), dim3((unsigned)g), dim3(kBlock), 0, ctx->stream, n, kk, ma, x);
      else hipLaunchKernelGGL((multi_axpy_kernel<2, false>), dim3((unsigned)g), dim3(kBlock), 0, ctx->stream, n, kk, ma, x);
    } else {
      if (nt) hipLaunchKernelGGL((multi_axpy_kernel<1, true>), dim3((unsigned)g), dim3(kBlock), 0, ctx->stream, n, kk, ma, x);
      else hipLaunchKernelGGL((multi_axpy_kernel<1, false>), dim3((unsigned)g), dim3(kBlock), 0, ctx->stream, n, kk, ma, x);
    }
    KHIP_CHECK_HIP(hipGetLastError());
  }
  return KHIP_OK;
}

// ------------------------------------------------------------ scratch / results ---
int ensure_reduction_scratch(khip_ctx *ctx, int64_t nwaves, int nout) {
  if (nout > kMaxNout) { set_error("too many reduction outputs"); return KHIP_ERR_INVALID; }
  if (!ctx->partials2) {                                   // fixed-size parts, allocated once
    KHIP_CHECK_HIP(hipMalloc(&ctx->partials2, sizeof(dd) * (size_t)kMaxNout * kFinishMaxBlocks));
    KHIP_CHECK_HIP(hipMalloc(&ctx->tickets, sizeof(unsigned) * 32));
    KHIP_CHECK_HIP(hipMemset(ctx->tickets, 0, sizeof(unsigned) * 32));
    ctx->scratch_word = reinterpret_cast<int *>(ctx->tickets + 16);
  }
  if (nwaves <= ctx->red_cap1) return KHIP_OK;
  int64_t cap1 = 1 << 18;
  while (cap1 < nwaves) cap1 <<= 1;
  dd *fresh = nullptr;
  KHIP_CHECK_HIP(hipMalloc(&fresh, sizeof(dd) * (size_t)kMaxNout * (size_t)cap1));
  if (ctx->partials) {
    // growth in the middle of a multi-launch reduction (split distributed SpMV): keep what the earlier
    // launches have published -- stream-ordered copy, then retire the old buffer
    for (int o = 0; o < kMaxNout; ++o)
      KHIP_CHECK_HIP(hipMemcpyAsync(fresh + (size_t)o * cap1, ctx->partials + (size_t)o * ctx->red_cap1,
                                    sizeof(dd) * (size_t)ctx->red_cap1, hipMemcpyDeviceToDevice, ctx->stream));
    KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    KHIP_CHECK_HIP(hipFree(ctx->partials));
  }
  ctx->partials = fresh;
  ctx->red_cap1 = cap1;
  return KHIP_OK;
}

// fold `nwaves` per-wave partials of the kernel just launched on ctx->stream into results[slot..]
int launch_finish(khip_ctx *ctx, int64_t nwaves, int nout, int slot) {
  RedArgs ra = make_red_args(ctx, slot);
  if (ctx->comm) ra.epi = EPI_NONE;      // the cross-rank combine kernel owns the epilogue
  static const int per_thread = [] { const char *e = getenv("KHIP_FINISH_PER_THREAD"); const int v = e ? atoi(e) : 8; return v >= 1 && v <= 64 ? v : 8; }();   // partials a thread folds (experiments: 4 / 8 / 16)
  int64_t want = (nwaves + (int64_t)kBlock * per_thread - 1) / ((int64_t)kBlock * per_thread);
  const unsigned g = (unsigned)(want < 1 ? 1 : (want > kFinishMaxBlocks ? kFinishMaxBlocks : want));
  if (nout == 1) hipLaunchKernelGGL((reduce_finish_kernel<1>), dim3(g), dim3(kBlock), 0, ctx->stream, ra, nwaves);
  else if (nout == 2) hipLaunchKernelGGL((reduce_finish_kernel<2>), dim3(g), dim3(kBlock), 0, ctx->stream, ra, nwaves);
  else if (nout == 4) hipLaunchKernelGGL((reduce_finish_kernel<4>), dim3(g), dim3(kBlock), 0, ctx->stream, ra, nwaves);
  else { set_error("launch_finish: unsupported output count %d", nout); return KHIP_ERR_INVALID; }
  KHIP_CHECK_HIP(hipGetLastError());
  return KHIP_OK;
}

int fetch_results(khip_ctx *ctx, int slot, int count, double *out_host, bool already_global) {
  if (ctx->comm && !already_global) return comm_allreduce_dd(ctx, ctx->results_dd + slot, count, out_host);
  KHIP_CHECK_HIP(hipMemcpyAsync(ctx->results_pinned, ctx->results + slot, sizeof(double) * (size_t)count,
                                hipMemcpyDeviceToHost, ctx->stream));
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < count; ++i) out_host[i] = ctx->results_pinned[i];
  return KHIP_OK;
}

// Split form of fetch_results for look-ahead: `begin` enqueues the copy of results[slot..] (already all-reduced if
// there are several ranks) and records an event; work enqueued afterwards does not delay `end`, which waits for
// the event only.
int results_copy_begin(khip_ctx *ctx, int slot, int count) {
  if (!ctx->ev_fetch) KHIP_CHECK_HIP(hipEventCreateWithFlags(&ctx->ev_fetch, hipEventDisableTiming));
  KHIP_CHECK_HIP(hipMemcpyAsync(ctx->results_pinned, ctx->results + slot, sizeof(double) * (size_t)count,
                                hipMemcpyDeviceToHost, ctx->stream));
  KHIP_CHECK_HIP(hipEventRecord(ctx->ev_fetch, ctx->stream));
  return KHIP_OK;
}
int results_copy_end(khip_ctx *ctx, int count, double *out_host) {
  KHIP_CHECK_HIP(hipEventSynchronize(ctx->ev_fetch));
  for (int i = 0; i < count; ++i) out_host[i] = ctx->results_pinned[i];
  return KHIP_OK;
}

// y = x / sqrt(*sumsq): kdivcopy!(n, V[k+1], q, Hbis) (src/gmres.jl:325) with Hbis = ||q|| still on the device
template <int VEC, bool NT>
__global__ __launch_bounds__(kBlock) void divcopy_dev_kernel(int64_t n, const double *sumsq, const double *x, double *y) {
  using T = typename VecT<VEC>::type;
  const double s = sqrt(*sumsq);
  const int64_t nvec = n / VEC;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < nvec) {
    const T xv = ldg<NT>(reinterpret_cast<const T *>(x) + i);
    T yo;
#pragma unroll
    for (int e = 0; e < VEC; ++e) vset(yo, e, vget(xv, e) / s);
    stg<NT>(yo, reinterpret_cast<T *>(y) + i);
  }
  if (VEC == 2 && (n & 1) && blockIdx.x == 0 && threadIdx.x == 0) y[n - 1] = x[n - 1] / s;
}

int launch_divcopy_dev(khip_ctx *ctx, int64_t n, const double *sumsq_dev, const double *x, double *y) {
  if (n <= 0) return KHIP_OK;
  const bool v2 = n >= 2 && aligned16(x) && aligned16(y);
  const bool nt = use_nt(ctx, n);
  const int64_t g = tiles_for(v2 ? n / 2 : n, 1);
  if (g > 0x7fffffffLL) { set_error("vector too long for one launch"); return KHIP_ERR_INVALID; }
#define KHIP_L(VEC, NT) \
  hipLaunchKernelGGL((divcopy_dev_kernel<VEC, NT>), dim3((unsigned)g), dim3(kBlock), 0, ctx->stream, n, sumsq_dev, x, y)
  if (v2) { if (nt) KHIP_L(2, true); else KHIP_L(2, false); }
  else    { if (nt) KHIP_L(1, true); else KHIP_L(1, false); }
#undef KHIP_L
  KHIP_CHECK_HIP(hipGetLastError());
  return KHIP_OK;
}

}  // namespace khip
