// blas1.hip -- the BLAS-1 shim of Krylov.jl (src/krylov_utils.jl:309-349) as gfx950 kernels.
//
// Every kernel here is HBM-bound.  Layout: plain contiguous f64 vectors.  Each lane moves
// 16 B per access (double2, the coalescing sweet spot on CDNA4), four independent accesses
// in flight per array per lane, grid capped at 8 workgroups per CU with a grid-stride loop.
// Algorithmic bytes per element: dot 16 (8 if x === y), nrm2 8, axpy/axpby 24, copy/divcopy/
// scalcopy 16, fill 8, scal 16, reflect 32; fused: axpy2_dot 48, axpy_dev_dot 24(+8), waxpy 24.
#include "device_reduce.hpp"

namespace khip {

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
};

template <int VEC> struct VecT;
template <> struct VecT<1> { using type = double; };
template <> struct VecT<2> { using type = double2; };

__device__ __forceinline__ double vget(const double &v, int) { return v; }
__device__ __forceinline__ double vget(const double2 &v, int i) { return i == 0 ? v.x : v.y; }
__device__ __forceinline__ void vset(double &v, int, double s) { v = s; }
__device__ __forceinline__ void vset(double2 &v, int i, double s) { if (i == 0) v.x = s; else v.y = s; }

template <int OP> __host__ __device__ constexpr bool reads_x() {
  return OP == OP_COPY || OP == OP_SCALCOPY || OP == OP_DIVCOPY || OP == OP_AXPY || OP == OP_AXPBY ||
         OP == OP_REF || OP == OP_WAXPY;
}
template <int OP> __host__ __device__ constexpr bool reads_y() {
  return OP == OP_SCAL || OP == OP_AXPY || OP == OP_AXPBY || OP == OP_REF || OP == OP_WAXPY;
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
}

// x, y, w deliberately NOT __restrict__: exact aliasing is legal (BLAS semantics, src/bicgstab.jl:153-157).
template <int OP, int VEC>
__global__ __launch_bounds__(kBlock) void map_kernel(int64_t n, double a, double b, const double *x, double *y,
                                                     double *w) {
  using T = typename VecT<VEC>::type;
  constexpr int U = 4;
  const int64_t nvec = n / VEC;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const T *X = reinterpret_cast<const T *>(x);
  T *Y = reinterpret_cast<T *>(y);
  T *W = reinterpret_cast<T *>(OP == OP_WAXPY ? w : y);
  T *XO = reinterpret_cast<T *>(const_cast<double *>(x));
  for (; i + (U - 1) * stride < nvec; i += U * stride) {
    T xv[U] = {}, yv[U] = {};
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (reads_x<OP>()) xv[u] = X[i + u * stride];
      if (reads_y<OP>()) yv[u] = Y[i + u * stride];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      T ox, oy;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        double sx, sy;
        map_scalar<OP>(a, b, reads_x<OP>() ? vget(xv[u], e) : 0.0, reads_y<OP>() ? vget(yv[u], e) : 0.0, sx, sy);
        vset(ox, e, sx);
        vset(oy, e, sy);
      }
      W[i + u * stride] = oy;
      if (OP == OP_REF) XO[i + u * stride] = ox;
    }
  }
  for (; i < nvec; i += stride) {
    T xv = {}, yv = {}, ox, oy;
    if (reads_x<OP>()) xv = X[i];
    if (reads_y<OP>()) yv = Y[i];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      double sx, sy;
      map_scalar<OP>(a, b, reads_x<OP>() ? vget(xv, e) : 0.0, reads_y<OP>() ? vget(yv, e) : 0.0, sx, sy);
      vset(ox, e, sx);
      vset(oy, e, sy);
    }
    W[i] = oy;
    if (OP == OP_REF) XO[i] = ox;
  }
  if (VEC == 2 && (n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {   // odd tail element
    const int64_t t = n - 1;
    double sx, sy;
    map_scalar<OP>(a, b, reads_x<OP>() ? x[t] : 0.0, reads_y<OP>() ? y[t] : 0.0, sx, sy);
    (OP == OP_WAXPY ? w : y)[t] = sy;
    if (OP == OP_REF) const_cast<double *>(x)[t] = sx;
  }
}

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static inline int grid_for(khip_ctx *ctx, int64_t nvec, int unroll) {
  int64_t want = (nvec + (int64_t)kBlock * unroll - 1) / ((int64_t)kBlock * unroll);
  int64_t cap = ctx->tune.blas1_blocks;
  if (cap > kMaxRedBlocks) cap = kMaxRedBlocks;
  if (want < 1) want = 1;
  return (int)(want < cap ? want : cap);
}

template <int OP>
static int launch_map_op(khip_ctx *ctx, int64_t n, double a, double b, const double *x, double *y, double *w) {
  if (n <= 0) return KHIP_OK;
  bool v2 = (!reads_x<OP>() || aligned16(x)) && aligned16(y) && (OP != OP_WAXPY || aligned16(w)) && n >= 2;
  if (v2) {
    int g = grid_for(ctx, n / 2, 4);
    hipLaunchKernelGGL((map_kernel<OP, 2>), dim3(g), dim3(kBlock), 0, ctx->stream, n, a, b, x, y, w);
  } else {
    int g = grid_for(ctx, n, 4);
    hipLaunchKernelGGL((map_kernel<OP, 1>), dim3(g), dim3(kBlock), 0, ctx->stream, n, a, b, x, y, w);
  }
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
    default: set_error("launch_map: unknown op %d", op); return KHIP_ERR_INVALID;
  }
}

// ---------------------------------------------------------------- reductions ----
enum RedOp {
  RED_DOT = 0,        // out0 = x . y
  RED_SQ = 1,         // out0 = x . x                 (x read once)
  RED_DOT2 = 2,       // out0 = x . y ; out1 = x . x
  RED_AXPY2 = 3,      // X += a p ; R -= a q ; out0 = R . R
  RED_AXPYDEV = 4,    // Y -= (*coef) x ; out0 = z . Y   (z == Y -> ||Y||^2)
};

template <int ROP> struct RedOut { static constexpr int n = (ROP == RED_DOT2) ? 2 : 1; };

struct RedPtrs {
  const double *x;      // DOT: x | SQ: x | DOT2: x | AXPY2: p | AXPYDEV: x
  const double *y;      // DOT: y |       | DOT2: y | AXPY2: q | AXPYDEV: z
  double *u;            //                           AXPY2: X | AXPYDEV: Y
  double *v;            //                           AXPY2: R
  const double *coef;   // AXPYDEV: device scalar
  double a;             // AXPY2
};

template <int ROP, bool COMP, int VEC>
__global__ __launch_bounds__(kBlock) void reduce_kernel(int64_t n, RedPtrs p, RedArgs ra) {
  using T = typename VecT<VEC>::type;
  constexpr int NOUT = RedOut<ROP>::n;
  constexpr int U = 4;
  dd acc[NOUT];
#pragma unroll
  for (int o = 0; o < NOUT; ++o) acc[o] = dd{0.0, 0.0};
  const int64_t nvec = n / VEC;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const T *X = reinterpret_cast<const T *>(p.x);
  const T *Y = reinterpret_cast<const T *>(p.y);
  T *Uv = reinterpret_cast<T *>(p.u);
  T *Vv = reinterpret_cast<T *>(p.v);
  double a = p.a;
  if (ROP == RED_AXPYDEV) a = -(*p.coef);
  const bool z_is_y = (ROP == RED_AXPYDEV) && (p.y == p.u);

  auto body = [&](T xv, T yv, T uv, T vv, int64_t idx) {
    if (ROP == RED_AXPY2) {
      T un, vn;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        vset(un, e, fma(a, vget(xv, e), vget(uv, e)));
        double rn = fma(-a, vget(yv, e), vget(vv, e));
        vset(vn, e, rn);
        acc_prod<COMP>(acc[0], rn, rn);
      }
      Uv[idx] = un;
      Vv[idx] = vn;
    } else if (ROP == RED_AXPYDEV) {
      T un;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        double yn = fma(a, vget(xv, e), vget(uv, e));
        vset(un, e, yn);
        acc_prod<COMP>(acc[0], z_is_y ? yn : vget(yv, e), yn);
      }
      Uv[idx] = un;
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        double xe = vget(xv, e);
        if (ROP == RED_DOT) acc_prod<COMP>(acc[0], xe, vget(yv, e));
        if (ROP == RED_SQ) acc_prod<COMP>(acc[0], xe, xe);
        if (ROP == RED_DOT2) {
          acc_prod<COMP>(acc[0], xe, vget(yv, e));
          acc_prod<COMP>(acc[NOUT - 1], xe, xe);
        }
      }
    }
  };
  constexpr bool rd_y = (ROP == RED_DOT || ROP == RED_DOT2 || ROP == RED_AXPY2 || ROP == RED_AXPYDEV);
  constexpr bool rd_u = (ROP == RED_AXPY2 || ROP == RED_AXPYDEV);
  constexpr bool rd_v = (ROP == RED_AXPY2);

  for (; i + (U - 1) * stride < nvec; i += U * stride) {
    T xv[U] = {}, yv[U] = {}, uv[U] = {}, vv[U] = {};
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t j = i + u * stride;
      xv[u] = X[j];
      if (rd_y && !(ROP == RED_AXPYDEV && z_is_y)) yv[u] = Y[j];
      if (rd_u) uv[u] = Uv[j];
      if (rd_v) vv[u] = Vv[j];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) body(xv[u], yv[u], uv[u], vv[u], i + u * stride);
  }
  for (; i < nvec; i += stride) {
    T xv = X[i], yv = {}, uv = {}, vv = {};
    if (rd_y && !(ROP == RED_AXPYDEV && z_is_y)) yv = Y[i];
    if (rd_u) uv = Uv[i];
    if (rd_v) vv = Vv[i];
    body(xv, yv, uv, vv, i);
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
    if (ROP == RED_AXPYDEV) {
      double yn = fma(a, xe, p.u[t]);
      double ze = z_is_y ? yn : p.y[t];
      p.u[t] = yn;
      acc_prod<COMP>(acc[0], ze, yn);
    }
  }
  grid_finish<NOUT>(acc, ra);
}

template <int ROP>
static int launch_reduce(khip_ctx *ctx, int64_t n, const RedPtrs &p, int slot) {
  if (n < 0) { set_error("negative length"); return KHIP_ERR_INVALID; }
  bool v2 = n >= 2 && aligned16(p.x) && (p.y == nullptr || aligned16(p.y)) && (p.u == nullptr || aligned16(p.u)) &&
            (p.v == nullptr || aligned16(p.v));
  RedArgs ra = make_red_args(ctx, slot);
  const bool comp = ctx->tune.compensated != 0;
  int g = grid_for(ctx, v2 ? n / 2 : n, 4);
#define KHIP_LAUNCH_RED(COMP, VEC) \
  hipLaunchKernelGGL((reduce_kernel<ROP, COMP, VEC>), dim3(g), dim3(kBlock), 0, ctx->stream, n, p, ra)
  if (comp) { if (v2) KHIP_LAUNCH_RED(true, 2); else KHIP_LAUNCH_RED(true, 1); }
  else      { if (v2) KHIP_LAUNCH_RED(false, 2); else KHIP_LAUNCH_RED(false, 1); }
#undef KHIP_LAUNCH_RED
  KHIP_CHECK_HIP(hipGetLastError());
  return KHIP_OK;
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

template <int VEC>
__global__ __launch_bounds__(kBlock) void multi_axpy_kernel(int64_t n, int k, MultiArgs ma, double *x) {
  using T = typename VecT<VEC>::type;
  const int64_t nvec = n / VEC;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  T *X = reinterpret_cast<T *>(x);
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nvec; i += stride) {
    T xv = X[i];
    int j = 0;
    for (; j + 4 <= k; j += 4) {
      T v0 = reinterpret_cast<const T *>(ma.v[j])[i];
      T v1 = reinterpret_cast<const T *>(ma.v[j + 1])[i];
      T v2 = reinterpret_cast<const T *>(ma.v[j + 2])[i];
      T v3 = reinterpret_cast<const T *>(ma.v[j + 3])[i];
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
      T v0 = reinterpret_cast<const T *>(ma.v[j])[i];
#pragma unroll
      for (int e = 0; e < VEC; ++e) vset(xv, e, fma(ma.c[j], vget(v0, e), vget(xv, e)));
    }
    X[i] = xv;
  }
  if (VEC == 2 && (n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t t = n - 1;
    double s = x[t];
    for (int j = 0; j < k; ++j) s = fma(ma.c[j], ma.v[j][t], s);
    x[t] = s;
  }
}

int launch_multi_axpy(khip_ctx *ctx, int64_t n, int k, const double *coef_host, const double *const *V_host,
                      double *x) {
  if (n <= 0 || k <= 0) return KHIP_OK;
  for (int base = 0; base < k; base += kMultiMax) {
    int kk = k - base < kMultiMax ? k - base : kMultiMax;
    MultiArgs ma;
    bool v2 = n >= 2 && aligned16(x);
    for (int j = 0; j < kMultiMax; ++j) {
      ma.v[j] = j < kk ? V_host[base + j] : nullptr;
      ma.c[j] = j < kk ? coef_host[base + j] : 0.0;
      if (j < kk && !aligned16(ma.v[j])) v2 = false;
    }
    if (v2) {
      int g = grid_for(ctx, n / 2, 1);
      hipLaunchKernelGGL((multi_axpy_kernel<2>), dim3(g), dim3(kBlock), 0, ctx->stream, n, kk, ma, x);
    } else {
      int g = grid_for(ctx, n, 1);
      hipLaunchKernelGGL((multi_axpy_kernel<1>), dim3(g), dim3(kBlock), 0, ctx->stream, n, kk, ma, x);
    }
    KHIP_CHECK_HIP(hipGetLastError());
  }
  return KHIP_OK;
}

int fetch_results(khip_ctx *ctx, int slot, int count, double *out_host) {
  if (ctx->comm) return comm_allreduce_dd(ctx, ctx->results_dd + slot, count, out_host);
  KHIP_CHECK_HIP(hipMemcpyAsync(ctx->results_pinned, ctx->results + slot, sizeof(double) * (size_t)count,
                                hipMemcpyDeviceToHost, ctx->stream));
  KHIP_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < count; ++i) out_host[i] = ctx->results_pinned[i];
  return KHIP_OK;
}

}  // namespace khip
