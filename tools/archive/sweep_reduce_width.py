#!/usr/bin/env python3
"""BLAS-1 reductions at n = 512^3: accesses per lane (red_u) for dot, axpy_sqnorm (r -= a Ap ; r.r), axpy2_dot."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import krylov_jl_amd as K
ctx = K.Context(0)
n = 512 ** 3
x, y, z, w = (ctx.empty(n) for _ in range(4))
for v in (x, y, z, w): K.kfill_(v, 1.0)
def timeit(fn, reps=20):
    fn(); ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.sync(); return (time.perf_counter() - t0) / reps
for rep in range(2):
    for u in (0, 1, 4):
        ctx.set_option("red_u", u)
        t1 = timeit(lambda: K.kdot(n, x, y))
        t2 = timeit(lambda: K.axpy_sqnorm(n, 1e-9, x, y))
        t3 = timeit(lambda: K.axpy2_dot(n, 1e-9, x, y, z, w))
        t4 = timeit(lambda: K.knorm(n, x))
        print(json.dumps({"red_u": u, "dot_gbps": round(16 * n / t1 / 1e9), "axpy_sqnorm_gbps": round(24 * n / t2 / 1e9),
                          "axpy2_dot_gbps": round(48 * n / t3 / 1e9), "nrm2_gbps": round(8 * n / t4 / 1e9)}), flush=True)
ctx.close()
