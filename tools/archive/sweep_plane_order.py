#!/usr/bin/env python3
"""SpMV at 512^3: plane-sweep tile order (spmv_xcd = -2) for the staged CSR kernel."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import krylov_jl_amd as K
ctx = K.Context(0)
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = n1 ** 3
A = K.CsrMatrix.stencil(ctx, "poisson", n1)
x, y, y0 = ctx.empty(n), ctx.empty(n), ctx.empty(n)
import numpy as np
K.kfill_(x, 1.0)
sb = A.spmv_bytes
def timeit(fn, reps=20):
    fn(); ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.sync(); return (time.perf_counter() - t0) / reps
A.matvec(x, y0)
def case(**o):
    for k, v in o.items(): ctx.set_option(k, v)
    t = timeit(lambda: A.matvec(x, y)); t2 = timeit(lambda: K.spmv_dot(A, x, y))
    K.kaxpy_(n, -1.0, y0, y)
    print(json.dumps(dict(o, ms=round(t * 1e3, 4), gbps=round(sb / t / 1e9), frac=round(sb / t / 8e12, 4), ms_dot=round(t2 * 1e3, 4), diff=K.knorm(n, y))), flush=True)
S = n1 * n1 // 256
case(spmv_xcd=0)
for w in (4, 8, 16, 32, 64):
    case(spmv_xcd=-2, spmv_sweep_s=S, spmv_sweep_w=w)
case(spmv_xcd=0)
ctx.close()
