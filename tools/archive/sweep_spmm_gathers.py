#!/usr/bin/env python3
"""SpMM p = 16 on the 27-point 216^3 operator: plane-sweep tile order (spmm_sweep) variants; checks Y is unchanged."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 216
p = 16
n = n1 ** 3
A = K.CsrMatrix.stencil(ctx, "stencil27", n1)
rng = np.random.default_rng(0)
X, Y, Y0 = K.Panel(ctx, n, p), K.Panel(ctx, n, p), K.Panel(ctx, n, p)
K.kfill_(X.buf, 1.0)
K.kscal_(n * p, 1.0, X.buf)
bytes_alg = 12 * A.nnz + 4 * n + 16 * n * p
def timeit(reps=5):
    K.spmm_(A, X, Y); ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps): K.spmm_(A, X, Y)
    ctx.sync(); return (time.perf_counter() - t0) / reps
ctx.set_option("spmm_sweep", 0)
ctx.set_option("spmm_wide", 0)
K.spmm_(A, X, Y0); ctx.sync()
y0 = Y0.to_host()
def case(**o):
    for k, v in o.items(): ctx.set_option(k, v)
    t = timeit()
    same = bool(np.array_equal(Y.to_host(), y0))
    print(json.dumps(dict(o, ms=round(t * 1e3, 3), gbps=round(bytes_alg / t / 1e9), same=same)), flush=True)
ctx.set_option("spmm_wide", 0)
case(spmm_wide=0, spmm_sweep=0)
case(spmm_wide=1, spmm_sweep=0)
case(spmm_wide=0, spmm_sweep=1, spmm_sweep_s=n1 * n1 // 16, spmm_sweep_w=64)
case(spmm_wide=1, spmm_sweep=0)
case(spmm_wide=0, spmm_sweep=0)
ctx.close()
