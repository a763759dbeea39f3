#!/usr/bin/env python3
"""int32 staged SpMV at 512^3 (spmv_codes = 0): XCD-aware tile orders, rows per block, plain and fused with the dot."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
ctx.set_option("spmv_codes", int(os.environ.get("CODES", "0")))
n1 = 512
A = K.CsrMatrix.stencil(ctx, "poisson", n1)
n = A.n
x = ctx.empty(n); K.kfill_(x, 1.0)
y = ctx.zeros(n)
alg = A.spmv_bytes
def run(fn, reps=20):
    fn(); ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.sync(); return (time.perf_counter() - t0) / reps
for opts in ([dict()] + [dict(spmv_xcd=r) for r in (1, 2, 4, 8, 16, 32, -1)] + [dict(spmv_rows=128), dict(spmv_rows=64), dict(spmv_tiles=2), dict(spmv_tiles=4), dict(spmv_nty=1)] + [dict()]):
    for k, v in opts.items(): ctx.set_option(k, v)
    tp = run(lambda: A.matvec(x, y)); tf = run(lambda: K.spmv_dot(A, x, y))
    print(json.dumps(dict(opts=opts, plain_ms=round(tp * 1e3, 4), plain_frac=round(alg / tp / 8e12, 4), fused_ms=round(tf * 1e3, 4), fused_frac=round(alg / tf / 8e12, 4))), flush=True)
    for k in opts: ctx.set_option(k, dict(spmv_xcd=0, spmv_rows=256, spmv_tiles=1, spmv_nty=0)[k])
ctx.close()
