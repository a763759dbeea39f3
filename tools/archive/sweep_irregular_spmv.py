#!/usr/bin/env python3
"""SpMV kernels on the banded + random operator (10 M rows, ~26.7 entries per row): which of the bit-exact kernels the
int32 path should pick for mid-length irregular rows.  Prints JSON lines."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10 * (1 << 20)
A = K.CsrMatrix.banded_random(ctx, n, seed=1)
x = ctx.array(np.cos(np.arange(n) * 1e-3) + 0.5)
y = ctx.zeros(n)
alg = A.spmv_bytes
def run(reps=10):
    A.matvec(x, y); ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps): A.matvec(x, y)
    ctx.sync(); return (time.perf_counter() - t0) / reps
ref = None
for opts in ([dict(spmv_kernel=3)] + [dict(spmv_kernel=3, spmv_lanes=l) for l in (8, 16, 32)] +
             [dict(spmv_kernel=4, spmv_rows=r, spmv_cap=c) for r in (256, 128, 64) for c in (2048, 1024)] +
             [dict(spmv_kernel=1, spmv_rows=r) for r in (256, 64)] + [dict(spmv_kernel=2)]):
    for k, v in opts.items(): ctx.set_option(k, v)
    try:
        t = run()
        h = y.to_host()
        if ref is None: ref = h
        print(json.dumps(dict(opts=opts, ms=round(t * 1e3, 4), alg_tbps=round(alg / t / 1e12, 3), frac=round(alg / t / 8e12, 4), same=bool(np.array_equal(h, ref)))), flush=True)
    except Exception as e:
        print(json.dumps(dict(opts=opts, error=str(e)[:200])), flush=True)
    for k in opts: ctx.set_option(k, dict(spmv_kernel=0, spmv_rows=256, spmv_cap=0, spmv_lanes=0)[k])
ctx.close()
