#!/usr/bin/env python3
"""Same n and nnz/row, different grid aspect ratios: does the x re-fetch distance set the SpMV time?"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import krylov_jl_amd as K
ctx = K.Context(0)
def timeit(fn, reps=10):
    fn(); ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.sync(); return (time.perf_counter() - t0) / reps
for dims in ((512, 512, 512), (128, 128, 8192), (64, 64, 32768), (2048, 256, 256), (134217728 // 4, 2, 2)):
    n = dims[0] * dims[1] * dims[2]
    A = K.CsrMatrix.stencil(ctx, "poisson", *dims)
    x, y = ctx.empty(n), ctx.empty(n)
    K.kfill_(x, 1.0)
    sb = A.spmv_bytes
    for kern, xcd in ((1, 0), (4, 0), (4, 16), (1, 16)):
        ctx.set_option("spmv_kernel", kern); ctx.set_option("spmv_tiles", 1); ctx.set_option("spmv_xcd", xcd)
        t = timeit(lambda: A.matvec(x, y))
        print(json.dumps(dict(dims=dims, nnz=A.nnz, kernel=kern, xcd=xcd, ms=round(t * 1e3, 4), gbps=round(sb / t / 1e9), frac=round(sb / t / 8e12, 4))), flush=True)
    del A, x, y
ctx.close()
