#!/usr/bin/env python3
"""Two waves per window x sliding windows (run lengths) at cfg 5 (27-point 216^3, p = 16) and on the 7-point grid."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
reps, p = 10, 16
def run(A, X, Y):
    K.spmm_(A, X, Y); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps): K.spmm_(A, X, Y)
    ctx.sync()
    return (time.perf_counter() - t0) / reps * 1e3
ctx.set_option("spmm_tile_dbuf", 0)
for kind in ("stencil27", "poisson"):
    ref = None
    for pair, slide in ((0, 0), (1, 0), (1, 1), (1, 54), (1, 27), (1, 14), (1, 6), (0, 14), (0, 0), (1, 0)):
        ctx.set_option("spmm_tile_pair", pair); ctx.set_option("spmm_tile_slide", slide)
        A = K.CsrMatrix.stencil(ctx, kind, 216)
        n = A.shape[0]
        X = K.Panel.from_host(ctx, np.cos(np.arange(n * p) * 1e-3).reshape(n, p) + 0.5)
        Y = K.Panel(ctx, n, p)
        ms = run(A, X, Y)
        h = Y.to_host()
        if ref is None: ref = h
        alg = 12 * A.nnz + 4 * n + 16 * n * p
        print(json.dumps(dict(kind=kind, pair=pair, slide=slide, ms=round(ms, 4), frac=round(alg / (ms * 1e-3) / 8e12, 4), same=bool(np.array_equal(h, ref)))), flush=True)
        del A, X, Y
ctx.close()
