#!/usr/bin/env python3
"""Tile SpMM with two waves per window (spmm_tile_pair, round 4) at p = 16 and 32: twice the waves per CU on the same LDS.
27-point 216^3 (cfg 5), 7-point 216^3, banded + random; pair on / off x workgroup counts.  Y must equal the one-wave result
bit for bit.  Prints JSON lines."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
reps = 10
def run(A, X, Y):
    K.spmm_(A, X, Y); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps): K.spmm_(A, X, Y)
    ctx.sync()
    return (time.perf_counter() - t0) / reps * 1e3
small = "--small" in sys.argv
OPS = [("27-point 216^3", lambda: K.CsrMatrix.stencil(ctx, "stencil27", 40 if small else 216)),
       ("7-point 216^3", lambda: K.CsrMatrix.stencil(ctx, "poisson", 40 if small else 216)),
       ("banded+random", lambda: K.CsrMatrix.banded_random(ctx, (1 << 17) if small else 10 * (1 << 20), seed=1))]
ctx.set_option("spmm_tile_slide", 0); ctx.set_option("spmm_tile_dbuf", 0); ctx.set_option("spmm_tile", 2)
for p in (16, 32):
    for name, make in OPS:
        if p == 32 and "27" not in name: continue
        A = make()
        n = A.shape[0]
        X = K.Panel.from_host(ctx, np.cos(np.arange(n * p) * 1e-3).reshape(n, p) + 0.5)
        Y = K.Panel(ctx, n, p)
        ctx.set_option("spmm_tile_slices", -1 if p == 32 else 0)
        ref = None
        for pair, grids in ((0, (0,)), (1, (0, 1280, 1536, 2048)), (0, (0,)), (1, (0,))):
            ctx.set_option("spmm_tile_pair", pair)
            for grid in grids:
                ctx.set_option("spmm_tile_grid", grid)
                ms = run(A, X, Y)
                h = Y.to_host()
                if ref is None: ref = h
                alg = 12 * A.nnz + 4 * n + 16 * n * p
                print(json.dumps(dict(operator=name, p=p, pair=pair, grid=grid, ms=round(ms, 4), frac=round(alg / (ms * 1e-3) / 8e12, 4),
                                      window=A.tile_info["window"], same=bool(np.array_equal(h, ref)))), flush=True)
            ctx.set_option("spmm_tile_grid", 0)
        del A, X, Y
ctx.close()
