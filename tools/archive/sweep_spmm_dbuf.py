#!/usr/bin/env python3
"""Tile SpMM with double-buffered windows (spmm_tile_dbuf, round 4) at p = 16: the copies of group g + 1 land while the products
of g run.  27-point 216^3 (cfg 5), 7-point 216^3, banded + random; dbuf on / off x persistent-wave counts.  Y must equal the
single-window result bit for bit.  Prints JSON lines."""
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
small = "--small" in sys.argv
OPS = [("27-point 216^3", lambda: K.CsrMatrix.stencil(ctx, "stencil27", 40 if small else 216)),
       ("7-point 216^3", lambda: K.CsrMatrix.stencil(ctx, "poisson", 40 if small else 216)),
       ("banded+random", lambda: K.CsrMatrix.banded_random(ctx, (1 << 17) if small else 10 * (1 << 20), seed=1))]
ctx.set_option("spmm_tile_slide", 0)
for name, make in OPS:
    A = make()
    n = A.shape[0]
    X = K.Panel.from_host(ctx, np.cos(np.arange(n * p) * 1e-3).reshape(n, p) + 0.5)
    Y = K.Panel(ctx, n, p)
    ref = None
    for dbuf, grids in ((0, (0,)), (1, (0, 768, 1024, 1280)), (0, (0,)), (1, (0,))):
        ctx.set_option("spmm_tile_dbuf", dbuf)
        for grid in grids:
            ctx.set_option("spmm_tile_grid", grid)
            ms = run(A, X, Y)
            h = Y.to_host()
            if ref is None: ref = h
            alg = 12 * A.nnz + 4 * n + 16 * n * p
            print(json.dumps(dict(operator=name, dbuf=dbuf, grid=grid, ms=round(ms, 4), frac=round(alg / (ms * 1e-3) / 8e12, 4),
                                  window=A.tile_info["window"], same=bool(np.array_equal(h, ref)))), flush=True)
        ctx.set_option("spmm_tile_grid", 0)
    del A, X, Y
ctx.set_option("spmm_tile_dbuf", 0)
ctx.close()
