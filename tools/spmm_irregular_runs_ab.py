import json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
def timed(A, X, Y, reps=10):
    K.spmm_(A, X, Y); ctx.sync(); best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps): K.spmm_(A, X, Y)
        ctx.sync(); best = min(best, (time.perf_counter() - t0) / reps)
    return best
p = 16
D = {"spmm_tile_slide": -1, "spmm_tile_ahead": 0, "spmm_tile_xcd": -1, "spmm_tile_grid": 0}
for opts in ({}, {"spmm_tile_slide": 27, "spmm_tile_xcd": 1}, {"spmm_tile_slide": 27, "spmm_tile_xcd": 1, "spmm_tile_ahead": 1}, {"spmm_tile_slide": 8, "spmm_tile_xcd": 1, "spmm_tile_ahead": 1},
             {"spmm_tile_slide": 27, "spmm_tile_xcd": 2, "spmm_tile_ahead": 1}, {"spmm_tile_slide": 64, "spmm_tile_xcd": 1, "spmm_tile_ahead": 1}, {}):
    for k, v in opts.items(): ctx.set_option(k, v)
    ctx.set_option("spmm_tile", 2); ctx.set_option("spmm_window", 1)
    A = K.CsrMatrix.banded_random(ctx, 10 * (1 << 20), seed=1)
    X, Y = K.Panel(ctx, A.n, p), K.Panel(ctx, A.n, p)
    h = np.zeros((K.panel_rows(A.n), p)); h[:A.n] = np.random.default_rng(0).standard_normal((A.n, p)); X.buf.copy_from_host(h.ravel())
    t = timed(A, X, Y); y = Y.buf.to_host()
    ctx.set_option("spmm_tile", 0); ctx.set_option("spmm_window", 0); K.spmm_(A, X, Y); ctx.sync()
    print(json.dumps(dict(op="banded + 3 links", opts=opts, ms=round(t * 1e3, 4), window=A.tile_info["window"], same=bool(np.array_equal(y, Y.buf.to_host())))), flush=True)
    for k in opts: ctx.set_option(k, D[k])
    del A, X, Y
