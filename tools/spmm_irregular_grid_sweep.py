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
A = K.CsrMatrix.banded_random(ctx, 10 * (1 << 20), seed=1)
X, Y = K.Panel(ctx, A.n, p), K.Panel(ctx, A.n, p)
ctx.set_option("spmm_tile", 2); ctx.set_option("spmm_window", 1)
alg = 12 * A.nnz + 4 * A.n + 16 * A.n * p
for grid in (0, 512, 768, 1024, 1280, 1536, 1792, 2048):
    ctx.set_option("spmm_tile_grid", grid)
    t = timed(A, X, Y)
    print(json.dumps(dict(op="banded + 3 links", grid=grid, ms=round(t * 1e3, 4), frac=round(alg / t / 8e12, 4))), flush=True)
