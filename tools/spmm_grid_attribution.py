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
A = K.CsrMatrix.stencil(ctx, "stencil27", 216)
X, Y = K.Panel(ctx, A.n, p), K.Panel(ctx, A.n, p)
ctx.set_option("spmm_tile", 2); ctx.set_option("spmm_window", 1)
for grid in (0, 1280, 1464, 1536, 1672, 1792, 0):
    ctx.set_option("spmm_tile_grid", grid)
    print(json.dumps(dict(lib=os.path.basename(os.path.dirname(os.environ.get("KHIP_LIBRARY", "x/library/l"))), grid=grid, ms=round(timed(A, X, Y) * 1e3, 4))), flush=True)
