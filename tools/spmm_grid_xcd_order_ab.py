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
for kind in ("stencil27", "poisson", "banded"):
    A = K.CsrMatrix.banded_random(ctx, 10 * (1 << 20), seed=1) if kind == "banded" else K.CsrMatrix.stencil(ctx, kind, 216)
    X, Y = K.Panel(ctx, A.n, p), K.Panel(ctx, A.n, p)
    h = np.zeros((K.panel_rows(A.n), p)); h[:A.n] = np.random.default_rng(0).standard_normal((A.n, p)); X.buf.copy_from_host(h.ravel())
    ctx.set_option("spmm_tile", 0); ctx.set_option("spmm_window", 0); K.spmm_(A, X, Y); ctx.sync(); ref = Y.buf.to_host()
    ctx.set_option("spmm_tile", 2); ctx.set_option("spmm_window", 1)
    for xo in (0, 1, 2, 0, 1, 2):
        ctx.set_option("spmm_tile_xcd", xo)
        t = timed(A, X, Y)
        print(json.dumps(dict(op=kind, xcd_order=xo, ms=round(t * 1e3, 4), same=bool(np.array_equal(ref, Y.buf.to_host())))), flush=True)
    ctx.set_option("spmm_tile_xcd", -1)
    del A, X, Y
