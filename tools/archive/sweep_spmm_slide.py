#!/usr/bin/env python3
"""Tile SpMM with sliding windows (spmm_tile_slide, round 4) at p = 16: every group copies only the panel rows its wave's window
does not hold yet.  Operators: 27-point 216^3 (cfg 5), 7-point 216^3, banded + random 10.5 M rows; tile shapes 4x4x2 and
8x4x1; slide on / off.  Y must equal the slide = 0 result bit for bit.  Prints JSON lines."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
reps = 10
p = 16
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
for name, make in OPS:
    ref = None
    for shape in ((0, 4, 3) if "banded" not in name else (0,)):
        for slide in (0, 1) + ((32, 128) if "banded" in name else ()):
            ctx.set_option("spmm_tile_shape", shape); ctx.set_option("spmm_tile_slide", slide)
            A = make()
            n = A.shape[0]
            X, Y = K.Panel(ctx, n, p), K.Panel(ctx, n, p)
            Xh = np.cos(np.arange(n * p) * 1e-3).reshape(n, p) + 0.5
            X = K.Panel.from_host(ctx, Xh)
            ms = run(A, X, Y)
            h = Y.to_host()
            if ref is None: ref = h
            alg = 12 * A.nnz + 4 * n + 16 * n * p
            print(json.dumps(dict(operator=name, shape=shape, slide=slide, ms=round(ms, 4), frac=round(alg / (ms * 1e-3) / 8e12, 4),
                                  tile_info=A.tile_info, same=bool(np.array_equal(h, ref)))), flush=True)
            del A, X, Y
ctx.set_option("spmm_tile_shape", 0); ctx.set_option("spmm_tile_slide", 1)
ctx.close()
