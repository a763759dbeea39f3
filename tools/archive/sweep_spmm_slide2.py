#!/usr/bin/env python3
"""Sliding-window tile SpMM: run length and grid size at 216^3 x 16 (27-point)."""
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
kind = sys.argv[1] if len(sys.argv) > 1 else "stencil27"
ref = None
for slide in (0, 1, 54, 27, 14, 6, 3):
    ctx.set_option("spmm_tile_slide", slide)
    A = K.CsrMatrix.stencil(ctx, kind, 216)
    n = A.shape[0]
    X = K.Panel.from_host(ctx, np.cos(np.arange(n * p) * 1e-3).reshape(n, p) + 0.5)
    Y = K.Panel(ctx, n, p)
    for grid in (0, 1792, 1536, 2048):
        ctx.set_option("spmm_tile_grid", grid)
        ms = run(A, X, Y)
        h = Y.to_host()
        if ref is None: ref = h
        print(json.dumps(dict(kind=kind, slide=slide, grid=grid, ms=round(ms, 4), groups=A.tile_info["groups"], same=bool(np.array_equal(h, ref)))), flush=True)
    ctx.set_option("spmm_tile_grid", 0)
    del A, X, Y
ctx.close()
