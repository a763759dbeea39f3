#!/usr/bin/env python3
"""Tile SpMM (spmm_tile.hip) at 216^3 x 16: group order (pencil width), phase switch-off experiments (WRONG results by
design: spmm_tile_exp), wave counts.  Env: PENCILS="2,1,3,4,8,54" EXPS="0,8,16,6,22,2,1,7" SHAPE=0"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 216
reps = 10
p = 16
def run(A, X, Y):
    K.spmm_(A, X, Y); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps): K.spmm_(A, X, Y)
    ctx.sync()
    return (time.perf_counter() - t0) / reps * 1e3
pencils = [int(v) for v in os.environ.get("PENCILS", "2,1,3,4,8,54").split(",")]
exps = [int(v) for v in os.environ.get("EXPS", "0,8,16,6,22,2,1,7").split(",")]
ctx.set_option("spmm_tile_shape", int(os.environ.get("SHAPE", "0")))
for i, pen in enumerate(pencils):
    ctx.set_option("spmm_tile_pencil", pen)
    A = K.CsrMatrix.stencil(ctx, "stencil27", n1)
    X, Y = K.Panel(ctx, A.n, p), K.Panel(ctx, A.n, p)
    K.kfill_(X.buf, 1.0)
    for exp in (exps if i == 0 else (0, 8)):
        ctx.set_option("spmm_tile_exp", exp)
        for nt in (0,):
            ctx.set_option("spmm_tile_nt", nt)
            print(json.dumps(dict(pencil=pen, exp=exp, nt=nt, ms=round(run(A, X, Y), 4), window=A.tile_info["window"])), flush=True)
    ctx.set_option("spmm_tile_nt", 0)
    ctx.set_option("spmm_tile_exp", 0)
    if True:
        for grid in (1280, 1536, 1792):
            ctx.set_option("spmm_tile_grid", grid)
            print(json.dumps(dict(pencil=pen, grid=grid, ms=round(run(A, X, Y), 4))), flush=True)
        ctx.set_option("spmm_tile_grid", 0)
    del A, X, Y
ctx.close()
