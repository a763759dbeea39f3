#!/usr/bin/env python3
"""Phase switch-off experiments of the tile SpMM at cfg 5 (27-point 216^3, p = 16): spmm_tile_exp bits 1 = no panel-row copies,
2 = no products, 4 = no (val, slot) loads (all give WRONG results; timing only), and combinations -- what the kernel's time is made of."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
A = K.CsrMatrix.stencil(ctx, "stencil27", 216)
p = 16
X, Y = K.Panel(ctx, A.n, p), K.Panel(ctx, A.n, p)


def timed(reps=10):
    K.spmm_(A, X, Y); ctx.sync(); best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps): K.spmm_(A, X, Y)
        ctx.sync(); best = min(best, (time.perf_counter() - t0) / reps)
    return best * 1e3


for exp, what in ((0, "everything"), (2, "no products"), (1, "no panel-row copies"), (4, "no (val, slot) loads"), (3, "no copies, no products"), (6, "no products, no entry loads"),
                  (5, "no copies, no entry loads"), (7, "records and Y only"), (0, "everything")):
    ctx.set_option("spmm_tile_exp", exp)
    print(json.dumps(dict(exp=exp, what=what, ms=round(timed(), 4))), flush=True)
ctx.set_option("spmm_tile_exp", 0)
ctx.close()
