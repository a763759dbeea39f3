#!/usr/bin/env python3
"""Pair / sliding defaults: banded + random at p = 16; the 27-point operator at p = 32 (one launch vs 16-column slices) and p = 8."""
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
def case(name, make, p, opts):
    for k, v in opts.items(): ctx.set_option(k, v)
    A = make()
    n = A.shape[0]
    X = K.Panel.from_host(ctx, np.cos(np.arange(n * p) * 1e-3).reshape(n, p) + 0.5)
    Y = K.Panel(ctx, n, p)
    ms = run(A, X, Y)
    alg = 12 * A.nnz + 4 * n + 16 * n * p
    print(json.dumps(dict(operator=name, p=p, opts=opts, ms=round(ms, 4), frac=round(alg / (ms * 1e-3) / 8e12, 4))), flush=True)
br = lambda: K.CsrMatrix.banded_random(ctx, 10 * (1 << 20), seed=1)
s27 = lambda: K.CsrMatrix.stencil(ctx, "stencil27", 216)
base = dict(spmm_tile_dbuf=0, spmm_tile=1, spmm_tile_slices=0)
for pair, slide in ((0, 0), (1, 0), (1, 32), (1, 64), (1, 16), (0, 0)):
    case("banded+random", br, 16, dict(base, spmm_tile_pair=pair, spmm_tile_slide=slide))
for pair, slide, slices in ((0, 0, 0), (1, 0, 0), (1, 27, 0), (1, 0, -1), (1, 27, -1), (0, 0, -1)):
    case("27-point", s27, 32, dict(base, spmm_tile_pair=pair, spmm_tile_slide=slide, spmm_tile_slices=slices))
for dbuf in (0, 1):
    case("27-point", s27, 8, dict(base, spmm_tile_pair=0, spmm_tile_slide=0, spmm_tile_dbuf=dbuf))
ctx.close()
