#!/usr/bin/env python3
"""Tile SpMM with 32-row groups (the library) against 64-row groups (a -DKHIP_TILE_ROWS=64 build of it, KHIP_LIBRARY=...): timing
and bit-equality with the direct-gather kernel on the cfg-5 operator (27-point 216^3), the 7-point grid and the banded + random
operator, p = 16 (and 8 / 32 on the 27-point operator).  The synthetic twin puts the floor of 64-row tiles 13 % below that of
32-row tiles (profiles/r05b_spmm_floor.log).  Usage: KHIP_LIBRARY=<.so> python tools/spmm_tile_rows_ab.py [--quick]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K

quick = "--quick" in sys.argv
ctx = K.Context(0)
reps = 10
tag = os.path.basename(os.environ.get("KHIP_LIBRARY", "libkrylov_hip.so"))


def timed(A, X, Y):
    K.spmm_(A, X, Y); ctx.sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps): K.spmm_(A, X, Y)
        ctx.sync()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best


def make(kind):
    if kind == "banded":
        return K.CsrMatrix.banded_random(ctx, 10 * (1 << 20), seed=1)
    return K.CsrMatrix.stencil(ctx, kind, 216)


cases = [("stencil27", 16, {}), ("stencil27", 16, {"spmm_tile_shape": 6}), ("stencil27", 16, {"spmm_tile_shape": 7}),
         ("stencil27", 16, {"spmm_tile_slide": 0}), ("stencil27", 16, {"spmm_tile_pair": 0}), ("stencil27", 16, {"spmm_tile_pencil": 2}),
         ("stencil27", 16, {"spmm_tile_pencil": 8})]
if not quick:
    cases += [("poisson", 16, {}), ("banded", 16, {}), ("stencil27", 8, {}), ("stencil27", 32, {})]
for kind, p, opts in cases:
    for k, v in opts.items():
        ctx.set_option(k, v)
    ctx.set_option("spmm_tile", 2); ctx.set_option("spmm_window", 1)
    A = make(kind)                                        # a fresh handle: the records are built under the options set above
    X, Y = K.Panel(ctx, A.n, p), K.Panel(ctx, A.n, p)
    h = np.zeros((K.panel_rows(A.n), p)); h[:A.n] = np.random.default_rng(0).standard_normal((A.n, p))
    X.buf.copy_from_host(h.ravel())
    t_tile = timed(A, X, Y)
    y_tile = Y.buf.to_host()
    info = A.tile_info
    ctx.set_option("spmm_tile", 0); ctx.set_option("spmm_window", 0)
    K.spmm_(A, X, Y); ctx.sync()
    same = bool(np.array_equal(y_tile, Y.buf.to_host()))
    alg = 12 * A.nnz + 4 * A.n + 16 * A.n * p
    print(json.dumps(dict(lib=tag, op=kind, p=p, opts=opts, ms=t_tile * 1e3, frac=alg / t_tile / 8e12, bit_identical_to_direct=same, tile_info=info)), flush=True)
    for k in opts:
        ctx.set_option(k, {"spmm_tile_slide": -1, "spmm_tile_pair": 1, "spmm_tile_shape": 0, "spmm_tile_pencil": 0}[k])
    del A, X, Y
ctx.close()
