#!/usr/bin/env python3
"""Tile SpMM, sliding windows with and without one group of look-ahead (ctx option spmm_tile_ahead, round 6): timing and bit-equality
with the direct-gather kernel on the cfg-5 operator (27-point 216^3, p = 16), the 7-point grid and the banded + random operator.
One JSON line per case.  Usage: python tools/spmm_ahead_ab.py [--quick] [--sweep]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K

quick = "--quick" in sys.argv
sweep = "--sweep" in sys.argv
ctx = K.Context(0)
reps = 10
DEFAULTS = {"spmm_tile_slide": -1, "spmm_tile_pair": 1, "spmm_tile_shape": 0, "spmm_tile_pencil": 0, "spmm_tile_ahead": 0, "spmm_tile_grid": 0}


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


# (the option's default is 0 since the measurements of round 6: profiles/r06c_spmm_ahead_ab.jsonl, r06d_spmm_chunk_ab.log)
cases = [("stencil27", 16, {}), ("stencil27", 16, {"spmm_tile_ahead": 1}), ("stencil27", 16, {}), ("stencil27", 16, {"spmm_tile_ahead": 1})]
if sweep:
    cases += [("stencil27", 16, {"spmm_tile_grid": 256 * 4}), ("stencil27", 16, {"spmm_tile_grid": 256 * 5}), ("stencil27", 16, {"spmm_tile_grid": 256 * 6}),
              ("stencil27", 16, {"spmm_tile_grid": 256 * 7}), ("stencil27", 16, {"spmm_tile_grid": 256 * 8}),
              ("stencil27", 16, {"spmm_tile_slide": 54}), ("stencil27", 16, {"spmm_tile_slide": 14}), ("stencil27", 16, {"spmm_tile_slide": 0}),
              ("stencil27", 16, {"spmm_tile_shape": 4}), ("stencil27", 16, {"spmm_tile_shape": 4, "spmm_tile_ahead": 1}),
              ("stencil27", 16, {"spmm_tile_ahead": 1, "spmm_tile_grid": 256 * 5})]
if not quick:
    cases += [("poisson", 16, {}), ("poisson", 16, {"spmm_tile_ahead": 1}), ("banded", 16, {}), ("stencil27", 32, {}), ("stencil27", 8, {})]
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
    print(json.dumps(dict(op=kind, p=p, opts=opts, ms=round(t_tile * 1e3, 4), frac=round(alg / t_tile / 8e12, 4), bit_identical_to_direct=same,
                          window=info["window"], direct_groups=info["direct_groups"], groups=info["groups"])), flush=True)
    for k in opts:
        ctx.set_option(k, DEFAULTS[k])
    del A, X, Y
ctx.close()
