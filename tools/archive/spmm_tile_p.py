#!/usr/bin/env python3
"""Tile SpMM (spmm_tile.hip) at p = 8, 16, 32 vs the window / direct kernels: bit-equality and timing on the 27-point and
7-point grid operators and on the banded + random operator.  Usage: python tools/archive/spmm_tile_p.py [n1] [reps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K

ctx = K.Context(0)
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 216
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10


def setopts(tile, window, slices):
    ctx.set_option("spmm_tile", tile); ctx.set_option("spmm_window", window); ctx.set_option("spmm_tile_slices", slices)


def timed(A, X, Y, tile, window, slices):
    setopts(tile, window, slices)
    for _ in range(3 * reps): K.spmm_(A, X, Y)      # the first few dozen launches of a process run at ramping clocks
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps): K.spmm_(A, X, Y)
    ctx.sync()
    return (time.perf_counter() - t0) / reps


def operators():
    yield "stencil27", K.CsrMatrix.stencil(ctx, "stencil27", n1, n1, n1)
    yield "poisson", K.CsrMatrix.stencil(ctx, "poisson", n1, n1, n1)
    yield "banded_random", K.CsrMatrix.banded_random(ctx, n1 ** 3, 13, 3, seed=1, dense_rows=4)


for kind, A in operators():
    for p in [int(t) for t in os.environ.get("KHIP_P", "8,16,32").split(",")]:
        X, Y = K.Panel(ctx, A.n, p), K.Panel(ctx, A.n, p)
        h = np.zeros((K.panel_rows(A.n), p)); h[:A.n] = np.random.default_rng(p).standard_normal((A.n, p))
        X.buf.copy_from_host(h.ravel())
        outs = []
        variants = [(2, 1, -1), (0, 1, 0), (0, 0, 0)] + ([(2, 1, 1)] if p >= 32 else [])     # slices: -1 one launch, 1 column slices of 16
        if p > 32: variants = variants[1:]
        for tile, window, slices in variants:
            setopts(tile, window, slices)
            K.spmm_(A, X, Y); ctx.sync()
            outs.append(Y.buf.to_host())
        same = bool(all(np.array_equal(o, outs[-2 if p >= 32 else -1]) for o in outs))
        alg = 12 * A.nnz + 4 * A.n + 16 * A.n * p
        for tile, window, slices in variants:
            dt = timed(A, X, Y, tile, window, slices)
            print(json.dumps(dict(op=kind, n1=n1, p=p, tile=tile, window=window, slices=slices, ms=round(dt * 1e3, 4), alg_gbps=round(alg / dt / 1e9, 1),
                                  frac=round(alg / dt / 8e12, 4), same=same, window_rows=A.tile_info["window"], direct=A.tile_info["direct_groups"])), flush=True)
        del X, Y
    del A
setopts(1, 1, 0)
ctx.close()
