#!/usr/bin/env python3
"""spmm_tile_kernel (spmm_tile.hip) vs the window and direct kernels at p = 16: bit-equality, then timing on the
27-point 216^3 operator (cfg 5) and on a banded + random operator.  Usage: python tools/spmm_tile_check.py [n1] [reps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K

ctx = K.Context(0)
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 216
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
p = 16


def timed(A, X, Y, tile, window):
    ctx.set_option("spmm_tile", tile); ctx.set_option("spmm_window", window)
    K.spmm_(A, X, Y); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps): K.spmm_(A, X, Y)
    ctx.sync()
    return (time.perf_counter() - t0) / reps


for kind, dims in (("stencil27", (n1, n1, n1)), ("poisson", (n1, n1, n1))):
    A = K.CsrMatrix.stencil(ctx, kind, *dims)
    X, Y = K.Panel(ctx, A.n, p), K.Panel(ctx, A.n, p)
    h = np.zeros((K.panel_rows(A.n), p)); h[:A.n] = np.random.default_rng(0).standard_normal((A.n, p))
    X.buf.copy_from_host(h.ravel())
    outs = []
    for tile, window in ((1, 1), (0, 1), (0, 0)):
        ctx.set_option("spmm_tile", tile); ctx.set_option("spmm_window", window)
        t0 = time.perf_counter(); K.spmm_(A, X, Y); ctx.sync(); first = time.perf_counter() - t0
        outs.append(Y.buf.to_host())
        print(json.dumps(dict(op=kind, n1=n1, tile=tile, window=window, first_call_ms=first * 1e3)), flush=True)
    same = bool(np.array_equal(outs[0], outs[2]) and np.array_equal(outs[1], outs[2]))
    alg = 12 * A.nnz + 4 * A.n + 16 * A.n * p
    for tile, window in ((1, 1), (0, 1), (0, 0), (1, 1)):
        dt = timed(A, X, Y, tile, window)
        print(json.dumps(dict(op=kind, n1=n1, tile=tile, window=window, ms=dt * 1e3, alg_gbps=alg / dt / 1e9, frac=alg / dt / 8e12,
                              same=same, info=A.tile_info)), flush=True)
    del A, X, Y
ctx.set_option("spmm_tile", 1); ctx.set_option("spmm_window", 1)
ctx.close()
