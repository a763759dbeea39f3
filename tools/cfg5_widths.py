#!/usr/bin/env python3
"""block_gmres! on the 27-point 216^3 operator with p = 8, 16, 32 right-hand sides, memory 5: ms per iteration with the tile
SpMM at that width (spmm_tile = 1: the library's rule) and without it (spmm_tile = 0: window / direct kernels).
Usage: python tools/cfg5_widths.py [n1]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 216
n = n1 ** 3
A = K.CsrMatrix.stencil(ctx, "stencil27", n1)
for p in (8, 16, 32):
    rng = np.random.default_rng(p)
    Xt = rng.standard_normal((n, p))
    dXt = K.Panel.from_host(ctx, Xt)
    dB = K.Panel(ctx, n, p)
    K.spmm_(A, dXt, dB)
    Bd = ctx.array(np.asfortranarray(dB.to_host()).ravel(order="F"))
    del dXt, dB
    out = {}
    for tile in (1, 0):
        ctx.set_option("spmm_tile", tile)
        ws = K.BlockGmresWorkspace(ctx, n, n, p, memory=5)
        K.block_gmres_(ws, A, Bd, restart=True, itmax=5, atol=0.0, rtol=0.0)
        ctx.sync(); t0 = time.perf_counter()
        K.block_gmres_(ws, A, Bd, restart=True, itmax=20, atol=0.0, rtol=0.0, history=True)
        ctx.sync(); dt = time.perf_counter() - t0
        out[tile] = (1e3 * dt / ws.stats.niter, ws.stats.residuals.copy())
        del ws
    ctx.set_option("spmm_tile", 1)
    print(json.dumps(dict(config=f"block_gmres! p={p} memory=5 stencil27 {n1}^3", ms_per_iter_tile=round(out[1][0], 3),
                          ms_per_iter_without=round(out[0][0], 3), same_history=bool(np.array_equal(out[1][1], out[0][1])))), flush=True)
    del Bd
ctx.close()
