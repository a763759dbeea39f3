#!/usr/bin/env python3
"""GMRES(30) inner-iteration time at 256^3 for MGS-cascade knobs."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import krylov_jl_amd as K
ctx = K.Context(0)
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = n1 ** 3
A = K.CsrMatrix.stencil(ctx, "kron_unsymmetric", n1)
ones = ctx.empty(n); K.kfill_(ones, 1.0)
b = ctx.empty(n); A.matvec(ones, b)
for rep in range(2):
    for keep in (-1, 0, 1):
        ctx.set_option("mgs_keep", keep)
        ws = K.GmresWorkspace(ctx, n, n, memory=30)
        K.gmres_(ws, A, b, restart=True, itmax=30, fused=True, atol=0.0, rtol=0.0)
        ctx.sync(); t0 = time.perf_counter()
        K.gmres_(ws, A, b, restart=True, itmax=90, fused=True, atol=0.0, rtol=0.0, history=True)
        ctx.sync(); dt = time.perf_counter() - t0
        print(json.dumps({"mgs_keep": keep, "ms_per_inner_iter": round(1e3 * dt / ws.stats.niter, 4), "last": float(ws.stats.residuals[-1])}), flush=True)
        del ws
ctx.close()
