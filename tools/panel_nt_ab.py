#!/usr/bin/env python3
"""cfg 5 block_gmres! (p = 16, memory 5, 27-point 216^3) with the panel kernels' streaming policy switched: ctx option panel_nt = 0 / 2
(non-temporal loads and stores to the panels).  Per setting: ms per iteration and the HIP-event averages of the panel kernels."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
n1 = 216
n, p = n1 ** 3, 16
A = K.CsrMatrix.stencil(ctx, "stencil27", n1)
t = (np.arange(n) + 1.0) / n
Xt = np.stack([np.cos(j * np.pi * t) + 0.1 * j for j in range(p)], axis=1)
dXt = K.Panel.from_host(ctx, Xt)
dB = K.Panel(ctx, n, p)
K.spmm_(A, dXt, dB)
Bd = ctx.array(np.asfortranarray(dB.to_host()).ravel(order="F"))
ws = K.BlockGmresWorkspace(ctx, n, n, p, memory=5)
ref = None
for rnd in range(2):
    for v in [int(a) for a in sys.argv[1:]] or [0, 2]:
        ctx.set_option("panel_nt", v)
        K.block_gmres_(ws, A, Bd, restart=True, itmax=5, atol=0.0, rtol=0.0)
        ctx.set_option("profile_spmv", 1); ctx.profile_kernels()
        ctx.sync(); t0 = time.perf_counter()
        K.block_gmres_(ws, A, Bd, restart=True, itmax=20, atol=0.0, rtol=0.0, history=True)
        ctx.sync(); dt = time.perf_counter() - t0
        prof = ctx.profile_kernels(); ctx.set_option("profile_spmv", 0)
        h = np.array(ws.stats.residuals)
        ref = h if ref is None else ref
        print(json.dumps({"panel_nt": v, "ms_per_iteration": round(1e3 * dt / ws.stats.niter, 4), "history_bit_identical": bool(np.array_equal(h, ref)),
                          "kernels_avg_ms": {k: round(ms / l, 4) for k, (l, ms) in prof.items() if l}}), flush=True)
ctx.close()
