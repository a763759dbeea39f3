#!/usr/bin/env python3
"""CG iterations/s on get_div_grad(n1^3) for a ladder of sizes and fusion levels (1 GPU): the latency -> bandwidth
transition.  JSON lines."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import krylov_jl_amd as K
ctx = K.Context(0)
for n1 in (32, 64, 96, 128, 192, 256, 384, 512):
    n = n1 ** 3
    A = K.CsrMatrix.stencil(ctx, "poisson", n1)
    b = ctx.empty(n); K.kfill_(b, 1.0)
    ws = K.CgWorkspace(ctx, n, n)
    row = {"n1": n1, "n": n, "spmv_bytes": A.spmv_bytes}
    steps = 2000 if n1 <= 128 else (600 if n1 <= 256 else 150)
    for fused in (0, 1, 2):
        K.cg_(ws, A, b, atol=0.0, rtol=0.0, itmax=20, fused=fused)
        ctx.sync(); t0 = time.perf_counter()
        K.cg_(ws, A, b, atol=0.0, rtol=0.0, itmax=steps, fused=fused)
        ctx.sync(); dt = time.perf_counter() - t0
        row[f"its_fused{fused}"] = round(ws.stats.niter / dt, 1)
        row[f"us_per_iter_fused{fused}"] = round(1e6 * dt / ws.stats.niter, 1)
    row["gbps_fused2"] = round((A.spmv_bytes + 64 * n) * row["its_fused2"] / 1e9)
    print(json.dumps(row), flush=True)
    del ws, A, b
ctx.close()
