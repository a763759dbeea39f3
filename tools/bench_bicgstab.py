#!/usr/bin/env python3
"""bicgstab! iterations/s on kron_unsymmetric(n1) for fused = 0 / 1 / 2 (host scalars vs device-resident loop)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import krylov_jl_amd as K
ctx = K.Context(0)
for n1 in (32, 64, 128, 256):
    n = n1 ** 3
    A = K.CsrMatrix.stencil(ctx, "kron_unsymmetric", n1)
    ones = ctx.empty(n); K.kfill_(ones, 1.0)
    b = ctx.empty(n); A.matvec(ones, b)
    ws = K.BicgstabWorkspace(ctx, n, n)
    row = {"n1": n1}
    steps = 400 if n1 <= 128 else 60
    for fused in (0, 1, 2):
        K.bicgstab_(ws, A, b, itmax=5, fused=fused, atol=0.0, rtol=0.0)
        ctx.sync(); t0 = time.perf_counter()
        K.bicgstab_(ws, A, b, itmax=steps, fused=fused, atol=0.0, rtol=0.0)
        ctx.sync(); dt = time.perf_counter() - t0
        row[f"us_per_iter_fused{fused}"] = round(1e6 * dt / ws.stats.niter, 1)
    print(json.dumps(row), flush=True)
    del ws, A, b, ones
ctx.close()
