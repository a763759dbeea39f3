#!/usr/bin/env python3
"""cfg 3 alone (gmres!(30, restart) on kron_unsymmetric 256^3, b = A*ones): one warm-up cycle + 90 timed inner iterations,
for rocprofv3 --kernel-trace --stats."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = n1 ** 3
A = K.CsrMatrix.stencil(ctx, "kron_unsymmetric", n1)
ones = ctx.empty(n); K.kfill_(ones, 1.0)
b = ctx.empty(n); A.matvec(ones, b)
ws = K.GmresWorkspace(ctx, n, n, memory=30)
K.gmres_(ws, A, b, restart=True, itmax=30, atol=0.0, rtol=0.0)
ctx.sync(); t0 = time.perf_counter()
K.gmres_(ws, A, b, restart=True, itmax=90, atol=0.0, rtol=0.0)
ctx.sync(); dt = time.perf_counter() - t0
print(f"cfg3: {ws.stats.niter} inner iterations, {1e3 * dt / ws.stats.niter:.3f} ms per inner iteration")
ctx.close()
