#!/usr/bin/env python3
"""cfg 5 alone (block_gmres!, p = 16, memory 5, 27-point 216^3): 5 warm-up + 20 timed iterations, for rocprofv3 --kernel-trace --stats."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 216
n, p = n1 ** 3, 16
A = K.CsrMatrix.stencil(ctx, "stencil27", n1)
t = (np.arange(n) + 1.0) / n
Xt = np.stack([np.cos(j * np.pi * t) + 0.1 * j for j in range(p)], axis=1)
dXt = K.Panel.from_host(ctx, Xt)
dB = K.Panel(ctx, n, p)
K.spmm_(A, dXt, dB)
Bd = ctx.array(np.asfortranarray(dB.to_host()).ravel(order="F"))
ws = K.BlockGmresWorkspace(ctx, n, n, p, memory=5)
K.block_gmres_(ws, A, Bd, restart=True, itmax=5, atol=0.0, rtol=0.0)
ctx.sync(); t0 = time.perf_counter()
K.block_gmres_(ws, A, Bd, restart=True, itmax=20, atol=0.0, rtol=0.0)
ctx.sync(); dt = time.perf_counter() - t0
print(f"cfg5: {ws.stats.niter} iterations, {1e3 * dt / ws.stats.niter:.3f} ms per iteration")
ctx.close()
