#!/usr/bin/env python3
"""Fixed cost of one cg! call at 512^3 (setup passes, syncs, history download): time solves of 1, 2, 5, 10, 20, 50, 100 iterations
and fit t = a + b k.  Usage: python tools/archive/cg_solve_overhead.py [n1]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 512
A = K.CsrMatrix.stencil(ctx, "poisson", n1)
b = ctx.empty(A.n); K.kfill_(b, 1.0)
ws = K.CgWorkspace(ctx, A.n, A.n)
K.cg_(ws, A, b, atol=0.0, rtol=0.0, itmax=200, fused=2)
ctx.sync()
ks = (1, 2, 5, 10, 20, 50, 100)
for hist in (True, False):
    ts = []
    for k in ks:
        best = 1e9
        for rep in range(3):
            ctx.sync(); t0 = time.perf_counter()
            K.cg_(ws, A, b, atol=0.0, rtol=0.0, itmax=k, history=hist, fused=2)
            ctx.sync(); best = min(best, time.perf_counter() - t0)
        ts.append(best)
    bfit, afit = np.polyfit(np.array(ks, float), np.array(ts), 1)
    print(json.dumps(dict(n1=n1, history=hist, ms_by_k={k: round(t * 1e3, 3) for k, t in zip(ks, ts)}, fixed_ms=round(afit * 1e3, 3), per_iteration_ms=round(bfit * 1e3, 4))), flush=True)
ctx.close()
