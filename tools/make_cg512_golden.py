#!/usr/bin/env python3
"""Residual history of cg! on get_div_grad(512,512,512), b = ones, on ONE GPU (fused = 0: the reference's primitive
sequence) -> tests/golden/cg512_residuals.json.  bench.py compares the history of every run (any GPU count, any
fusion level) against it, so the scaling runs carry their own parity evidence."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import krylov_jl_amd as K
ctx = K.Context(0)
n1 = 512
n = n1 ** 3
A = K.CsrMatrix.stencil(ctx, "poisson", n1)
b = ctx.empty(n); K.kfill_(b, 1.0)
ws = K.CgWorkspace(ctx, n, n)
out = {}
for fused in (0, 2):
    K.cg_(ws, A, b, atol=0.0, rtol=0.0, itmax=600, history=True, fused=fused)
    out[fused] = [float(v) for v in ws.stats.residuals]
dev = max(abs(a - c) / a for a, c in zip(out[0], out[2]))
json.dump({"operator": "get_div_grad(512,512,512)", "b": "ones", "x0": "zeros", "fused": 0, "n_gpus": 1,
           "max_rel_dev_fused2_vs_fused0": dev, "residuals": out[0]},
          open(os.path.join(ROOT, "gpurun_out", "cg512_residuals.json"), "w"))
print("iterations", len(out[0]) - 1, "fused2 vs fused0 max rel dev", dev)
