#!/usr/bin/env python3
"""SpMM p=16 on the 27-point 216^3 operator, a few launches (for rocprofv3 counter passes)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 216
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
p = 16
n = n1 ** 3
for kv in filter(None, os.environ.get("KHIP_OPTS", "").split(",")):      # e.g. KHIP_OPTS=spmm_tile_pencil=54,spmm_tile_exp=8
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
A = K.CsrMatrix.stencil(ctx, "stencil27", n1)
X, Y = K.Panel(ctx, n, p), K.Panel(ctx, n, p)
K.spmm_(A, X, Y); ctx.sync()
t0 = time.perf_counter()
for _ in range(reps): K.spmm_(A, X, Y)
ctx.sync()
dt = (time.perf_counter() - t0) / reps
print(f"spmm p={p} n1={n1}: {dt*1e3:.3f} ms, {(12*A.nnz + 4*n + 16*n*p)/dt/1e9:.0f} GB/s algorithmic")
ctx.close()
