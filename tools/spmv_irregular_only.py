#!/usr/bin/env python3
"""A handful of SpMV launches on the banded + random operator (10.5 M rows) for rocprofv3 counter passes.
argv: key=value tuning options (spmv_kernel, spmv_delta, spmv_wide, ...)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
n = 10 * (1 << 20)
A = K.CsrMatrix.banded_random(ctx, n, seed=1)
x = ctx.array(np.cos(np.arange(n) * 1e-3) + 0.5)
y = ctx.zeros(n)
for _ in range(4):
    A.matvec(x, y)
ctx.sync()
print("spmv_bytes", A.spmv_bytes, "stored", A.spmv_bytes_stored, "delta", A.delta_info, "code", A.code_info)
ctx.close()
