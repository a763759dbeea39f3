#!/usr/bin/env python3
"""A handful of SpMV (+dot) launches at 512^3 for rocprofv3 (kernel trace / PMC passes)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import krylov_jl_amd as K  # noqa: E402
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = K.Context(0)
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
n = n1 ** 3
A = K.CsrMatrix.stencil(ctx, "poisson", n1)
x, y = ctx.empty(n), ctx.empty(n)
K.kfill_(x, 1.0)
for _ in range(reps):
    A.matvec(x, y)
for _ in range(reps):
    K.spmv_dot(A, x, y)
for _ in range(reps):
    K.kaxpy_(n, 1e-9, x, y)
    K.kdot(n, x, y)
ctx.sync()
print("spmv_bytes", A.spmv_bytes)
ctx.close()
