#!/usr/bin/env python3
"""A handful of SpMM launches (p = 16) on the banded + random operator (10.5 M rows) for rocprofv3 counter passes.
argv: key=value tuning options (spmm_tile, spmm_tile_slide, spmm_window, ...)."""
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
p = 16
A = K.CsrMatrix.banded_random(ctx, n, seed=1)
X, Y = K.Panel(ctx, n, p), K.Panel(ctx, n, p)
for _ in range(4):
    K.spmm_(A, X, Y)
ctx.sync()
print("nnz", A.nnz, "algorithmic bytes", 12 * A.nnz + 4 * n + 16 * n * p, "tile_info", A.tile_info)
ctx.close()
