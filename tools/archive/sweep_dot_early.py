#!/usr/bin/env python3
"""Staged SpMV fused with the dot at 512^3: dotw[row] loaded before the row block's windows (spmv_dot_early = 1) or after the
row walk (0); int32 columns (CODES=0) and the coded stream (CODES=1).  A/B/A/B in one process."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for codes in (0, 1):
    ctx.set_option("spmv_codes", codes)
    A = K.CsrMatrix.stencil(ctx, "poisson", n1)
    x = ctx.empty(A.n); K.kfill_(x, 1.0)
    y = ctx.zeros(A.n)
    alg = A.spmv_bytes
    def run(fn, reps=20):
        for _ in range(5): fn()
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        ctx.sync(); return (time.perf_counter() - t0) / reps
    vals = {}
    for early in (0, 1, 0, 1, 0, 1):
        ctx.set_option("spmv_dot_early", early)
        tp = run(lambda: A.matvec(x, y)); tf = run(lambda: K.spmv_dot(A, x, y))
        vals.setdefault(early, K.spmv_dot(A, x, y))
        print(json.dumps(dict(codes=codes, dot_early=early, plain_ms=round(tp * 1e3, 4), plain_frac=round(alg / tp / 8e12, 4),
                              fused_ms=round(tf * 1e3, 4), fused_frac=round(alg / tf / 8e12, 4), dot=repr(vals[early]))), flush=True)
    assert vals[0] == vals[1]
    del A, x, y
ctx.set_option("spmv_dot_early", 0)
ctx.close()
