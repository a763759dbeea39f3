#!/usr/bin/env python3
"""Staged SpMV at 512^3, plain and fused with the dot: a row's end pointer by a second vector load (spmv_rp_shift = 0) or from the
neighbouring lane by a DPP wave shift (1); int32 columns and the coded stream.  A/B/A/B in one process; y compared bit for bit.
NEGATIVE RESULT (profiles/r03_sweep_rp_shift.log): bit-identical and 2-4 % slower; the option it drove (spmv_rp_shift) was not kept --
the script documents the experiment and needs that option re-added to run."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for codes in (0, 1):
    ctx.set_option("spmv_codes", codes)
    A = K.CsrMatrix.stencil(ctx, "poisson", n1)
    x = ctx.array(np.random.default_rng(1).standard_normal(A.n))
    y = ctx.zeros(A.n)
    alg = A.spmv_bytes
    def run(fn, reps=20):
        for _ in range(5): fn()
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        ctx.sync(); return (time.perf_counter() - t0) / reps
    vals, ys = {}, {}
    for early in (0, 1, 0, 1, 0, 1):
        ctx.set_option("spmv_rp_shift", early)
        tp = run(lambda: A.matvec(x, y)); tf = run(lambda: K.spmv_dot(A, x, y))
        vals.setdefault(early, K.spmv_dot(A, x, y))
        ys.setdefault(early, y.to_host())
        print(json.dumps(dict(codes=codes, rp_shift=early, plain_ms=round(tp * 1e3, 4), plain_frac=round(alg / tp / 8e12, 4),
                              fused_ms=round(tf * 1e3, 4), fused_frac=round(alg / tf / 8e12, 4), dot=repr(vals[early]))), flush=True)
    assert vals[0] == vals[1] and np.array_equal(ys[0], ys[1])
    del A, x, y
ctx.set_option("spmv_rp_shift", 0)
ctx.close()
