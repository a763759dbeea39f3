#!/usr/bin/env python3
"""Direct-gather SpMM (spmm2_kernel, 16 B per lane) at 216^3 x 16 with and without the plane-sweep tile order."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
n1, p = 216, 16
n = n1 ** 3
A = K.CsrMatrix.stencil(ctx, "stencil27", n1)
X, Y = K.Panel(ctx, n, p), K.Panel(ctx, n, p)
h = np.random.default_rng(1).standard_normal(K.panel_rows(n) * p); h[n * p:] = 0
X.buf.copy_from_host(h)
def timeit(reps=10):
    K.spmm_(A, X, Y); ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps): K.spmm_(A, X, Y)
    ctx.sync(); return (time.perf_counter() - t0) / reps
ref = None
for c in [dict(spmm_window=1), dict(spmm_window=0, spmm_sweep=0)] + [dict(spmm_window=0, spmm_sweep=1, spmm_sweep_w=w) for w in (8, 16, 32, 64, 128)]:
    for k, v in c.items(): ctx.set_option(k, v)
    t = timeit(); y = Y.buf.to_host()
    if ref is None: ref = y
    print(json.dumps(dict(c, ms=round(t * 1e3, 4), same=bool(np.array_equal(y, ref)))), flush=True)
ctx.close()
