#!/usr/bin/env python3
"""SpMM window kernel at 216^3 x 16: plane-sweep group order (spmm_win_sweep, spmm_sweep_w) A/B, bit-equality of Y."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 216
p = 16
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
cases = [dict(spmm_win_sweep=0)] + [dict(spmm_win_sweep=1, spmm_sweep_w=w) for w in (16, 32, 64, 96, 128)] + [dict(spmm_win_sweep=0)]
for extra in sys.argv[2:]:
    k, v = extra.split("="); ctx.set_option(k, int(v))
for c in cases:
    for k, v in c.items(): ctx.set_option(k, v)
    t = timeit()
    y = Y.buf.to_host()
    if ref is None: ref = y
    print(json.dumps(dict(c, ms=round(t * 1e3, 4), alg_gbps=round((12 * A.nnz + 4 * n + 16 * n * p) / t / 1e9), same=bool(np.array_equal(y, ref)))), flush=True)
ctx.close()
