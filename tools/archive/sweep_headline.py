#!/usr/bin/env python3
"""Headline kernel A/B at 512^3 (round 4): plain and fused (p.Ap) coded SpMV under tuning switches, same box, same process.
argv: option sets "k=v,k=v" (an empty string = defaults).  Prints JSON lines; diff = ||y - y_default|| must be 0."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import krylov_jl_amd as K
ctx = K.Context(0)
n1 = 512
n = n1 ** 3
A = K.CsrMatrix.stencil(ctx, "poisson", n1)
x, y, y0 = ctx.empty(n), ctx.empty(n), ctx.empty(n)
K.kfill_(x, 1.0)
sb, mv = A.spmv_bytes, None
def timeit(fn, reps=30):
    fn(); ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.sync(); return (time.perf_counter() - t0) / reps
A.matvec(x, y0)
sets = sys.argv[1:] or [""]
for rnd in range(2):
    for sset in sets:
        opts = dict(kv.split("=") for kv in sset.split(",") if kv)
        saved = {k: ctx.get_option(k) for k in opts}
        for k, v in opts.items(): ctx.set_option(k, int(v))
        t = timeit(lambda: A.matvec(x, y)); d = None
        K.kaxpy_(n, -1.0, y0, y); d = K.knorm(n, y)
        t2 = timeit(lambda: K.spmv_dot(A, x, y))
        val = K.spmv_dot(A, x, y)
        print(json.dumps(dict(round=rnd, opts=sset or "defaults", ms=round(t * 1e3, 4), ms_dot=round(t2 * 1e3, 4), frac=round(sb / t / 8e12, 4),
                              frac_dot=round(sb / t2 / 8e12, 4), frac_moved_dot=round(A.spmv_bytes_stored / t2 / 8e12, 4), diff=d, dot=val)), flush=True)
        for k, v in saved.items(): ctx.set_option(k, v)
ctx.close()
