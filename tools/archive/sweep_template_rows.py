#!/usr/bin/env python3
"""Row-template SpMV at 512^3: rows per lane, nt store of y, XCD remap."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import krylov_jl_amd as K
ctx = K.Context(0)
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = n1 ** 3
A = K.CsrMatrix.stencil(ctx, "poisson", n1)
T = A.compress()
x, y = ctx.empty(n), ctx.empty(n)
K.kfill_(x, 1.0)
sb = A.spmv_bytes_stored
def timeit(fn, reps=20):
    fn(); ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.sync(); return (time.perf_counter() - t0) / reps
def case(**o):
    for k, v in o.items(): ctx.set_option(k, v)
    t = timeit(lambda: A.matvec(x, y)); t2 = timeit(lambda: K.spmv_dot(A, x, y))
    print(json.dumps(dict(o, templates=T, ms=round(t * 1e3, 4), gbps_stored=round(sb / t / 1e9), ms_dot=round(t2 * 1e3, 4))), flush=True)
for rpt in (1, 2, 4, 8, 16):
    for nty in (0, 1):
        case(spmv_tmpl_rows=rpt, spmv_nty=nty)
ctx.close()
