#!/usr/bin/env python3
"""SpMV-only sweep at 512^3: staged-kernel LDS window (spmv_cap) x XCD remap x nt-store of y."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import krylov_jl_amd as K
ctx = K.Context(0)
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = n1 ** 3
A = K.CsrMatrix.stencil(ctx, "poisson", n1)
x, y = ctx.empty(n), ctx.empty(n)
K.kfill_(x, 1.0)
sb = A.spmv_bytes
def timeit(fn, reps=20):
    fn(); ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.sync(); return (time.perf_counter() - t0) / reps
def case(**o):
    for k, v in o.items(): ctx.set_option(k, v)
    t = timeit(lambda: A.matvec(x, y)); t2 = timeit(lambda: K.spmv_dot(A, x, y))
    print(json.dumps(dict(o, ms=round(t * 1e3, 4), gbps=round(sb / t / 1e9), frac=round(sb / t / 8e12, 4), ms_dot=round(t2 * 1e3, 4))), flush=True)
base = dict(spmv_kernel=4, spmv_rows=256)
for cap in (0, 2048):
    for xcd in (0, 8):
        for nty in (0, 1):
            case(**base, spmv_cap=cap, spmv_xcd=xcd, spmv_nty=nty)
for rows in ():
    case(spmv_kernel=4, spmv_rows=rows, spmv_cap=0, spmv_xcd=0, spmv_nty=0)
ctx.close()
