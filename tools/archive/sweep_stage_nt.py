#!/usr/bin/env python3
"""SpMV-only sweep at 512^3: nt (streaming) cache policy on the staged kernel's val/col window loads."""
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
for rep in range(2):
    for nt in (0, 1):
        for nty in (0, 1):
            case(spmv_kernel=4, spmv_rows=256, spmv_nt=nt, spmv_nty=nty)
ctx.close()
