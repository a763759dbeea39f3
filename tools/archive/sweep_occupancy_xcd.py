#!/usr/bin/env python3
"""SpMV-only sweep at 512^3 for the XCD run-length / nt-store knobs."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import krylov_jl_amd as K
ctx = K.Context(0)
n1 = 512; n = n1 ** 3
A = K.CsrMatrix.stencil(ctx, "poisson", n1)
x, y = ctx.empty(n), ctx.empty(n)
K.kfill_(x, 1.0)
sb = A.spmv_bytes
def timeit(fn, reps=10):
    fn(); ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.sync(); return (time.perf_counter() - t0) / reps
def case(**o):
    for k, v in o.items(): ctx.set_option(k, v)
    t = timeit(lambda: A.matvec(x, y)); t2 = timeit(lambda: K.spmv_dot(A, x, y))
    print(json.dumps(dict(o, ms=round(t * 1e3, 4), gbps=round(sb / t / 1e9), frac=round(sb / t / 8e12, 4), ms_dot=round(t2 * 1e3, 4))), flush=True)
base = dict(spmv_vec=1, spmv_nt=0, spmv_persist=0, spmv_nty=0, spmv_fake_gather=0, spmv_tiles=1, spmv_rows=256, spmv_kernel=4, spmv_xcd=0)
for pad in (0, 8192, 16384, 32768, 57344, 0):      # 6, 4, 3, 2, 1(+) workgroups per CU
    case(**base, spmv_lds_pad=pad)
ctx.close()
