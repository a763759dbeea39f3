#!/usr/bin/env python3
"""SpMV-only A/B at 512^3: pipelined staged kernel (spmv_pipe = row blocks per workgroup, 0 = off) x column stream
(spmv_codes 0 = int32, 1 = 8-bit diagonal codes), plain product and product fused with x.y.  JSON lines."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import krylov_jl_amd as K
ctx = K.Context(0)
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 512
pipes = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 2, 4, 8, 16]
extra = dict(kv.split("=") for kv in sys.argv[3:])
for k, v in extra.items():
    ctx.set_option(k, int(v))
n = n1 ** 3
x, y = ctx.empty(n), ctx.empty(n)
K.kfill_(x, 1.0)
def timeit(fn, reps=20):
    fn(); ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.sync(); return (time.perf_counter() - t0) / reps
for codes in (1, 0):
    ctx.set_option("spmv_codes", codes)
    A = K.CsrMatrix.stencil(ctx, "poisson", n1)
    A.matvec(x, y); ctx.sync()
    sb, moved = A.spmv_bytes, A.spmv_bytes_stored
    for rnd in range(2):
        for pipe in pipes:
            ctx.set_option("spmv_pipe", pipe)
            t = timeit(lambda: A.matvec(x, y)); t2 = timeit(lambda: K.spmv_dot(A, x, y))
            print(json.dumps(dict(n1=n1, spmv_codes=codes, spmv_pipe=pipe, **extra, ms=round(t * 1e3, 4), ms_dot=round(t2 * 1e3, 4),
                                  alg_frac=round(sb / t / 8e12, 4), alg_frac_dot=round(sb / t2 / 8e12, 4),
                                  moved_gbps=round(moved / t / 1e9), moved_gbps_dot=round(moved / t2 / 1e9))), flush=True)
    del A
ctx.close()
