#!/usr/bin/env python3
"""SpMV-only A/B at 512^3 (or argv[1]^3): int32 column stream vs 8-bit / 16-bit diagonal codes (csrc/colcode.hip),
plain product and the product fused with x.y, same box, interleaved.  Prints JSON lines."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 512
kind = sys.argv[2] if len(sys.argv) > 2 else "poisson"
n = n1 ** 3
x, y = ctx.empty(n), ctx.empty(n)
K.kfill_(x, 1.0)
def timeit(fn, reps=20):
    fn(); ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.sync(); return (time.perf_counter() - t0) / reps
ref = None
for rnd in range(2):
    for mode in (0, 1, 16):
        ctx.set_option("spmv_codes", mode)
        ctx.set_option("spmv_kernel", 4)
        A = K.CsrMatrix.stencil(ctx, kind, n1)
        t0 = time.perf_counter(); A.matvec(x, y); ctx.sync(); t_first = time.perf_counter() - t0
        sb, moved = A.spmv_bytes, A.spmv_bytes_stored
        t = timeit(lambda: A.matvec(x, y)); t2 = timeit(lambda: K.spmv_dot(A, x, y))
        if n1 <= 256:
            yh = y.to_host()
            if ref is None: ref = yh
            assert np.array_equal(yh, ref)
        print(json.dumps(dict(kind=kind, n1=n1, spmv_codes=mode, code_info=A.code_info, first_call_ms=round(t_first * 1e3, 2),
                              ms=round(t * 1e3, 4), ms_dot=round(t2 * 1e3, 4), alg_gbps=round(sb / t / 1e9), alg_frac=round(sb / t / 8e12, 4),
                              alg_frac_dot=round(sb / t2 / 8e12, 4), moved_gbps=round(moved / t / 1e9), moved_gbps_dot=round(moved / t2 / 1e9))), flush=True)
        del A
ctx.close()
