#!/usr/bin/env python3
"""Which bit-exact SpMV kernel per operator class: stream (1), ordered sub-wave (3), staged rows (4) on the benchmark
operators -- the data behind spmv_kernel_choice()'s thresholds.  Prints JSON lines."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
ctx.set_option("spmv_codes", int(os.environ.get("CODES", "0")))          # 0: the int32 column stream throughout
ops = [("poisson 512^3 (7/row)", lambda: K.CsrMatrix.stencil(ctx, "poisson", 512)),
       ("kron_unsymmetric 256^3 (7/row)", lambda: K.CsrMatrix.stencil(ctx, "kron_unsymmetric", 256)),
       ("stencil27 216^3 (27/row)", lambda: K.CsrMatrix.stencil(ctx, "stencil27", 216)),
       ("banded+random 10M (26.7/row)", lambda: K.CsrMatrix.banded_random(ctx, 10 << 20, seed=1)),
       ("banded+random hb=4 links=2 16M (9.9/row)", lambda: K.CsrMatrix.banded_random(ctx, 16 << 20, half_band=4, links=2, seed=1)),
       ("banded+random hb=40 links=4 4M (74/row)", lambda: K.CsrMatrix.banded_random(ctx, 4 << 20, half_band=40, links=4, seed=1)),
       ("banded+random unsym + 4 dense rows 10M", lambda: K.CsrMatrix.banded_random(ctx, 10 << 20, seed=1, unsym=True, dense_rows=4))]
only = os.environ.get("ONLY")
for name, make in ops:
    if only and only not in name: continue
    A = make()
    n = A.n
    x = ctx.array(np.cos(np.arange(n) * 1e-3) + 0.5)
    y = ctx.zeros(n)
    alg = A.spmv_bytes
    ref = None
    for kern in (0, 1, 3, 4):
        for fused in (False, True):
            ctx.set_option("spmv_kernel", kern)
            fn = (lambda: K.spmv_dot(A, x, y)) if fused else (lambda: A.matvec(x, y))
            try:
                fn(); ctx.sync(); t0 = time.perf_counter()
                for _ in range(10): fn()
                ctx.sync(); t = (time.perf_counter() - t0) / 10
                h = y.to_host()
                if ref is None: ref = h
                print(json.dumps(dict(op=name, nnz_per_row=round(A.nnz / n, 2), codes=A.code_info, kernel=kern, fused_dot=fused, ms=round(t * 1e3, 4),
                                      frac=round(alg / t / 8e12, 4), same=bool(np.array_equal(h, ref)))), flush=True)
            except Exception as e:
                print(json.dumps(dict(op=name, kernel=kern, fused_dot=fused, error=str(e)[:160])), flush=True)
    ctx.set_option("spmv_kernel", 0)
    del A, x, y
ctx.close()
