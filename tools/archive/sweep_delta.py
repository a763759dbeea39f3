#!/usr/bin/env python3
"""The stream SpMV's forms on the non-stencil operators (round 4): spmv_stream_kernel (8 + 4 byte loads), the 16-byte-load
form on int32 columns (spmv_delta_kernel<int32_t>), the block-delta column stream with 8- and 16-bit codes (coldelta.hip),
the staged kernel for comparison.  Operators: banded + random 10.5 M rows (symmetric; nonsymmetric + 4 rows of 3000
entries), the 27-point 216^3 operator with its dictionary switched off, kron_unsymmetric 256^3 int32.  Fractions are of
8.0 TB/s on the ALGORITHMIC bytes and on the bytes the form streams.  y of every form must equal the first
one's bit for bit.  Prints JSON lines."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
small = "--small" in sys.argv
n = (1 << 17) if small else 10 * (1 << 20)
reps = 20

def timeit(fn):
    fn(); ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.sync(); return (time.perf_counter() - t0) / reps

FORMS = [("stream 8+4 B loads", dict(spmv_kernel=1, spmv_delta=0, spmv_wide=0)),
         ("stream 16 B loads, int32 columns", dict(spmv_kernel=1, spmv_delta=0, spmv_wide=1)),
         ("stream, block-delta 16 bit", dict(spmv_kernel=1, spmv_delta=16, spmv_wide=1)),
         ("stream, block-delta 8 bit", dict(spmv_kernel=1, spmv_delta=8, spmv_wide=1)),
         ("default choice", dict(spmv_kernel=0, spmv_delta=1, spmv_wide=1)),
         ("staged rows, int32", dict(spmv_kernel=4, spmv_delta=0, spmv_wide=0))]
BASE = dict(spmv_kernel=0, spmv_delta=1, spmv_wide=1, spmv_codes=1)

def operators():
    yield "banded+random sym 10.5M", lambda: K.CsrMatrix.banded_random(ctx, n, seed=1), {}
    yield "banded+random unsym + 4 dense rows", lambda: K.CsrMatrix.banded_random(ctx, n, seed=1, unsym=True, dense_rows=4), {}
    if not small:
        yield "27-point 216^3, dictionary off", lambda: K.CsrMatrix.stencil(ctx, "stencil27", 216), dict(spmv_codes=0)
        yield "kron_unsymmetric 256^3, dictionary off", lambda: K.CsrMatrix.stencil(ctx, "kron_unsymmetric", 256), dict(spmv_codes=0)

for name, make, extra in operators():
    ref = None
    for form, opts in FORMS:
        for k, v in {**BASE, **extra, **opts}.items(): ctx.set_option(k, v)
        A = make()                                   # a fresh handle: the column stream is built at the first product
        m = A.shape[0]
        x = ctx.array(np.cos(np.arange(m) * 1e-3) + 0.5)
        y = ctx.zeros(m)
        try:
            t0 = time.perf_counter(); A.matvec(x, y); ctx.sync(); first_ms = 1e3 * (time.perf_counter() - t0)
            t = timeit(lambda: A.matvec(x, y))
            h = y.to_host()
            if ref is None: ref = h
            td = timeit(lambda: K.spmv_dot(A, x, y))
            alg, moved = A.spmv_bytes, A.spmv_bytes_stored
            print(json.dumps(dict(operator=name, form=form, ms=round(t * 1e3, 4), ms_fused_dot=round(td * 1e3, 4), frac=round(alg / t / 8e12, 4),
                                  frac_moved=round(moved / t / 8e12, 4), alg_bytes=alg, moved_bytes=moved, delta_info=list(A.delta_info),
                                  code_info=list(A.code_info), first_product_ms=round(first_ms, 2), same=bool(np.array_equal(h, ref)))), flush=True)
        except Exception as e:
            print(json.dumps(dict(operator=name, form=form, error=str(e)[:300])), flush=True)
        del A, x, y
for k, v in BASE.items(): ctx.set_option(k, v)
ctx.close()
