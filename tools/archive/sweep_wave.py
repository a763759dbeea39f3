#!/usr/bin/env python3
"""The wave-private LDS-DMA SpMV (spmv_wave_kernel, spmv_kernel = 6) against the stream and staged kernels (round 4), int32
columns throughout: banded + random 10.5 M rows with 0 / 3 links per row, 27-point 216^3, 7-point 512^3, kron_unsymmetric 256^3.
Fractions of 8 TB/s on the algorithmic bytes; y must equal the first form's bit for bit.  Prints JSON lines."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
reps = 20
def timeit(fn):
    fn(); ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.sync(); return (time.perf_counter() - t0) / reps
BASE = dict(spmv_kernel=0, spmv_delta=0, spmv_wide=0, spmv_codes=0, spmv_cap=0, spmv_nty=0)
FORMS = [("stream", dict(spmv_kernel=1)), ("staged", dict(spmv_kernel=4))] + \
        [("wave cap=%s" % (c or "auto"), dict(spmv_kernel=6, spmv_cap=c)) for c in (0, 512, 1024, 2048)] + \
        [("wave cap=auto nt y", dict(spmv_kernel=6, spmv_nty=1))]
n = 10 * (1 << 20)
OPS = [("banded+random links=0", lambda: K.CsrMatrix.banded_random(ctx, n, links=0, seed=1)),
       ("banded+random links=3", lambda: K.CsrMatrix.banded_random(ctx, n, links=3, seed=1)),
       ("27-point 216^3", lambda: K.CsrMatrix.stencil(ctx, "stencil27", 216)),
       ("7-point 512^3", lambda: K.CsrMatrix.stencil(ctx, "poisson", 512)),
       ("kron_unsymmetric 256^3", lambda: K.CsrMatrix.stencil(ctx, "kron_unsymmetric", 256))]
only = [a for a in sys.argv[1:] if not a.startswith("-")]
for name, make in OPS:
    if only and not any(o in name for o in only): continue
    for k, v in BASE.items(): ctx.set_option(k, v)
    A = make()
    m = A.shape[0]
    x = ctx.array(np.cos(np.arange(m) * 1e-3) + 0.5)
    y = ctx.zeros(m)
    ref = None
    for form, opts in FORMS:
        for k, v in {**BASE, **opts}.items(): ctx.set_option(k, v)
        try:
            t = timeit(lambda: A.matvec(x, y))
            h = y.to_host()
            if ref is None: ref = h
            td = timeit(lambda: K.spmv_dot(A, x, y))
            alg = A.spmv_bytes
            print(json.dumps(dict(operator=name, form=form, ms=round(t * 1e3, 4), ms_fused_dot=round(td * 1e3, 4), frac=round(alg / t / 8e12, 4),
                                  frac_fused=round(alg / td / 8e12, 4), same=bool(np.array_equal(h, ref)))), flush=True)
        except Exception as e:
            print(json.dumps(dict(operator=name, form=form, error=str(e)[:300])), flush=True)
    del A, x, y
ctx.close()
