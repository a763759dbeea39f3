#!/usr/bin/env python3
"""What the long-range links of the banded + random operator cost (round 4): the same band (half width 13, 7 of 8 kept)
with 0, 1 and 3 random links per row, 10.5 M rows; stream kernel forms x non-temporal matrix stream.  Fractions of 8 TB/s on
the algorithmic bytes.  Prints JSON lines."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
n = 10 * (1 << 20)
reps = 20
def timeit(fn):
    fn(); ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.sync(); return (time.perf_counter() - t0) / reps
FORMS = [("stream 8+4 B loads", dict(spmv_kernel=1, spmv_delta=0, spmv_wide=0, spmv_nt=0, spmv_stream_nt=0)),
         ("stream 8+4 B loads, nt stream", dict(spmv_kernel=1, spmv_delta=0, spmv_wide=0, spmv_nt=1, spmv_stream_nt=0)),
         ("16 B loads int32", dict(spmv_kernel=1, spmv_delta=0, spmv_wide=1, spmv_nt=0, spmv_stream_nt=0)),
         ("16 B loads int32, nt stream", dict(spmv_kernel=1, spmv_delta=0, spmv_wide=1, spmv_nt=0, spmv_stream_nt=1)),
         ("block-delta 8 bit", dict(spmv_kernel=1, spmv_delta=8, spmv_wide=1, spmv_nt=0, spmv_stream_nt=0)),
         ("block-delta 8 bit, nt stream", dict(spmv_kernel=1, spmv_delta=8, spmv_wide=1, spmv_nt=0, spmv_stream_nt=1)),
         ("staged rows int32", dict(spmv_kernel=4, spmv_delta=0, spmv_wide=0, spmv_nt=0, spmv_stream_nt=0))]
for links in (0, 1, 3):
    ref = None
    for form, opts in FORMS:
        for k, v in opts.items(): ctx.set_option(k, v)
        A = K.CsrMatrix.banded_random(ctx, n, links=links, seed=1)
        x = ctx.array(np.cos(np.arange(n) * 1e-3) + 0.5)
        y = ctx.zeros(n)
        A.matvec(x, y); ctx.sync()
        t = timeit(lambda: A.matvec(x, y))
        h = y.to_host()
        if ref is None: ref = h
        td = timeit(lambda: K.spmv_dot(A, x, y))
        alg, moved = A.spmv_bytes, A.spmv_bytes_stored
        print(json.dumps(dict(links=links, nnz_per_row=round(A.nnz / n, 2), form=form, ms=round(t * 1e3, 4), ms_fused_dot=round(td * 1e3, 4),
                              frac=round(alg / t / 8e12, 4), frac_moved=round(moved / t / 8e12, 4), delta_info=list(A.delta_info),
                              same=bool(np.array_equal(h, ref)))), flush=True)
        del A, x, y
ctx.close()
