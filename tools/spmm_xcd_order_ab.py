#!/usr/bin/env python3
"""Tile SpMM on operators WITHOUT a grid: which XCD gets which groups (round 6).  spmm_tile_exp bit 3 deals the 32-row groups
round-robin to the persistent workgroups (all eight XCDs sweep the matrix as ONE front) instead of giving XCD x the x-th eighth of
the groups.  On the banded + random operator the long-range columns reach anywhere inside a block of 2^20 rows: with one front the
block's panel rows (134 MB at p = 16) stay in the 256 MB Infinity Cache while the front crosses it; with eight fronts eight
blocks compete for it.  Timing + bit-equality with the direct-gather kernel, one JSON line per case."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)


def timed(A, X, Y, reps=10):
    K.spmm_(A, X, Y); ctx.sync(); best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps): K.spmm_(A, X, Y)
        ctx.sync(); best = min(best, (time.perf_counter() - t0) / reps)
    return best


p = 16
n = 10 * (1 << 20)
ops = [("banded + 3 links", lambda: K.CsrMatrix.banded_random(ctx, n, seed=1)),
       ("band only", lambda: K.CsrMatrix.banded_random(ctx, n, links=0, seed=1)),
       ("banded + 3 links + 4 dense rows", lambda: K.CsrMatrix.banded_random(ctx, n, seed=1, unsym=True, dense_rows=4)),
       ("banded hb = 4, 1 link", lambda: K.CsrMatrix.banded_random(ctx, n, half_band=4, links=1, seed=2))]
for name, make in ops:
    A = make()
    X, Y = K.Panel(ctx, A.n, p), K.Panel(ctx, A.n, p)
    h = np.zeros((K.panel_rows(A.n), p)); h[:A.n] = np.random.default_rng(0).standard_normal((A.n, p)); X.buf.copy_from_host(h.ravel())
    ctx.set_option("spmm_tile", 0); ctx.set_option("spmm_window", 0); K.spmm_(A, X, Y); ctx.sync(); ref = Y.buf.to_host()
    ctx.set_option("spmm_tile", 2); ctx.set_option("spmm_window", 1)
    alg = 12 * A.nnz + 4 * A.n + 16 * A.n * p
    # spmm_tile_xcd: 0 = XCD x takes the x-th eighth of the groups, 1 = round-robin (one front), 2 = one front in chunks of G / 8 per XCD
    for opts in ({}, {"spmm_tile_xcd": 0}, {"spmm_tile_xcd": 1}, {"spmm_tile_xcd": 2}, {"spmm_tile_xcd": 0}, {"spmm_tile_xcd": 1}, {"spmm_tile_xcd": 2},
                 {"spmm_tile_xcd": 1, "spmm_tile_pair": 0}):
        for k, v in opts.items(): ctx.set_option(k, v)
        t = timed(A, X, Y)
        print(json.dumps(dict(op=name, opts=opts, ms=round(t * 1e3, 4), frac=round(alg / t / 8e12, 4), same=bool(np.array_equal(ref, Y.buf.to_host())),
                              window=A.tile_info["window"])), flush=True)
        for k in opts: ctx.set_option(k, {"spmm_tile_xcd": -1, "spmm_tile_pair": 1, "spmm_tile_nt": 0}[k])
    del A, X, Y
ctx.close()
