#!/usr/bin/env python3
"""Does the tile SpMM's time depend on WHICH panels it reads and writes?  In block_gmres! the product reads V[k] and writes V[k + 1]
(panels allocated one after the other); under rocprofv3 its launches spread from 1.18 to 1.31 ms while a loop over one fixed pair
of panels sits at 1.17-1.19 ms (profiles/r06g_*).  Times every (X = P[i], Y = P[j]) pair of seven separately allocated panels, and the
same with the panels' starts staggered by a few KiB."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
n1, p = 216, 16
A = K.CsrMatrix.stencil(ctx, "stencil27", n1)
n = A.n


def timed(X, Y, reps=8):
    K.spmm_(A, X, Y); ctx.sync(); best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps): K.spmm_(A, X, Y)
        ctx.sync(); best = min(best, (time.perf_counter() - t0) / reps)
    return best * 1e3


P = [K.Panel(ctx, n, p) for _ in range(7)]
base = [q.buf.ptr for q in P]
print(json.dumps({"panel_starts_mod_1MiB": [b % (1 << 20) for b in base], "gaps": [base[i + 1] - base[i] for i in range(6)]}))
row = {}
for i in range(6):
    row[f"P{i}->P{i+1}"] = round(timed(P[i], P[i + 1]), 4)
row["P0->P6"] = round(timed(P[0], P[6]), 4)
row["P3->P0"] = round(timed(P[3], P[0]), 4)
print(json.dumps(row), flush=True)
# the solver's sequence: one product per pair, in order, timed as a whole (cold-ish panels each time)
ctx.sync(); t0 = time.perf_counter()
for rep in range(4):
    for i in range(6): K.spmm_(A, P[i], P[i + 1])
ctx.sync()
print(json.dumps({"sequence_of_24_products_ms_each": round((time.perf_counter() - t0) / 24 * 1e3, 4)}), flush=True)
# with another kernel between the products (a panel-sized axpy, as the Gram-Schmidt sweep leaves the caches)
ctx.sync(); t0 = time.perf_counter(); tk = 0.0
L = K.panel_rows(n) * p
for rep in range(4):
    for i in range(6):
        K.spmm_(A, P[i], P[i + 1])
        K.kaxpy_(L, 0.0, P[(i + 3) % 7].buf, P[(i + 4) % 7].buf)
ctx.sync(); tot = (time.perf_counter() - t0) / 24 * 1e3
t0 = time.perf_counter()
for rep in range(24): K.kaxpy_(L, 0.0, P[3].buf, P[4].buf)
ctx.sync(); ax = (time.perf_counter() - t0) / 24 * 1e3
print(json.dumps({"product_plus_axpy_ms": round(tot, 4), "axpy_alone_ms": round(ax, 4), "product_in_that_sequence_ms": round(tot - ax, 4)}), flush=True)
ctx.close()
