#!/usr/bin/env python3
"""spmm_window_kernel vs spmm2_kernel: bit-equality on banded and scattered operators for several panel widths, then
timing on the 27-point 216^3 operator (cfg 5).  Usage: python tools/archive/spmm_window_check.py [n1] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp
import krylov_jl_amd as K

ctx = K.Context(0)


def panel(n, p, seed):
    X = K.Panel(ctx, n, p)
    h = np.zeros((K.panel_rows(n), p)); h[:n] = np.random.default_rng(seed).standard_normal((n, p))
    X.buf.copy_from_host(h.ravel())
    return X


def run(A, X, p, window):
    ctx.set_option("spmm_window", window)
    Y = K.Panel(ctx, A.m, p)
    K.spmm_(A, X, Y); ctx.sync()
    return Y.buf.to_host()


ok = True
mats = {"stencil27 40^3": K.CsrMatrix.stencil(ctx, "stencil27", 40), "poisson 37^3": K.CsrMatrix.stencil(ctx, "poisson", 37)}
S = (sp.random(5000, 5000, density=0.01, random_state=1, format="csr") + sp.eye(5000, format="csr")).tocsr(); S.sort_indices()
mats["random 5000 (scattered: direct path)"] = K.CsrMatrix.from_host(ctx, S.indptr.astype(np.int64), S.indices.astype(np.int32), S.data, S.shape)
for name, A in mats.items():
    for p in (2, 4, 6, 8, 12, 16, 24, 32, 64):
        X = panel(A.n, p, p)
        same = np.array_equal(run(A, X, p, 1), run(A, X, p, 0))
        ok &= same
        print(f"{name:40s} p={p:2d} window == direct: {same}")
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 216
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
A = K.CsrMatrix.stencil(ctx, "stencil27", n1)
for p in (16, 8, 4):
    Xb, Yb = K.Panel(ctx, A.n, p), K.Panel(ctx, A.n, p)
    ctx.set_option("spmm_window", 1); ctx.sync()
    t0 = time.perf_counter(); K.spmm_(A, Xb, Yb); ctx.sync(); t1 = time.perf_counter(); K.spmm_(A, Xb, Yb); ctx.sync(); t2 = time.perf_counter()
    print(f"p={p}: first SpMM (builds the window metadata) {1e3 * (t1 - t0):.1f} ms, next {1e3 * (t2 - t1):.2f} ms; device memory free {ctx.mem_info()[0] / 2**30:.2f} GiB")
    del Xb, Yb
    X, Y = K.Panel(ctx, A.n, p), K.Panel(ctx, A.n, p)
    K.kfill_(X.buf, 1.0)
    for window in (0, 1, 0, 1):
        ctx.set_option("spmm_window", window)
        K.spmm_(A, X, Y); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps): K.spmm_(A, X, Y)
        ctx.sync()
        dt = (time.perf_counter() - t0) / reps
        print(f"spmm p={p} n1={n1} window={window}: {dt*1e3:.3f} ms, {(12*A.nnz + 4*A.n + 16*A.n*p)/dt/1e9:.0f} GB/s algorithmic")
X, Y = K.Panel(ctx, A.n, 16), K.Panel(ctx, A.n, 16)
K.kfill_(X.buf, 1.0)
ctx.set_option("spmm_window", 1)
for grid in (256, 512, 768, 1024, 1536, 2048):
    ctx.set_option("spmm_window_grid", grid)
    K.spmm_(A, X, Y); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps): K.spmm_(A, X, Y)
    ctx.sync()
    print(f"p=16 window grid {grid}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms")
ctx.set_option("spmm_window_grid", 0)
print("ALL EQUAL" if ok else "MISMATCH")
ctx.close()
