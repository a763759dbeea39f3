#!/usr/bin/env python3
"""Would 2-D row tiles pay for the SpMM window kernel?  The 27-point operator on n1^3 points, once in the natural numbering
(a group of 32 consecutive rows = 32 points of one grid line: 9 x 34 = 306 distinct panel rows) and once renumbered so that 32
consecutive rows are an 8 x 4 tile of a grid plane (10 x 6 x 3 = 180 distinct panel rows).  Same kernel, same nonzeros."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp
import krylov_jl_amd as K

n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 128
tx, ty = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (8, 4)
p = 16
t0 = time.time()
T = sp.diags([np.ones(n1 - 1), np.ones(n1), np.ones(n1 - 1)], [-1, 0, 1], format="csr")
A = sp.kron(sp.kron(T, T, format="csr"), T, format="csr")
A.data = np.random.default_rng(0).standard_normal(A.nnz)
n = n1 ** 3
z, y, x = np.meshgrid(np.arange(n1), np.arange(n1), np.arange(n1), indexing="ij")
new = (((z * (n1 // ty) + y // ty) * (n1 // tx) + x // tx) * (tx * ty) + (y % ty) * tx + x % tx).ravel()   # old -> new
perm = np.empty(n, dtype=np.int64); perm[new] = np.arange(n)                                               # new -> old
Ap = A[perm][:, perm].tocsr(); Ap.sort_indices()
print(f"built {n1}^3 ({A.nnz} nonzeros) and its {tx} x {ty} tile renumbering in {time.time() - t0:.1f} s", flush=True)
ctx = K.Context(0)
for name, M in (("natural", A), (f"{tx}x{ty} tiles", Ap)):
    H = K.CsrMatrix.from_host(ctx, M.indptr.astype(np.int64), M.indices.astype(np.int32), M.data, M.shape)
    X, Y = K.Panel(ctx, n, p), K.Panel(ctx, n, p)
    h = np.random.default_rng(1).standard_normal(K.panel_rows(n) * p); h[n * p:] = 0
    X.buf.copy_from_host(h)
    for window in (1, 0):
        ctx.set_option("spmm_window", window)
        K.spmm_(H, X, Y); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(10): K.spmm_(H, X, Y)
        ctx.sync(); dt = (time.perf_counter() - t0) / 10
        print(f"{name:12s} window={window}: {dt * 1e3:.3f} ms  {dt / n * 1e12:.0f} ps/row", flush=True)
    del H, X, Y
ctx.close()
