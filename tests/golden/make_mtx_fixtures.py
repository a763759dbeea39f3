"""Two tiny MatrixMarket fixtures for tools/bench_mtx.py (tests/test_bench_mtx.py):
  tiny_spd_sym.mtx    2-D 5-point Laplacian on a 6 x 7 grid + 0.5 I, `coordinate real symmetric` (lower triangle only, 1-based,
                      as SuiteSparse ships SPD matrices: the loader has to mirror it)
  tiny_unsym.mtx      a 30 x 30 nonsymmetric, diagonally dominant `coordinate real general` matrix, entries in shuffled order
Run: python tests/golden/make_mtx_fixtures.py"""
import os
import numpy as np
import scipy.sparse as sp
import scipy.io

HERE = os.path.dirname(os.path.abspath(__file__))
n1, n2 = 6, 7
T = lambda n: sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(n, n))
A = (sp.kron(sp.identity(n2), T(n1)) + sp.kron(T(n2), sp.identity(n1)) + 0.5 * sp.identity(n1 * n2)).tocoo()
L = sp.tril(A).tocoo()
with open(os.path.join(HERE, "tiny_spd_sym.mtx"), "w") as f:
    f.write("%%MatrixMarket matrix coordinate real symmetric\n% 2-D Laplacian 6 x 7 + 0.5 I (tests/golden/make_mtx_fixtures.py)\n")
    f.write(f"{A.shape[0]} {A.shape[1]} {L.nnz}\n")
    for i, j, v in zip(L.row, L.col, L.data):
        f.write(f"{i + 1} {j + 1} {float(v)!r}\n")
rng = np.random.default_rng(11)
n = 30
B = sp.random(n, n, density=0.15, random_state=5, format="coo")
B = (B + sp.diags(np.asarray(abs(B).sum(axis=1)).ravel() + 1.0)).tocoo()
perm = rng.permutation(B.nnz)
with open(os.path.join(HERE, "tiny_unsym.mtx"), "w") as f:
    f.write("%%MatrixMarket matrix coordinate real general\n")
    f.write(f"{n} {n} {B.nnz}\n")
    for k in perm:
        f.write(f"{B.row[k] + 1} {B.col[k] + 1} {float(B.data[k])!r}\n")
print("wrote tiny_spd_sym.mtx", A.shape, L.nnz, "and tiny_unsym.mtx", B.shape, B.nnz)
