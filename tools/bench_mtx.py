#!/usr/bin/env python3
"""MatrixMarket files through the HIP path: the reference's benchmark loop over SuiteSparse matrices
(benchmark/cg_bmark.jl:29-54: `A = MatrixMarket.mmread(path); b = ones(n); cg(A, b, atol = 0, rtol = 1e-6, itmax = n)`;
benchmark/gpu.jl:15-47: cg for symmetric positive definite, bicgstab for nonsymmetric matrices, rtol = 1e-8) for whoever
holds the files -- this image has no network, so none are shipped; tests/golden/tiny_*.mtx are two fixtures.

  python tools/bench_mtx.py [--rtol 1e-8] [--solver auto|cg|bicgstab|gmres] [--oracle] file.mtx | directory ...

Per matrix one JSON line: shape, entries, the SpMV kernel the handle takes and its column stream (dictionary codes /
block-delta codes / int32), SpMV time and fraction of the 8 TB/s HBM peak on the algorithmic bytes (12 nnz + 4 (m + 1) +
8 n + 8 m) -- the rows tools/bench_irregular.py prints for the synthetic operators -- then the solve: iterations,
seconds, status, true relative residual.  --oracle also runs the CPU oracle's solver on the same system and reports its
iteration count (small matrices only).  --dry: load and describe only (no GPU)."""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def load_mtx(path):
    """scipy.io.mmread -> CSR with sorted columns, duplicates summed (`symmetric` / `skew-symmetric` files are mirrored by
    mmread); returns (csr, symmetric_pattern_and_values)."""
    import scipy.io
    import scipy.sparse as sp
    M = scipy.io.mmread(path)
    S = sp.csr_matrix(M, dtype=np.float64)
    S.sum_duplicates()
    S.sort_indices()
    sym = S.shape[0] == S.shape[1] and (abs(S - S.T)).nnz == 0
    return S, bool(sym)


def describe(path, S, sym):
    m, n = S.shape
    rl = np.diff(S.indptr)
    r_of = np.repeat(np.arange(m), rl)
    band = int(np.abs(S.indices - r_of).max()) if S.nnz else 0
    return dict(matrix=os.path.basename(path), rows=m, cols=n, nnz=int(S.nnz), symmetric=sym, mean_row=float(rl.mean()) if m else 0.0,
                max_row=int(rl.max()) if m else 0, bandwidth=band, distinct_diagonals=int(np.unique(S.indices - r_of).size) if S.nnz else 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("paths", nargs="+")
    ap.add_argument("--rtol", type=float, default=1e-8)
    ap.add_argument("--solver", default="auto", choices=["auto", "cg", "bicgstab", "gmres"])
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--dry", action="store_true")
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    files = []
    for p in args.paths:
        files += sorted(glob.glob(os.path.join(p, "**", "*.mtx"), recursive=True)) if os.path.isdir(p) else [p]
    if not files:
        print("bench_mtx: no .mtx file found", file=sys.stderr)
        return 2
    ctx = K = None
    if not args.dry:
        import krylov_jl_amd as K
        ctx = K.Context(0)
    for path in files:
        S, sym = load_mtx(path)
        row = describe(path, S, sym)
        m, n = S.shape
        if m != n:
            row["skipped"] = "not square"
            print(json.dumps(row), flush=True)
            continue
        solver = args.solver if args.solver != "auto" else ("cg" if sym else "bicgstab")
        row["solver"] = solver
        if args.oracle:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle as ok
            Ao = ok.CsrMatrix.from_arrays(S.indptr.astype(np.int64), S.indices.astype(np.int32), S.data)
            f = {"cg": ok.cg, "bicgstab": ok.bicgstab, "gmres": lambda A, b, **kw: ok.gmres(A, b, memory=30, restart=True, **kw)}[solver]
            r = f(Ao, np.ones(n), atol=0.0, rtol=args.rtol, itmax=n)
            row["oracle"] = dict(niter=r.niter, solved=r.solved, status=r.status)
        if not args.dry:
            A = K.CsrMatrix.from_scipy(ctx, S)
            x = ctx.array(np.cos(np.arange(n) * 1e-3) + 0.5)
            y = ctx.zeros(n)
            A.matvec(x, y); ctx.sync()                    # builds the column stream the handle takes
            t0 = time.perf_counter()
            for _ in range(args.reps):
                A.matvec(x, y)
            ctx.sync()
            t = (time.perf_counter() - t0) / args.reps
            alg = A.spmv_bytes
            row.update(spmv_ms=1e3 * t, alg_bytes=alg, frac=alg / t / 8e12, moved_bytes=A.spmv_bytes_stored,
                       column_stream=dict(dictionary_bits=A.code_info[0], diagonals=A.code_info[1], delta_bits=A.delta_info[0],
                                          delta_rows=A.delta_info[1], escapes=A.delta_info[2]))
            b = ctx.empty(n)
            K.kfill_(b, 1.0)
            t0 = time.perf_counter()
            if solver == "cg":
                xs, st, _ = K.cg(A, b, atol=0.0, rtol=args.rtol, itmax=n)
            elif solver == "bicgstab":
                xs, st, _ = K.bicgstab(A, b, atol=0.0, rtol=args.rtol, itmax=n)
            else:
                xs, st, _ = K.gmres(A, b, memory=30, restart=True, atol=0.0, rtol=args.rtol, itmax=n)
            ctx.sync()
            dt = time.perf_counter() - t0
            r = ctx.zeros(n)
            A.matvec(xs, r)
            K.kaxpby_(n, 1.0, b, -1.0, r)
            row.update(niter=st.niter, solved=bool(st.solved), status=st.status, solve_s=dt, iters_per_s=st.niter / dt if dt > 0 else None,
                       true_rel_residual=K.knorm(n, r) / K.knorm(n, b))
            del A, x, y, b, r, xs
        print(json.dumps(row), flush=True)
    if ctx is not None:
        ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
