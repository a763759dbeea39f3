#!/usr/bin/env python3
"""Bandwidth sweep of the hot-path kernels on one MI355X (run through gpurun; results ->
gpurun_out/sweep.jsonl).  Algorithmic bytes / host-timed launch loops (sync before and after)."""
import argparse
import itertools
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import krylov_jl_amd as K  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n1", type=int, default=512)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "sweep.jsonl"))
ap.add_argument("--quick", action="store_true")
args = ap.parse_args()
os.makedirs(os.path.dirname(args.out), exist_ok=True)
fout = open(args.out, "a")


def emit(**kw):
    fout.write(json.dumps(kw) + "\n")
    fout.flush()
    print(kw, flush=True)


def timeit(fn, reps):
    fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps


ctx = K.Context(0)
n1 = args.n1
n = n1 ** 3
A = K.CsrMatrix.stencil(ctx, "poisson", n1)
x, y, z, w = (ctx.empty(n) for _ in range(4))
K.kfill_(x, 1.0); K.kfill_(y, 0.5); K.kfill_(z, 0.25); K.kfill_(w, 2.0)
GB = 1e9

# ---- BLAS-1 ----
ops = {
    "copy(16n)": (16, lambda: K.kcopy_(n, y, x)),
    "fill(8n)": (8, lambda: K.kfill_(y, 0.5)),
    "axpy(24n)": (24, lambda: K.kaxpy_(n, 1e-9, x, y)),
    "axpby(24n)": (24, lambda: K.kaxpby_(n, 1e-9, x, 1.0, y)),
    "dot(16n)": (16, lambda: K.kdot(n, x, y)),
    "nrm2(8n)": (8, lambda: K.knorm(n, x)),
    "axpy2_dot(48n)": (48, lambda: K.axpy2_dot(n, 1e-9, x, y, z, w)),
    "waxpy(24n)": (24, lambda: K.waxpy_(n, z, x, 1e-9, y)),
}
for name, (bpe, fn) in ops.items():
    for comp in ((1, 0) if "dot" in name or "nrm2" in name else (1,)):
        ctx.set_option("compensated", comp)
        t = timeit(fn, args.reps)
        emit(kernel=name, n=n, compensated=comp, ms=t * 1e3, gbps=bpe * n / t / GB)
ctx.set_option("compensated", 1)
for nt_min in (1 << 30, 1 << 22):
    ctx.set_option("nt_min_elems", nt_min)
    t = timeit(lambda: K.kaxpy_(n, 1e-9, x, y), args.reps)
    emit(kernel="axpy(24n)", nt=(nt_min <= n), ms=t * 1e3, gbps=24 * n / t / GB)
    t = timeit(lambda: K.kdot(n, x, y), args.reps)
    emit(kernel="dot(16n)", nt=(nt_min <= n), ms=t * 1e3, gbps=16 * n / t / GB)

# ---- SpMV variants ----
sb = A.spmv_bytes


def spmv_case(label, **opts):
    for k, v in opts.items():
        ctx.set_option(k, v)
    t = timeit(lambda: A.matvec(x, y), args.reps)
    emit(kernel="spmv", label=label, **opts, ms=t * 1e3, gbps=sb / t / GB, frac_of_8TBs=sb / t / 8e12)
    t = timeit(lambda: K.spmv_dot(A, x, y), args.reps)
    emit(kernel="spmv+dot", label=label, **opts, ms=t * 1e3, gbps=sb / t / GB)


for lanes in (8, 4):
    for nt in (1, 0):
        spmv_case("ordered flat", spmv_kernel=3, spmv_lanes=lanes, spmv_nt=nt, spmv_persist=0)
spmv_case("ordered persistent", spmv_kernel=3, spmv_lanes=8, spmv_nt=1, spmv_persist=1)
for vec in (1, 2):
    for nt in (1, 0):
        spmv_case("stream flat", spmv_kernel=1, spmv_rows=256, spmv_vec=vec, spmv_nt=nt, spmv_persist=0)
spmv_case("stream flat xcd", spmv_kernel=1, spmv_rows=256, spmv_vec=1, spmv_nt=0, spmv_persist=0, spmv_xcd=1)
ctx.set_option("spmv_xcd", 0)
spmv_case("stream persistent", spmv_kernel=1, spmv_rows=256, spmv_vec=1, spmv_nt=1, spmv_persist=1)
spmv_case("stream persistent xcd", spmv_kernel=1, spmv_rows=256, spmv_vec=1, spmv_nt=1, spmv_persist=1, spmv_xcd=1)
ctx.set_option("spmv_xcd", 0)
for lanes in (4, 8):
    spmv_case("vector flat", spmv_kernel=2, spmv_lanes=lanes, spmv_persist=0)
for k, v in dict(spmv_kernel=0, spmv_lanes=0, spmv_rows=256, spmv_vec=1, spmv_nt=0, spmv_persist=0).items():
    ctx.set_option(k, v)

# ---- CG iteration, fused vs unfused ----
b = ctx.empty(n)
K.kfill_(b, 1.0)
ws = K.CgWorkspace(ctx, n, n)
for fused in (1, 0):
    K.cg_(ws, A, b, atol=0.0, rtol=0.0, itmax=5, fused=bool(fused))
    ctx.sync()
    t0 = time.perf_counter()
    K.cg_(ws, A, b, atol=0.0, rtol=0.0, itmax=40, fused=bool(fused))
    ctx.sync()
    dt = (time.perf_counter() - t0) / 40
    emit(kernel="cg_iteration", fused=fused, ms=dt * 1e3, its=1 / dt,
         gbps_algorithmic=(sb + (64 if fused else 104) * n) / dt / GB)      # the bytes the path that ran moves, never the other one's
ctx.close()
