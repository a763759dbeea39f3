#!/usr/bin/env python3
"""The non-stencil operators through the whole path on one MI355X: "banded + random, fixed seed"
(csrc/gen_irregular.cpp) with ~10 M rows and ~26 entries per row, far more than 2048 distinct diagonals, so
none of the stencil-specific mechanisms (diagonal column codes, grid tiles, grid-detected ILU blocks) applies; and its
nonsymmetric variant with four rows of 3000 more entries.  SpMV, fused SpMV + dot, SpMM p = 16, cg! / gmres!(30),
block_gmres! p = 16.  Roofline fractions on the ALGORITHMIC bytes of SURVEY.md 8d (12 nnz + 4 (m + 1) + 8 n + 8 m for a
product, + 16 n p for the SpMM), 8.0 TB/s.  Prints JSON lines (-> gpurun_out/bench_irregular.jsonl)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
out = open(os.path.join(ROOT, "gpurun_out", "bench_irregular.jsonl"), "a")
def emit(**kw):
    out.write(json.dumps(kw) + "\n"); out.flush(); print(json.dumps(kw), flush=True)

ctx = K.Context(0)
small = "--small" in sys.argv
n = (1 << 17) if small else 10 * (1 << 20)
PEAK = 8.0e12
reps = 20

def timeit(fn, reps=reps):
    fn(); ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.sync(); return (time.perf_counter() - t0) / reps

for name, kw in (("banded+random sym", dict(seed=1)), ("banded+random unsym + 4 dense rows", dict(seed=1, unsym=True, dense_rows=4))):
    t0 = time.perf_counter()
    A = K.CsrMatrix.banded_random(ctx, n, **kw)
    gen_s = time.perf_counter() - t0
    nnz = A.nnz
    emit(operator=name, n=n, nnz=nnz, nnz_per_row=nnz / n, generate_s=gen_s)
    x = ctx.array(np.cos(np.arange(n) * 1e-3) + 0.5)
    y = ctx.zeros(n)
    alg = A.spmv_bytes
    t = timeit(lambda: A.matvec(x, y))
    emit(operator=name, kernel="spmv", ms=t * 1e3, alg_bytes=alg, alg_tbps=alg / t / 1e12, frac=alg / t / PEAK, column_stream_bits=A.code_info[0],
         kernel_choice=A.spmv_kernel_choice if hasattr(A, "spmv_kernel_choice") else None)
    t = timeit(lambda: K.spmv_dot(A, x, y))
    emit(operator=name, kernel="spmv + x.Ax (one host-synchronised call each)", ms=t * 1e3, alg_bytes=alg, alg_tbps=alg / t / 1e12, frac=alg / t / PEAK)
    # SpMM p = 16
    p = 16
    X, Y = K.Panel(ctx, n, p), K.Panel(ctx, n, p)
    K.kfill_(X.buf, 1.0)
    algm = 12 * nnz + 4 * n + 16 * n * p
    for tile, window in ((1, 1), (0, 1), (0, 0)):
        ctx.set_option("spmm_tile", tile); ctx.set_option("spmm_window", window)
        t = timeit(lambda: K.spmm_(A, X, Y), reps=10)
        emit(operator=name, kernel="spmm p=16", spmm_tile=tile, spmm_window=window, ms=t * 1e3, alg_bytes=algm, alg_tbps=algm / t / 1e12,
             frac=algm / t / PEAK, tile_info=A.tile_info)
    ctx.set_option("spmm_tile", 1); ctx.set_option("spmm_window", 1)
    del X, Y
    b = ctx.zeros(n); A.matvec(x, b)                       # b = A x_true
    if not kw.get("unsym"):
        ws = K.CgWorkspace(ctx, n, n)
        K.cg_(ws, A, b, itmax=10, atol=0.0, rtol=0.0)
        ctx.sync(); t0 = time.perf_counter()
        K.cg_(ws, A, b, itmax=100, atol=0.0, rtol=0.0)
        ctx.sync(); dt = time.perf_counter() - t0
        it = ws.stats.niter
        emit(operator=name, solver="cg! 100 iterations (fused = 2)", ms_per_iter=1e3 * dt / it, iters_per_s=it / dt,
             fused_alg_bytes_per_iter=alg + 64 * n, alg_tbps=(alg + 64 * n) * it / dt / 1e12, frac=(alg + 64 * n) * it / dt / PEAK)
        ctx.sync(); t0 = time.perf_counter()
        K.cg_(ws, A, b, atol=0.0, rtol=1e-8, history=True)
        ctx.sync(); dt = time.perf_counter() - t0
        r = ctx.zeros(n); A.matvec(ws.x, r); K.kaxpby_(n, 1.0, b, -1.0, r)
        emit(operator=name, solver="cg! to rtol 1e-8", niter=ws.stats.niter, solved=ws.stats.solved, seconds=dt,
             true_rel_residual=K.knorm(n, r) / K.knorm(n, b))
        del ws
    else:
        ws = K.GmresWorkspace(ctx, n, n, memory=30)
        K.gmres_(ws, A, b, restart=True, itmax=30, atol=0.0, rtol=0.0)
        ctx.sync(); t0 = time.perf_counter()
        K.gmres_(ws, A, b, restart=True, itmax=90, atol=0.0, rtol=0.0)
        ctx.sync(); dt = time.perf_counter() - t0
        it = ws.stats.niter
        emit(operator=name, solver="gmres!(30) restart, 90 inner iterations", ms_per_inner_iter=1e3 * dt / it)
        ctx.sync(); t0 = time.perf_counter()
        K.gmres_(ws, A, b, restart=True, atol=0.0, rtol=1e-8)
        ctx.sync(); dt = time.perf_counter() - t0
        emit(operator=name, solver="gmres!(30) to rtol 1e-8", niter=ws.stats.niter, solved=ws.stats.solved, seconds=dt)
        del ws
    # block_gmres! p = 16
    t_ = (np.arange(n) + 1.0) / n
    Xt = np.stack([np.cos(j * np.pi * t_) + 0.1 * j for j in range(p)], axis=1)
    dXt = K.Panel.from_host(ctx, Xt)
    dB = K.Panel(ctx, n, p)
    K.spmm_(A, dXt, dB)
    Bd = ctx.array(np.asfortranarray(dB.to_host()).ravel(order="F"))
    del dXt, dB
    ws = K.BlockGmresWorkspace(ctx, n, n, p, memory=5)
    K.block_gmres_(ws, A, Bd, restart=True, itmax=5, atol=0.0, rtol=0.0)
    ctx.sync(); t0 = time.perf_counter()
    K.block_gmres_(ws, A, Bd, restart=True, itmax=20, atol=0.0, rtol=0.0)
    ctx.sync(); dt = time.perf_counter() - t0
    emit(operator=name, solver="block_gmres! p=16 memory=5, 20 iterations", ms_per_iter=1e3 * dt / ws.stats.niter)
    ctx.sync(); t0 = time.perf_counter()
    try:
        K.block_gmres_(ws, A, Bd, restart=True, atol=0.0, rtol=1e-8, itmax=400)
        ctx.sync(); dt = time.perf_counter() - t0
        emit(operator=name, solver="block_gmres! to rtol 1e-8 (itmax 400)", niter=ws.stats.niter, solved=ws.stats.solved, seconds=dt,
             max_err=float(np.abs(ws.X - Xt).max()))
    except K.KhipError as e:
        emit(operator=name, solver="block_gmres! to rtol 1e-8", error=str(e))
    del ws, A, x, y, b, Bd
ctx.close()
