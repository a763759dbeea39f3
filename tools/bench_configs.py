#!/usr/bin/env python3
"""Timings of the other BASELINE.json configs on one MI355X (not the bench.py headline):
cfg 3: gmres! restart, memory 30, kron_unsymmetric 256^3;  cfg 5: block_gmres! p = 16 on the 27-point 216^3 operator;
plus bicgstab! on cfg 3's operator.  Prints JSON lines (-> gpurun_out/bench_configs.jsonl)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K

out = open(os.path.join(ROOT, "gpurun_out", "bench_configs.jsonl"), "a")
def emit(**kw):
    out.write(json.dumps(kw) + "\n"); out.flush(); print(json.dumps(kw), flush=True)

ctx = K.Context(0)
small = "--small" in sys.argv

# ---- cfg 3: GMRES(30) ----
n1 = 64 if small else 256
n = n1 ** 3
A = K.CsrMatrix.stencil(ctx, "kron_unsymmetric", n1)
ones = ctx.empty(n); K.kfill_(ones, 1.0)
b = ctx.empty(n); A.matvec(ones, b)                       # b = A * ones (test/test_utils.jl:166-167)
for fused in (True, False):
    ws = K.GmresWorkspace(ctx, n, n, memory=30)
    K.gmres_(ws, A, b, restart=True, itmax=30, fused=fused, atol=0.0, rtol=0.0)     # warm-up cycle
    ctx.sync(); t0 = time.perf_counter()
    K.gmres_(ws, A, b, restart=True, itmax=90, fused=fused, atol=0.0, rtol=0.0, history=True)
    ctx.sync(); dt = time.perf_counter() - t0
    st = ws.stats
    # bytes per 30-step cycle as the reference issues it: sum_k [SpMV + k*(16n+24n) + 8n + 16n] + restart work
    sb = A.spmv_bytes
    cyc = sum(sb + k * 40 * n + 24 * n for k in range(1, 31)) + 30 * 8 * n + 30 * 24 * n + sb + 24 * n + 16 * n
    # a bandwidth only for the UNFUSED run: `cyc` counts the bytes of the reference's primitive sequence, which the fused
    # path does not move (it used to print 8.3 TB/s for it)
    extra = {} if fused else dict(gbps_reference_sequence=cyc * (st.niter / 30) / dt / 1e9)
    emit(config="cfg3 gmres(30) restart kron_unsymmetric %d^3" % n1, fused=fused, iters=st.niter, seconds=dt,
         ms_per_inner_iter=1e3 * dt / st.niter, inner_iters_per_s=st.niter / dt, last_residual=float(st.residuals[-1]), **extra)
    del ws
# a full solve to the default tolerance
ws = K.GmresWorkspace(ctx, n, n, memory=30)
ctx.sync(); t0 = time.perf_counter()
K.gmres_(ws, A, b, restart=True, history=True)
ctx.sync(); dt = time.perf_counter() - t0
st = ws.stats
x = ws.x
r = ctx.empty(n); A.matvec(x, r); K.kaxpby_(n, 1.0, b, -1.0, r)
emit(config="cfg3 full solve", niter=st.niter, solved=st.solved, seconds=dt, true_rel_residual=K.knorm(n, r) / K.knorm(n, b))
del ws
# ---- BiCGSTAB on the same operator ----
for fused in (True, False):
    ws = K.BicgstabWorkspace(ctx, n, n)
    K.bicgstab_(ws, A, b, itmax=5, fused=fused, atol=0.0, rtol=0.0)
    ctx.sync(); t0 = time.perf_counter()
    K.bicgstab_(ws, A, b, itmax=40, fused=fused, atol=0.0, rtol=0.0)
    ctx.sync(); dt = time.perf_counter() - t0
    it = ws.stats.niter
    ref_bytes = 2 * A.spmv_bytes + (4 * 16 + 8 + 5 * 24 + 24 + 3 * 16) * n          # the reference's primitive sequence
    fused_bytes = 2 * A.spmv_bytes + 128 * n                                         # the five fused passes
    emit(config="bicgstab kron_unsymmetric %d^3" % n1, fused=fused, iters=it, ms_per_iter=1e3 * dt / it,
         gbps_algorithmic=(fused_bytes if fused else ref_bytes) * it / dt / 1e9)
    del ws
del A, b, ones

# ---- cfg 5: block-GMRES p = 16 ----
n1 = 48 if small else 216
n = n1 ** 3
p = 16
A = K.CsrMatrix.stencil(ctx, "stencil27", n1)
t = (np.arange(n) + 1.0) / n
Xt = np.stack([np.cos(j * np.pi * t) + 0.1 * j for j in range(p)], axis=1)
dXt = K.Panel.from_host(ctx, Xt)
dB = K.Panel(ctx, n, p)
K.spmm_(A, dXt, dB)
Bh = dB.to_host()
ws = K.BlockGmresWorkspace(ctx, n, n, p, memory=5)
Bd = ctx.array(np.asfortranarray(Bh).ravel(order="F"))
K.block_gmres_(ws, A, Bd, restart=True, itmax=5, atol=0.0, rtol=0.0)
ctx.sync(); t0 = time.perf_counter()
K.block_gmres_(ws, A, Bd, restart=True, itmax=20, atol=0.0, rtol=0.0, history=True)
ctx.sync(); dt = time.perf_counter() - t0
st = ws.stats
panel = 8 * n * p
emit(config="cfg5 block_gmres p=16 stencil27 %d^3" % n1, iters=st.niter, seconds=dt, ms_per_iter=1e3 * dt / st.niter,
     nnz=A.nnz, panel_bytes=panel, last_residual=float(st.residuals[-1]), first_residual=float(st.residuals[0]))
ctx.sync(); t0 = time.perf_counter()
K.block_gmres_(ws, A, Bd, restart=True, history=True)
ctx.sync(); dt = time.perf_counter() - t0
st = ws.stats
X = ws.X
emit(config="cfg5 full solve", niter=st.niter, solved=st.solved, seconds=dt, max_err=float(np.abs(X - Xt).max()))
# panel kernel micro-timings
V, Q = K.Panel.from_host(ctx, Xt), K.Panel.from_host(ctx, Bh)
def timeit(fn, reps=10):
    fn(); ctx.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.sync(); return (time.perf_counter() - t0) / reps
tt = timeit(lambda: K.panel_gemm_tn(V, Q)); emit(kernel="panel_gemm_tn", ms=tt * 1e3, gbps=2 * panel / tt / 1e9)
M = np.eye(p) * 1e-3
tt = timeit(lambda: K.panel_gemm_nn_(-1.0, V, M, 1.0, Q)); emit(kernel="panel_gemm_nn", ms=tt * 1e3, gbps=3 * panel / tt / 1e9)
Y = K.Panel(ctx, n, p)
tt = timeit(lambda: K.spmm_(A, V, Y)); emit(kernel="spmm p=16", ms=tt * 1e3, gbps=(12 * A.nnz + 4 * n + 2 * panel) / tt / 1e9)
tt = timeit(lambda: K.panel_qr_(Q), reps=3); emit(kernel="panel_qr (CholQR2)", ms=tt * 1e3, gbps=5 * panel / tt / 1e9,
                                                           bytes_note="5 panel passes: Gram | scale + Gram fused (read, write) | scale (read, write)")
ctx.close()
