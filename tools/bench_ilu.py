#!/usr/bin/env python3
"""IC(0)/ILU(0) preconditioner on the device: setup time, time per application (graph replay vs plain launches),
and IC(0)-CG vs plain CG on get_div_grad(n1^3).  Usage: python tools/bench_ilu.py [n1 ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import krylov_jl_amd as K

ctx = K.Context(0)
for n1 in [int(a) for a in sys.argv[1:]] or [64, 128, 256]:
    n = n1 ** 3
    A = K.CsrMatrix.stencil(ctx, "poisson", n1)
    b = ctx.empty(n); K.kfill_(b, 1.0)
    t0 = time.perf_counter(); P = K.Ilu0(A); ctx.sync(); t_setup = time.perf_counter() - t0
    x, y = ctx.empty(n), ctx.empty(n); K.kfill_(x, 1.0)
    res = {"n1": n1, "n": n, "nnz": A.nnz, "levels": P.levels, "setup_s": round(t_setup, 3)}
    for graph in (1, 0):
        K.lib().khip_ilu0_set_graph(K.C.byref(P.op), graph)
        P(x, y); ctx.sync()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps): P(x, y)
        ctx.sync()
        t = (time.perf_counter() - t0) / reps
        # bytes of one application: both triangles once (12 B per off-diagonal entry + 8 per diagonal) + x, y traffic
        bytes_apply = 12 * (A.nnz - n) + 8 * n + 4 * 6 * n + 8 * 5 * n
        res["apply_ms_graph" if graph else "apply_ms_launches"] = round(t * 1e3, 3)
        res["apply_gbps_graph" if graph else "apply_gbps_launches"] = round(bytes_apply / t / 1e9, 1)
    K.lib().khip_ilu0_set_graph(K.C.byref(P.op), 1)
    for name, M in (("cg", None), ("ic0_cg", P)):
        t0 = time.perf_counter()
        xs, st, ws = K.cg(A, b, M=M, rtol=1e-8, atol=0.0, itmax=20000, fused=2)
        ctx.sync()
        t = time.perf_counter() - t0
        res[name] = {"niter": st.niter, "solved": bool(st.solved), "seconds": round(t, 4), "ms_per_iter": round(1e3 * t / max(st.niter, 1), 3)}
    print(json.dumps(res), flush=True)
    del P, A
ctx.close()
