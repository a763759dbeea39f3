#!/usr/bin/env python3
"""khip_panel_mgs: fused sweep (panel_nn_tn_kernel) vs the two-kernel sequence -- bit-equality of Q and of every Psi block,
then timing on 216^3 x 16 panels (cfg 5).  Usage: python tools/archive/panel_mgs_check.py [n1] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import krylov_jl_amd as K

ctx = K.Context(0)
ok = True
for n, p, k in [(1000, 16, 1), (1000, 16, 4), (4099, 16, 3), (777, 8, 5), (5000, 32, 3), (300, 5, 2), (100000, 16, 6), (257, 24, 2)]:
    rng = np.random.default_rng(n + p + k)
    Vall = np.linalg.qr(rng.standard_normal((n, p * k)))[0]                  # k mutually orthonormal panels
    Vh = [np.ascontiguousarray(Vall[:, i * p:(i + 1) * p]) for i in range(k)]
    Qh = rng.standard_normal((n, p))
    res = []
    for fuse in (1, 0):
        ctx.set_option("panel_fuse", fuse)
        V = [K.Panel.from_host(ctx, v) for v in Vh]
        Q = K.Panel.from_host(ctx, Qh)
        blocks = K.panel_mgs_(V, Q)
        blocks2 = K.panel_mgs_(V, Q, accumulate_into=blocks)          # second sweep, accumulated (reorthogonalisation)
        res.append((Q.to_host(), np.array(blocks), np.array(blocks2)))
    same = all(np.array_equal(a, b) for a, b in zip(res[0], res[1]))
    # against numpy (tolerance): Q_final orthogonal to every V_i
    orth = max(np.abs(v.T @ res[0][0]).max() for v in Vh)
    ok &= same and orth < 1e-10
    print(f"n={n:6d} p={p:2d} k={k}: fused == two-kernel sequence: {same}; max |V_i' Q| after two sweeps {orth:.1e}")
ctx.set_option("panel_fuse", 1)
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 216
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n, p = n1 ** 3, 16
V = [K.Panel(ctx, n, p) for _ in range(5)]
Q = K.Panel(ctx, n, p)
for v in V: K.kfill_(v.buf, 1e-4)
for k in (1, 3, 5):
    for fuse in (0, 1, 0, 1):
        ctx.set_option("panel_fuse", fuse)
        K.kfill_(Q.buf, 1.0); K.panel_mgs_(V[:k], Q); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps): K.panel_mgs_(V[:k], Q)
        ctx.sync()
        dt = (time.perf_counter() - t0) / reps
        print(f"sweep over k={k} panels of {n1}^3 x {p}, fuse={fuse}: {dt*1e3:.3f} ms ({dt*1e3/k:.3f} per step)")
print("ALL EQUAL" if ok else "MISMATCH")
ctx.close()
