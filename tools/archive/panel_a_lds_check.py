#!/usr/bin/env python3
"""Fused Gram-Schmidt step (panel_nn_tn_kernel) with the A operand staged through LDS (panel_a_lds = 1) vs loaded in operand
layout (0): bit-equality of Q and every Psi block, then timing at 216^3 x 16."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
ok = True
for n, p, k in [(1000, 16, 1), (4099, 16, 3), (100000, 16, 4), (257, 16, 2)]:
    rng = np.random.default_rng(n + k)
    Vall = np.linalg.qr(rng.standard_normal((n, p * k)))[0]
    Vh = [np.ascontiguousarray(Vall[:, i * p:(i + 1) * p]) for i in range(k)]
    Qh = rng.standard_normal((n, p))
    res = []
    for a in (1, 0):
        ctx.set_option("panel_a_lds", a)
        ctx.set_option("panel_nt", 2 * a)
        V = [K.Panel.from_host(ctx, v) for v in Vh]
        Q = K.Panel.from_host(ctx, Qh)
        blocks = K.panel_mgs_(V, Q)
        R = K.panel_qr_(Q)
        res.append((Q.to_host(), np.array(blocks), R))
    same = all(np.array_equal(x, y) for x, y in zip(res[0], res[1]))
    ok &= same
    print(json.dumps(dict(n=n, k=k, same=bool(same))), flush=True)
n, p = 216 ** 3, 16
V = [K.Panel(ctx, n, p) for _ in range(4)]
Q = K.Panel(ctx, n, p)
for v in V: K.kfill_(v.buf, 1e-4)
ctx.set_option("panel_a_lds", 1)
for k in (1, 3):
    for a in (0, 2, 0, 2):
        ctx.set_option("panel_nt", a)
        K.kfill_(Q.buf, 1.0); K.panel_mgs_(V[:k], Q); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(10): K.panel_mgs_(V[:k], Q)
        ctx.sync()
        dt = (time.perf_counter() - t0) / 10
        print(json.dumps(dict(sweep_panels=k, panel_nt=a, ms=round(dt * 1e3, 4))), flush=True)
print("ALL EQUAL" if ok else "MISMATCH")
ctx.close()
