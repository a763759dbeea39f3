#!/usr/bin/env python3
"""Q <- beta Q + alpha V Psi at 216^3 x 16 (k = 1 form of the LDS-factor kernel): in place (V = Q, beta = 0: the scaling of the
panel QR) and out of place (beta = 1: the Gram-Schmidt update), panel_multi_tiles sweep, bit-equality with the one-tile kernel."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
n, p = 216 ** 3, 16
rng = np.random.default_rng(0)
h = rng.standard_normal((n, p))
M = np.triu(rng.standard_normal((p, p))) + 4 * np.eye(p)
V = K.Panel.from_host(ctx, h)
ref = {}
for name, inplace in (("in place, beta = 0", True), ("out of place, beta = 1", False)):
    for tiles in (0, 1, 2, 4, 8, 16, 32, 64):
        ctx.set_option("panel_multi_tiles", tiles)
        Q = K.Panel.from_host(ctx, h)
        if inplace: K.panel_gemm_nn_(1.0, Q, M, 0.0, Q)
        else: K.panel_gemm_nn_(-1.0, V, M * 1e-3, 1.0, Q)
        ctx.sync()
        out = Q.to_host()
        if tiles == 0: ref[name] = out
        same = bool(np.array_equal(out, ref[name]))
        t0 = time.perf_counter()
        for _ in range(10):
            if inplace: K.panel_gemm_nn_(1.0, Q, np.eye(p), 0.0, Q)
            else: K.panel_gemm_nn_(-1.0, V, M * 1e-3, 1.0, Q)
        ctx.sync(); t = (time.perf_counter() - t0) / 10
        passes = 2 if inplace else 3
        print(json.dumps(dict(case=name, panel_multi_tiles=tiles, ms=round(t * 1e3, 4), gbps=round(passes * 8 * n * p / t / 1e9), same=same)), flush=True)
        del Q
ctx.close()
