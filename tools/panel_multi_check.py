#!/usr/bin/env python3
"""X += sum_{i<k} V_i Y_i at 216^3 x 16 (cfg 5 panels): panel_multi_tiles sweep, GB/s of the (k + 2) panel passes."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
n, p = 216 ** 3, 16
rng = np.random.default_rng(0)
for k in (5, 2):
    M = np.cos(np.arange(n * p, dtype=np.float64) * 0.001).reshape(n, p)
    Vs = [K.Panel.from_host(ctx, M * (1.0 + 0.37 * j)) for j in range(k)]
    del M
    Ys = [rng.standard_normal((p, p)) * 1e-3 for _ in range(k)]
    X = K.Panel(ctx, n, p)
    ref = None
    for rnd in range(2):
        for tiles in (0, 1, 2, 4, 8, 16):
            ctx.set_option("panel_multi_tiles", tiles)
            if rnd == 0:                                  # bit-identity of every variant on random panels, from X = 0
                K.panel_multi_nn_(Vs, Ys, 0.0, X); K.panel_multi_nn_(Vs, Ys, 1.0, X)
                got = X.to_host()
                ref = got if ref is None else ref
                assert np.array_equal(got, ref), (k, tiles)
            K.panel_multi_nn_(Vs, Ys, 1.0, X); ctx.sync()
            t0 = time.perf_counter()
            for _ in range(10): K.panel_multi_nn_(Vs, Ys, 1.0, X)
            ctx.sync(); t = (time.perf_counter() - t0) / 10
            print(json.dumps(dict(k=k, panel_multi_tiles=tiles, ms=round(t * 1e3, 4), gbps=round((k + 2) * 8 * n * p / t / 1e9), frac=round((k + 2) * 8 * n * p / t / 8e12, 3))), flush=True)
    del Vs, X
ctx.close()
