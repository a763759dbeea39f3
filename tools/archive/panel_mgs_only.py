#!/usr/bin/env python3
"""A three-panel block Gram-Schmidt sweep at 216^3 x 16, a few launches (for rocprofv3 counter passes)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
for kv in filter(None, os.environ.get("KHIP_OPTS", "").split(",")):
    k, v = kv.split("="); ctx.set_option(k, int(v))
n, p = 216 ** 3, 16
V = [K.Panel(ctx, n, p) for _ in range(3)]
Q = K.Panel(ctx, n, p)
for v in V: K.kfill_(v.buf, 1e-4)
K.kfill_(Q.buf, 1.0)
K.panel_mgs_(V, Q); ctx.sync()
t0 = time.perf_counter()
for _ in range(3): K.panel_mgs_(V, Q)
ctx.sync()
print(f"sweep: {(time.perf_counter() - t0) / 3 * 1e3:.3f} ms")
ctx.close()
