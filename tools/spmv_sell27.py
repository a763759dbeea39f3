#!/usr/bin/env python3
"""The 27-point 216^3 operator (27 diagonals, 8-bit codes) through the SpMV: coded CSR kernel (64-row blocks: its LDS window) against the
sliced form (256-row blocks, one row per lane), plain and fused with a dot."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 216
A = K.CsrMatrix.stencil(ctx, "stencil27", n1)
n = A.n
x = ctx.array(np.cos(np.arange(n) * 0.37) + 0.1)
y = ctx.empty(n)
ref = None
for rnd in range(2):
    for sell in (0, 2):
        ctx.set_option("spmv_sell", sell)
        A.matvec(x, y); K.spmv_dot(A, x, y); ctx.sync()
        yh = y.to_host()
        ref = yh if ref is None else ref
        t0 = time.perf_counter()
        for _ in range(20): A.matvec(x, y)
        ctx.sync(); tp = (time.perf_counter() - t0) / 20
        t0 = time.perf_counter()
        for _ in range(20): K.spmv_dot(A, x, y)
        ctx.sync(); td = (time.perf_counter() - t0) / 20
        print(json.dumps({"operator": f"stencil27 {n1}^3", "kernel_choice": A.spmv_kernel_choice, "spmv_sell": sell, "sell_info": A.sell_info, "y_bit_identical": bool(np.array_equal(yh, ref)),
                          "plain_ms": round(1e3 * tp, 4), "fused_ms_incl_host_sync": round(1e3 * td, 4), "frac_algorithmic_plain": round(A.spmv_bytes / tp / 8e12, 3)}), flush=True)
ctx.close()
