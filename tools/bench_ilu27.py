#!/usr/bin/env python3
"""ILU(0) application on the 27-point operator (cfg 5's): block schedule (skewed blocks, general path) against level scheduling."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import krylov_jl_amd as K
ctx = K.Context(0)
for n1 in [int(a) for a in sys.argv[1:]] or [64, 128, 216]:
    n = n1 ** 3
    A = K.CsrMatrix.stencil(ctx, "stencil27", n1)
    x, y = ctx.empty(n), ctx.empty(n); K.kfill_(x, 1.0)
    res = {"n1": n1, "n": n, "nnz": A.nnz}
    for blocks in (1, 0):
        ctx.set_option("ilu_blocks", blocks)
        t0 = time.perf_counter(); P = K.Ilu0(A); ctx.sync(); ts = time.perf_counter() - t0
        P(x, y); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(5): P(x, y)
        ctx.sync(); t = (time.perf_counter() - t0) / 5
        key = "blocks" if blocks else "levels"
        res[key] = {"setup_s": round(ts, 3), "apply_ms": round(t * 1e3, 3), "info": P.block_info() if blocks else P.levels}
        if blocks: yb = y.to_host()
        else: res["same"] = bool((y.to_host() == yb).all())
        del P
    ctx.set_option("ilu_blocks", 1)
    print(json.dumps(res), flush=True)
    del A
ctx.close()
