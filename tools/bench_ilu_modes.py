import json, os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
for kind, n1 in (("poisson", 128), ("poisson", 256), ("stencil27", 128)):
    n = n1 ** 3
    A = K.CsrMatrix.stencil(ctx, kind, n1)
    x, y = ctx.empty(n), ctx.empty(n); K.kfill_(x, 1.0)
    res = {"op": kind, "n1": n1}
    ref = None
    for mode, name in ((1, "grid blocks"), (3, "level-sequence blocks"), (0, "level launches")):
        ctx.set_option("ilu_blocks", mode)
        t0 = time.perf_counter(); P = K.Ilu0(A); ctx.sync(); ts = time.perf_counter() - t0
        P(x, y); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(5): P(x, y)
        ctx.sync(); t = (time.perf_counter() - t0) / 5
        yh = y.to_host()
        if ref is None: ref = yh
        res[name] = {"apply_ms": round(t * 1e3, 3), "setup_s": round(ts, 2), "blocks": P.block_info()[1], "failed": P.block_info()[2], "same": bool(np.array_equal(yh, ref))}
        del P
    ctx.set_option("ilu_blocks", 1)
    print(json.dumps(res), flush=True)
    del A
