#!/usr/bin/env python3
"""BASELINE cfg 4 (cg! on get_div_grad(1024^3), n = 2^30 rows, 7.5e9 nonzeros, row-partitioned over 8 ranks) run
FUNCTIONALLY on one MI355X: the 8 ranks are in-process contexts (khip_comm_init_local), 130 GB of HBM in total.
Not a performance number (the ranks time-share one GPU) -- it checks the 1-D partition at full size: global column
indices up to 2^30, 8 MiB halo planes, the [owned | ghost] renumbering, rank-ordered all-reduces.

Checks: ||b|| = sqrt(n) = 32768 exactly; sum(A * 1) = 6 n1^2 exactly (every missing Dirichlet neighbour counts once);
identical residual histories on all ranks; the true residual ||b - A x|| equals the recurrence's."""
import json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K

n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
halo_mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0          # 0 per operator, 1 neighbour exchange, 2 all-gather of x
n = n1 ** 3
starts = K.row_partition(n, world)
out, errs = [None] * world, []

def run(rank):
    try:
        c = K.Context(0)
        c.comm_init_local(rank, world, 4040)
        c.set_option("halo_mode", halo_mode)
        r0, r1 = starts[rank], starts[rank + 1]
        t0 = time.time()
        A = K.CsrMatrix.stencil(c, "poisson", n1, rows=(r0, r1), distributed=True)
        m = r1 - r0
        ones, y = c.empty(m), c.empty(m)
        K.kfill_(ones, 1.0)
        A.matvec(ones, y)
        s = K.kdot(m, ones, y)                      # all-reduced: sum over the GLOBAL vector
        bn = K.knorm(m, ones)
        ws = K.CgWorkspace(c, m, m)
        K.cg_(ws, A, ones, atol=0.0, rtol=0.0, itmax=steps, history=True, fused=2)
        hist = ws.stats.residuals.copy()
        r = c.empty(m)
        A.matvec(ws.x, r)
        K.kaxpby_(m, 1.0, ones, -1.0, r)
        true_res = K.knorm(m, r)
        out[rank] = dict(halo=A.halo_info, code=A.code_info, nnz=A.nnz, rows=m, sum_A1=s, bnorm=bn, hist=hist, true_res=true_res, setup_s=time.time() - t0)
        c.barrier()
        c.close()
    except Exception as e:      # noqa: BLE001
        import traceback
        errs.append((rank, repr(e), traceback.format_exc()))

ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
t0 = time.time()
[t.start() for t in ts]
[t.join() for t in ts]
assert not errs, errs
o = out[0]
res = {
    "config": f"cg! on get_div_grad({n1}^3) over {world} in-process ranks on ONE GPU (functional run of cfg 4)",
    "n": n, "nnz_total": int(sum(x["nnz"] for x in out)), "nnz_expected": 7 * n - 6 * n1 * n1,
    "sum_A_times_ones": o["sum_A1"], "sum_expected": 6.0 * n1 * n1, "norm_b": o["bnorm"], "norm_b_expected": float(np.sqrt(n)),
    "iterations": steps, "history_first_last": [float(o["hist"][0]), float(o["hist"][-1])],
    "history_identical_on_all_ranks": bool(all(np.array_equal(x["hist"], o["hist"]) for x in out)),
    "recurrence_vs_true_residual_rel": abs(o["true_res"] - float(o["hist"][-1])) / float(o["hist"][0]),
    "wall_s": round(time.time() - t0, 1),
}
res["halo_info_rank0"] = list(o["halo"]); res["code_info_rank0"] = list(o["code"])
oracle_golden = os.path.join(ROOT, "tests", "golden", "oracle_cfg2_cg512.json")
if n1 == 512 and os.path.exists(oracle_golden):   # the CPU oracle's history (tests/golden/make_scale_golden.py)
    g = json.load(open(oracle_golden))["residuals"]
    k = min(len(g), len(o["hist"]))
    res["iterations_compared_with_oracle"] = k - 1
    res["max_rel_dev_vs_cpu_oracle"] = max(abs(float(o["hist"][i]) - g[i]) / g[i] for i in range(k))
golden = os.path.join(ROOT, "tests", "golden", "cg512_residuals.json")
if n1 == 512 and os.path.exists(golden):          # the 1-GPU history bench.py's `parity` uses
    g = json.load(open(golden))["residuals"]
    k = min(len(g), len(o["hist"]))
    res["max_rel_dev_vs_1gpu_golden"] = max(abs(float(o["hist"][i]) - g[i]) / g[i] for i in range(k))
res["ok"] = bool(res["nnz_total"] == res["nnz_expected"] and res["sum_A_times_ones"] == res["sum_expected"]
                 and res["norm_b"] == res["norm_b_expected"] and res["history_identical_on_all_ranks"]
                 and res["recurrence_vs_true_residual_rel"] < 1e-10)
print(json.dumps(res), flush=True)
sys.exit(0 if res["ok"] else 1)
