#!/usr/bin/env python3
"""One rank's CG iteration at the N = 1 / 2 / 4 / 8 slab shapes of cfg 2 (512^3 row-partitioned), on ONE GPU, with the real
exchange code.

N = 1 is the plain single-GPU solve.  For N > 1 the rank owns 512 / N interior planes and runs the distributed path against a
ONE-rank RCCL communicator with test hook khip_test_set_halo_self: pack kernel -> grouped ncclSend / ncclRecv of the two boundary planes (to
itself) on the halo stream and communicator -> interior rows -> boundary ranges -> 16-byte all-gather of the (hi, lo) dot
partials + combine kernel.  Everything an interior rank of an N-GPU run launches is launched; what is missing is the xGMI
transfer itself (2 planes x 2 MB at >= 50 GB/s: <= 0.1 ms, overlapped with the interior rows) and the skew between ranks.  So
t(1) / t_slab(N) is a measured UPPER BOUND of the N-GPU speed-up, not a scaling curve.

  python tools/slab_iteration.py [--n1 512] [--iters 100] [--out gpurun_out/r05_slab_iteration.jsonl] [--only N]
Per-kernel times: run under `rocprofv3 --kernel-trace --stats` with --only N (tools/evidence.sh does).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import krylov_jl_amd as K  # noqa: E402


def run(n1, N, iters, fused, variant=0, opts=()):
    plane = n1 * n1
    ctx = K.Context(0)
    for k, v in opts:
        ctx.set_option(k, v)
    if N == 1:
        A = K.CsrMatrix.stencil(ctx, "poisson", n1)
        m = n1 ** 3
        info = {"rccl_ranks": 0, "halo_comm_separate": 0}
        halo = (0, 0, 0)
    else:
        ctx.test_set_halo_self(1)
        ctx.comm_init(0, 1, K.Context.comm_unique_id())
        info = ctx.comm_info()
        planes = n1 // N
        k0 = (n1 - planes) // 2                       # an interior slab: two neighbours
        r0, r1 = k0 * plane, (k0 + planes) * plane
        m = r1 - r0
        A = K.CsrMatrix.stencil(ctx, "poisson", n1, rows=(r0, r1), distributed=True)
        halo = A.halo_info
    b = ctx.empty(m)
    K.kfill_(b, 1.0)
    ws = K.CgWorkspace(ctx, m, m)
    K.cg_(ws, A, b, atol=0.0, rtol=0.0, itmax=10, fused=fused, variant=variant)          # warm-up (builds the coded stream)
    ctx.sync()
    best = None
    for _ in range(3):
        ctx.sync()
        t0 = time.perf_counter()
        K.cg_(ws, A, b, atol=0.0, rtol=0.0, itmax=iters, fused=fused, variant=variant, history=True)
        ctx.sync()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    st = ws.stats
    # the phases of one iteration from the HIP-event brackets bench.py reports per rank at N > 1 (khip_profile_kernels): pack kernel,
    # halo transfer on its stream, interior and boundary launch of the product, each dot's 16-byte all-gather + combine
    ctx.set_option("profile_spmv", 1); ctx.profile_kernels()
    K.cg_(ws, A, b, atol=0.0, rtol=0.0, itmax=iters, fused=fused, variant=variant)
    prof = ctx.profile_kernels(); ctx.set_option("profile_spmv", 0)
    it_p = max(ws.stats.niter, 1)
    phases = {k: {"launches_per_iteration": l / it_p, "avg_us": 1e3 * ms / l} for k, (l, ms) in prof.items() if l}
    bits, diags = A.code_info
    rec = {"n1": n1, "N": N, "planes": n1 // N, "rows": m, "iters": int(st.niter), "fused": fused, "variant": variant,
           "ms_per_iteration": 1e3 * best / max(st.niter, 1), "rccl_ranks": info["rccl_ranks"],
           "halo_comm_separate": info["halo_comm_separate"], "halo_entries_recv": halo[1], "halo_entries_sent": halo[2],
           "column_code_bits": bits, "spmv_bytes_algorithmic": A.spmv_bytes, "residual_last": float(st.residuals[-1]),
           "opts": dict(opts), "phases_hip_events": phases}
    ctx.close()
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n1", type=int, default=512)
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--fused", type=int, default=2)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r05_slab_iteration.jsonl"))
    ap.add_argument("--only", type=int, default=0)
    ap.add_argument("--variants", action="store_true", help="also single-reduction CG (variant 1) and overlap_halo = 0 per N")
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    recs = []
    for N in ([args.only] if args.only else [1, 2, 4, 8]):
        cases = [dict(variant=0, opts=())]
        if args.variants and N > 1:
            cases += [dict(variant=1, opts=()), dict(variant=0, opts=(("overlap_halo", 0),)), dict(variant=0, opts=(("comm_priority", 0),))]
        for c in cases:
            r = run(args.n1, N, args.iters, args.fused, **c)
            recs.append(r)
            print(json.dumps(r), flush=True)
    t1 = next((r["ms_per_iteration"] for r in recs if r["N"] == 1 and r["variant"] == 0), None)
    with open(args.out, "a") as f:
        for r in recs:
            if t1 and r["N"] > 1:
                r["speedup_upper_bound_t1_over_tslab"] = t1 / r["ms_per_iteration"]
            f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
