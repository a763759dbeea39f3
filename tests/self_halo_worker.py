"""One process, ONE RCCL rank that exchanges its halo with ITSELF (test hook khip_test_set_halo_self, csrc/comm.cpp): a slab of planes of the
7-point grid whose off-slab columns wrap onto the slab (a periodic slab), against the same periodic operator as a plain
single-GPU CSR.  Driven by tests/test_gpu_self_halo.py (own process: a hang of the self Send/Recv must not take the suite down).
argv: n1 k0 k1 out.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import krylov_jl_amd as K  # noqa: E402


def main():
    n1, k0, k1, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    plane = n1 * n1
    r0, r1 = k0 * plane, k1 * plane
    m = r1 - r0
    res = {}
    ctx = K.Context(0)
    ctx.test_set_halo_self(1)                    # before the communicator: it splits off the halo communicator for one rank too
    ctx.comm_init(0, 1, K.Context.comm_unique_id())
    info = ctx.comm_info()
    res["rccl_ranks"] = info["rccl_ranks"]
    res["halo_comm_separate"] = info["halo_comm_separate"]
    A = K.CsrMatrix.stencil(ctx, "poisson", n1, rows=(r0, r1), distributed=True)
    gm, ng, ns = A.halo_info
    res["gather_mode"], res["n_ghost"], res["n_send"] = gm, ng, ns
    # the same periodic slab as an ordinary operator on a context without a communicator
    ctx2 = K.Context(0)
    rp, col, val = K.gen_stencil_arrays(ctx2, "poisson", n1, rows=(r0, r1))
    colp = ((col.astype(np.int64) - r0) % m).astype(np.int32)
    P = K.CsrMatrix.from_host(ctx2, rp, colp, val, (m, m))
    rng = np.random.default_rng(3)
    xh = rng.standard_normal(m)
    ys = A.matvec(ctx.array(xh)).to_host()
    yp = P.matvec(ctx2.array(xh)).to_host()
    res["spmv_bit_identical"] = bool(np.array_equal(ys, yp))
    res["spmv_uses_halo"] = bool(not np.array_equal(ys[:plane], (K.CsrMatrix.from_host(ctx2, rp, np.clip(col - r0, 0, m - 1).astype(np.int32), val, (m, m))
                                                                  .matvec(ctx2.array(xh)).to_host())[:plane]))
    for overlap in (1, 0):
        ctx.set_option("overlap_halo", overlap)
        ys2 = A.matvec(ctx.array(xh)).to_host()
        res[f"spmv_overlap{overlap}_bit_identical"] = bool(np.array_equal(ys2, yp))
    ctx.set_option("overlap_halo", 1)
    # round 6: the communication stream's priority is switchable between solves (bench.py's A/B at N > 1), and the exchange's phases are
    # bracketed by HIP events on the stream each runs on (khip_profile_kernels): same bits either way, every phase seen once per product
    ctx.set_option("comm_priority", 0)
    res["spmv_priority0_bit_identical"] = bool(np.array_equal(A.matvec(ctx.array(xh)).to_host(), yp))
    ctx.set_option("comm_priority", 1)
    ctx.set_option("profile_spmv", 1); ctx.profile_kernels()
    xd = ctx.array(xh); yd = ctx.empty(m)
    for _ in range(3):
        A.matvec(xd, yd)
    prof = ctx.profile_kernels(); ctx.set_option("profile_spmv", 0)
    res["phase_launches"] = {k: prof[k][0] for k in ("halo_pack", "halo_transfer", "spmv", "spmv_boundary", "dot_allgather_combine")}
    res["phase_ms_positive"] = bool(all(prof[k][1] > 0 for k in ("halo_pack", "halo_transfer", "spmv", "spmv_boundary")))
    b = ctx.array(np.ones(m)); b2 = ctx2.array(np.ones(m))
    for fused in (0, 2):
        ws, ws2 = K.CgWorkspace(ctx, m, m), K.CgWorkspace(ctx2, m, m)
        # the periodic slab's Laplacian is singular only if every row sums to zero; the Dirichlet faces in x and y keep it SPD
        K.cg_(ws, A, b, history=True, fused=fused, itmax=60, atol=0.0, rtol=1e-10)
        K.cg_(ws2, P, b2, history=True, fused=fused, itmax=60, atol=0.0, rtol=1e-10)
        h1, h2 = ws.stats.residuals, ws2.stats.residuals
        res[f"cg_fused{fused}_niter"] = [int(ws.stats.niter), int(ws2.stats.niter)]
        res[f"cg_fused{fused}_max_rel_dev"] = float(np.max(np.abs(h1 - h2) / h2)) if len(h1) == len(h2) else None
        res[f"cg_fused{fused}_x_max_abs_dev"] = float(np.max(np.abs(ws.x.to_host() - ws2.x.to_host())))
    json.dump(res, open(out, "w"))
    print(json.dumps(res))
    os._exit(0)           # skip the communicator teardown: nothing to learn from it here


if __name__ == "__main__":
    main()
