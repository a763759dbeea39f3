"""Parity AT THE BASELINE SIZES against the CPU oracle: the HIP solvers, through the C ABI, against
histories the oracle itself produced at cfg 2 / 3 / 5 (tests/golden/oracle_cfg*.json, made by
tests/golden/make_scale_golden.py from oracle/krylov_oracle.c -- the restatement of src/cg.jl:120-291,
src/gmres.jl:121-384, src/block_gmres.jl:110-358).  Nothing here compares the GPU with itself.

Stated tolerances (fp64):
  * cfg 2, cg!: every residual norm within 1e-12 relative of the oracle's over 100 iterations (the north star's figure),
    for the reference's primitive sequence (fused = 0) and for the fused / device-resident paths; solution samples
    within 1e-12 of max|x|.
  * cfg 3, gmres!(30, restart), 45 iterations (one cycle, the restart, half a cycle): every residual norm within 1e-12
    relative (measured 3.4e-14), iteration count and status equal, solution samples within 1e-12 of max|x|.
  * cfg 5, block_gmres!(5, restart), p = 16, 7 iterations: every residual norm within 1e-12 relative (measured 3.0e-14),
    solution samples within 1e-10 of max|X| (measured 5e-12).
  Measured values are logged to gpurun_out/parity_log.jsonl (committed per round under profiles/).
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EPS = np.finfo(float).eps


def _golden(name):
    return json.load(open(os.path.join(ROOT, "tests", "golden", name)))


def _rel(h, g):
    return float(np.max(np.abs(h - g) / g))


@pytest.mark.parametrize("fused", [0, 1, 2])
def test_cg_512_matches_oracle_prefix(K, ctx, parity_log, fused):
    g = _golden("oracle_cfg2_cg512.json")
    href = np.array(g["residuals"])
    n1 = 512
    n = n1 ** 3
    assert g["n"] == n
    A = K.CsrMatrix.stencil(ctx, "poisson", n1)
    assert A.nnz == g["nnz"]
    b = ctx.empty(n)
    K.kfill_(b, 1.0)
    ws = K.CgWorkspace(ctx, n, n)
    K.cg_(ws, A, b, atol=0.0, rtol=0.0, itmax=g["niter"], history=True, fused=fused)
    st = ws.stats
    assert st.niter == g["niter"] and st.status == g["status"]
    h = st.residuals
    assert len(h) == len(href)
    dev = _rel(h, href)
    xs = ws.x.to_host()
    xg = np.array(g["x_sample"])
    xdev = float(np.max(np.abs(xs[g["x_index"]] - xg)) / np.max(np.abs(xg)))
    parity_log(test="cg_512_vs_oracle", fused=fused, iterations=st.niter, hist_max_rel=dev, x_sample_rel=xdev,
               code_info=list(A.code_info))
    assert dev <= 1e-12, dev
    assert xdev <= 1e-12, xdev


@pytest.mark.parametrize("fused", [True, False])
def test_gmres_cfg3_cycle_matches_oracle(K, ctx, parity_log, fused):
    g = _golden("oracle_cfg3_gmres256.json")
    href = np.array(g["residuals"])
    n1 = 256
    n = n1 ** 3
    A = K.CsrMatrix.stencil(ctx, "kron_unsymmetric", n1)
    assert A.nnz == g["nnz"] and n == g["n"]
    ones = ctx.empty(n)
    K.kfill_(ones, 1.0)
    b = ctx.empty(n)
    A.matvec(ones, b)                                   # b = A * ones (test/test_utils.jl:166-167); bit-identical to the oracle's
    ws = K.GmresWorkspace(ctx, n, n, memory=g["memory"])
    K.gmres_(ws, A, b, restart=True, atol=0.0, rtol=0.0, itmax=g["niter"], history=True, fused=fused)
    st = ws.stats
    assert st.niter == g["niter"] and st.status == g["status"]
    h = st.residuals
    assert len(h) == len(href)
    units = _rel(h, href) / 1e-12
    xs = ws.x.to_host()
    xg = np.array(g["x_sample"])
    xdev = float(np.max(np.abs(xs[g["x_index"]] - xg)) / np.max(np.abs(xg)))
    parity_log(test="gmres_cfg3_vs_oracle", fused=bool(fused), iterations=st.niter, hist_max_rel=_rel(h, href),
               hist_tol_units=units, x_sample_rel=xdev)
    assert units <= 1.0, units
    assert xdev <= 1e-12, xdev


def test_block_gmres_cfg5_matches_oracle(K, ctx, parity_log):
    g = _golden("oracle_cfg5_block216.json")
    href = np.array(g["residuals"])
    n1, p = 216, g["p"]
    n = n1 ** 3
    A = K.CsrMatrix.stencil(ctx, "stencil27", n1)
    assert A.nnz == g["nnz"] and n == g["n"]
    t = (np.arange(n) + 1.0) / n
    Xt = np.stack([np.cos(j * np.pi * t) + 0.1 * j for j in range(p)], axis=1)     # as make_scale_golden.cfg5_xtrue
    dXt = K.Panel.from_host(ctx, Xt)
    dB = K.Panel(ctx, n, p)
    K.spmm_(A, dXt, dB)                                 # B = A * X_true; the SpMM is bit-identical to the oracle's products
    Bh = dB.to_host()
    del dXt, dB
    ws = K.BlockGmresWorkspace(ctx, n, n, p, memory=g["memory"])
    Bd = ctx.array(np.asfortranarray(Bh).ravel(order="F"))
    K.block_gmres_(ws, A, Bd, restart=True, atol=0.0, rtol=0.0, itmax=g["niter"], history=True)
    st = ws.stats
    assert st.niter == g["niter"] and st.status == g["status"]
    h = st.residuals
    assert len(h) == len(href)
    units = _rel(h, href) / 1e-12
    X = ws.X
    xg = np.array(g["x_sample"])
    xdev = float(np.max(np.abs(X[g["x_index"], :] - xg)) / np.max(np.abs(xg)))
    parity_log(test="block_gmres_cfg5_vs_oracle", iterations=st.niter, hist_max_rel=_rel(h, href), hist_tol_units=units,
               x_sample_rel=xdev)
    assert units <= 1.0, units
    assert xdev <= 1e-10, xdev


def test_block_gmres_cfg5_banded_random_matches_oracle(K, ctx, parity_log):
    """cfg 5 BEYOND THE STENCIL: block_gmres!(memory = 5, restart = true), p = 16, on the 10.5 M-row
    "banded + random" operator of tools/bench_irregular.py (none of the stencil mechanisms applies: > 70 000 diagonals), 20
    iterations = four cycles, against the oracle's history (tests/golden/oracle_cfg5_banded_block.json, make_scale_golden.py leg
    25, 6 minutes on 8 cores).

    Tolerance, derived and not fitted: this recurrence is violently sensitive ON THE ORACLE'S SIDE -- moving every entry of B by
    ONE ULP changes the oracle's own residual norms by 1.8e-8 at iteration 2, 1e-7 .. 1e-6 up to iteration 7 and 2.6e-5 at iteration 8
    (golden field `one_ulp_sensitivity`, leg 26; binary128 is out of reach at this size).  Two correct double-precision
    implementations whose every operation rounds differently (CholeskyQR2 + FP64-MFMA panel products here, unblocked Householder
    and long-double accumulation in the oracle) cannot agree better than a modest multiple of that: iterations 0 and 1 are held to
    1e-12, iteration k in 2..8 to 200 x the running maximum of the one-ulp sensitivity up to k, all 20 to 1 % -- and iteration
    count, status and the SLOW decay (3.2e4 -> 3.2e2 in 20 iterations; max |X - X_true| still 0.8 in the oracle) must be the
    oracle's: the 400-iteration stall of round 4 on this operator is the algorithm's."""
    g = _golden("oracle_cfg5_banded_block.json")
    href = np.array(g["residuals"])
    sens = np.maximum.accumulate(np.array(g["one_ulp_sensitivity"]))
    n, p = g["n"], g["p"]
    A = K.CsrMatrix.banded_random(ctx, n, seed=1)
    assert A.nnz == g["nnz"]
    t = (np.arange(n) + 1.0) / n
    Xt = np.stack([np.cos(j * np.pi * t) + 0.1 * j for j in range(p)], axis=1)
    dXt = K.Panel.from_host(ctx, Xt)
    dB = K.Panel(ctx, n, p)
    K.spmm_(A, dXt, dB)                                 # B = A * X_true: the SpMM is bit-identical to the oracle's products
    del dXt
    ws = K.BlockGmresWorkspace(ctx, n, n, p, memory=g["memory"])
    K.block_gmres_(ws, A, dB if ws.adopted else ctx.array(np.asfortranarray(dB.to_host()).ravel(order="F")),
                   restart=True, atol=0.0, rtol=0.0, itmax=g["niter"], history=True)
    st = ws.stats
    assert st.niter == g["niter"] and st.status == g["status"]
    h = st.residuals
    assert len(h) == len(href)
    dev = np.abs(h - href) / href
    X = ws.X
    xg = np.array(g["x_sample"])
    xdev = float(np.max(np.abs(X[g["x_index"], :] - xg)) / np.max(np.abs(xg)))
    parity_log(test="block_gmres_cfg5_banded_random_vs_oracle", iterations=st.niter, hist_rel_per_iteration=[float(v) for v in dev],
               oracle_one_ulp_sensitivity=[float(v) for v in g["one_ulp_sensitivity"]], hist_max_rel=float(dev.max()), x_sample_rel=xdev,
               residual_first=float(h[0]), residual_last=float(h[-1]), oracle_residual_last=float(href[-1]))
    assert dev[0] <= 1e-12 and dev[1] <= 1e-12, dev[:2]
    for k in range(2, len(sens)):
        assert dev[k] <= 200.0 * sens[k] + 1e-12, (k, dev[k], sens[k])
    assert dev.max() <= 1e-2, dev.max()
    assert h[-1] > 5e-3 * h[0]                 # ... and it is as slow here as in the oracle


@pytest.mark.parametrize("fused", [2, 0])
def test_bicgstab_256_matches_oracle_within_the_derived_tolerance(K, ctx, parity_log, fused):
    """bicgstab! (the fourth north-star solver; no BASELINE config of its own) on cfg 3's operator at full size, 25 iterations.
    BiCGSTAB's alpha and omega are ratios of cancelling dots: after 25 iterations at 256^3 the CPU oracle ITSELF is
    d = 1.5e-6 away from the binary128 history of the recurrence (tests/golden/oracle_bicgstab256.json, leg 4 of
    make_scale_golden.py).  Tolerances derived from d: HIP path within 8 d of the exact history, hence within
    9 d of the oracle; iteration count and status equal."""
    g = _golden("oracle_bicgstab256.json")
    href, hq, d = np.array(g["residuals"]), np.array(g["quad_residuals"]), float(g["oracle_double_max_rel_dev"])
    n1 = 256
    n = n1 ** 3
    A = K.CsrMatrix.stencil(ctx, "kron_unsymmetric", n1)
    assert A.nnz == g["nnz"]
    ones = ctx.empty(n)
    K.kfill_(ones, 1.0)
    b = ctx.empty(n)
    A.matvec(ones, b)
    ws = K.BicgstabWorkspace(ctx, n, n)
    K.bicgstab_(ws, A, b, atol=0.0, rtol=0.0, itmax=g["niter"], history=True, fused=fused)
    st = ws.stats
    assert st.niter == g["niter"] and st.status == g["status"] and len(st.residuals) == len(href)
    d_gpu = _rel(st.residuals, hq)
    parity_log(test="bicgstab_256_vs_oracle", fused=fused, iterations=st.niter, gpu_vs_quad=d_gpu, cpu_oracle_vs_quad=d,
               gpu_vs_oracle=_rel(st.residuals, href))
    assert d_gpu <= 8 * d and _rel(st.residuals, href) <= 9 * d


# ---- full solves TO CONVERGENCE at the BASELINE sizes ----------------------------------------------
# The benchmark definition of the reference, cg(A, b, atol = 0, rtol = 1e-8, itmax = n) (benchmark/benchmarks.jl:14-21), on
# cfg 2 / 3 / 5; goldens: tests/golden/oracle_cfg*_full.json (make_scale_golden.py legs 12 / 13 / 15, the oracle's own full
# solves: 1225 / 940 / 35 iterations).  Asserted: the SAME iteration count, status and solved flag, and every residual norm of
# the whole history within FULL_TOL of the oracle's.  FULL_TOL is DERIVED from
# tests/golden/full_solve_tolerance.json: the oracle's own double-precision history against the binary128 build of its source,
# same operators and settings, on a ladder of sizes where binary128 is affordable.  cg!: 1.3e-15 ... 3e-14 over 39 ... 159
# iterations -> 1e-8 leaves four orders for the 1225 iterations at 512^3 (measured against the oracle: 7.7e-11).
# gmres!(30, restart): 7.7e-10 / 1.6e-9 / 7.4e-9 after 97 / 180 / 209 iterations -- the restarts feed the Givens estimate of
# the residual back into the basis (src/gmres.jl:229-231) and the drift grows about tenfold per 110 iterations; at 940
# iterations two correct double implementations are therefore expected up to ~1e-2 apart (measured: 4.0e-4 in rounds 3 and 4).
# The derivation bounds what CAN be asserted (anything below a quarter of the 2 % margin by which the oracle's last iterates
# clear the stopping threshold implies equal iteration counts); the bound itself is set to 3 x the measured gap, 1.2e-3, so
# that a regression of one order cannot hide under it.  block_gmres!: 2e-6 / 6e-5 / 4e-7 at 10^3 / 14^3 / 18^3 x 16 over 20-27 iterations (the restart re-orthogonalises a residual block that
# loses conditioning as columns converge); at 216^3 the two implementations stay 1e-11 apart over 35 iterations, bound 1e-6.
FULL_TOL = {"cg": 1e-8, "gmres": 1.2e-3, "block_gmres": 1e-6}


def _full_check(g, st, parity_log, name, extra):
    href = np.array(g["residuals"])
    h = np.asarray(st.residuals)
    m = min(len(h), len(href))
    devs = np.abs(h[:m] - href[:m]) / href[:m]
    eps = g["rtol"] * href[0]
    margin = float(np.min(np.abs(href[-3:] / eps - 1.0)))          # how clearly the oracle's last iterates decide the stop
    parity_log(test=name, iterations=st.niter, ref_iterations=g["niter"], status=st.status, hist_max_rel=float(devs.max()),
               hist_max_rel_first100=float(devs[:101].max()), oracle_stop_margin=margin, **extra)
    assert st.niter == g["niter"] and st.status == g["status"] and bool(st.solved) == bool(g["solved"]), (st.niter, g["niter"], st.status)
    return float(devs.max())


@pytest.mark.parametrize("fused", [2, 0])
def test_cg_512_full_solve_equal_iteration_count(K, ctx, parity_log, fused):
    g = _golden("oracle_cfg2_cg512_full.json")
    n = 512 ** 3
    A = K.CsrMatrix.stencil(ctx, "poisson", 512)
    b = ctx.empty(n)
    K.kfill_(b, 1.0)
    ws = K.CgWorkspace(ctx, n, n)
    K.cg_(ws, A, b, atol=g["atol"], rtol=g["rtol"], itmax=n, history=True, fused=fused)
    dev = _full_check(g, ws.stats, parity_log, "cg_512_full_vs_oracle", dict(fused=fused))
    xs = ws.x.to_host()
    xg = np.array(g["x_sample"])
    xdev = float(np.max(np.abs(xs[g["x_index"]] - xg)) / np.max(np.abs(xg)))
    assert dev <= FULL_TOL["cg"] and xdev <= 1e-9, (dev, xdev)


def test_gmres_cfg3_full_solve_equal_iteration_count(K, ctx, parity_log):
    g = _golden("oracle_cfg3_gmres256_full.json")
    n = 256 ** 3
    A = K.CsrMatrix.stencil(ctx, "kron_unsymmetric", 256)
    ones = ctx.empty(n)
    K.kfill_(ones, 1.0)
    b = ctx.empty(n)
    A.matvec(ones, b)
    ws = K.GmresWorkspace(ctx, n, n, memory=g["memory"])
    K.gmres_(ws, A, b, restart=True, atol=g["atol"], rtol=g["rtol"], itmax=n, history=True)
    dev = _full_check(g, ws.stats, parity_log, "gmres_cfg3_full_vs_oracle", {})
    xs = ws.x.to_host()
    xg = np.array(g["x_sample"])
    xdev = float(np.max(np.abs(xs[g["x_index"]] - xg)) / np.max(np.abs(xg)))
    assert dev <= FULL_TOL["gmres"] and xdev <= 1e-8, (dev, xdev)


def test_block_gmres_cfg5_full_solve_equal_iteration_count(K, ctx, parity_log):
    g = _golden("oracle_cfg5_block216_full.json")
    n1, p = 216, g["p"]
    n = n1 ** 3
    A = K.CsrMatrix.stencil(ctx, "stencil27", n1)
    t = (np.arange(n) + 1.0) / n
    Xt = np.stack([np.cos(j * np.pi * t) + 0.1 * j for j in range(p)], axis=1)
    dXt = K.Panel.from_host(ctx, Xt)
    dB = K.Panel(ctx, n, p)
    K.spmm_(A, dXt, dB)
    Bh = dB.to_host()
    del dXt, dB
    ws = K.BlockGmresWorkspace(ctx, n, n, p, memory=g["memory"])
    Bd = ctx.array(np.asfortranarray(Bh).ravel(order="F"))
    K.block_gmres_(ws, A, Bd, restart=True, atol=g["atol"], rtol=g["rtol"], itmax=n, history=True)
    dev = _full_check(g, ws.stats, parity_log, "block_gmres_cfg5_full_vs_oracle", {})
    X = ws.X
    xg = np.array(g["x_sample"])
    xdev = float(np.max(np.abs(X[g["x_index"], :] - xg)) / np.max(np.abs(xg)))
    assert dev <= FULL_TOL["block_gmres"] and xdev <= 1e-8, (dev, xdev)


# ---- BASELINE cfg 4: cg! on get_div_grad(1024^3) row-partitioned over 8 ranks ------------------------
# Golden: tests/golden/oracle_cfg4_cg1024.json = the oracle's cg! (ko_cg, unchanged) on the MATRIX-FREE 7-point operator
# (oracle/krylov_oracle.c ko_stencil7_matvec, pinned bit for bit to the CSR operator by tests/test_oracle.py), 100 iterations,
# atol = rtol = 0.  The HIP side is the distributed path at full size on ONE GPU: 8 in-process ranks (khip_comm_init_local),
# each with its 128-plane slab of the CSR operator (global columns up to 2^30, 8 MiB halo planes), 155 GB of HBM in total --
# the same code the 8-GPU run executes except for the transport under the collectives (tests/test_gpu_rccl_multi.py covers RCCL).
#
# Tolerance.  At n = 2^30 the DOCUMENTED oracle dot -- a sequential sum in x87 extended precision, 64-bit mantissa -- carries a
# rounding of its own of ~sqrt(n) 2^-64 = 2e-15 per dot, which cg! amplifies: the golden holds a second history of the same
# ko_cg with every dot computed by Dot2 (double-double accumulation, error ~1e-31; ko_set_dot_mode(1), make_scale_golden.py
# leg 41), and the documented oracle is d = oracle_vs_exact_dots_max_rel_dev (5e-12 over the 100 iterations) away from it.
# binary128 (the yardstick at the smaller sizes) is out of reach here.  Asserted: the HIP path within the north
# star's 1e-12 of the exact-dot oracle history, and within d + 1e-12 of the documented one (512^3: 4e-13 against the documented
# oracle, whose own rounding is three times smaller there).
def _cfg4_ranks(K, n1, world, iters, x_index, halo_mode=0):
    import threading
    n = n1 ** 3
    starts = K.row_partition(n, world)
    out, errs = [None] * world, []

    def run(rank):
        try:
            c = K.Context(0)
            c.comm_init_local(rank, world, 4141 + halo_mode)
            c.set_option("halo_mode", halo_mode)
            r0, r1 = starts[rank], starts[rank + 1]
            A = K.CsrMatrix.stencil(c, "poisson", n1, rows=(r0, r1), distributed=True)
            m = r1 - r0
            b = c.empty(m)
            K.kfill_(b, 1.0)
            ws = K.CgWorkspace(c, m, m)
            K.cg_(ws, A, b, atol=0.0, rtol=0.0, itmax=iters, history=True, fused=2)
            mine = [(k, i - r0) for k, i in enumerate(x_index) if r0 <= i < r1]
            xs = {}
            if mine:
                xh = ws.x.to_host()
                xs = {k: float(xh[j]) for k, j in mine}
                del xh
            out[rank] = dict(nnz=A.nnz, niter=ws.stats.niter, status=ws.stats.status, hist=ws.stats.residuals.copy(), xs=xs,
                             halo=list(A.halo_info), code=list(A.code_info))
            c.barrier()
            del ws, A, b
            c.close()
        except Exception as e:      # noqa: BLE001
            import traceback
            errs.append((rank, repr(e), traceback.format_exc()))

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    return out


@pytest.mark.parametrize("halo_mode", [0, 2])          # per-operator choice (neighbour planes) / all-gather of x (the north star's recipe)
def test_cg_1024_over_8_ranks_matches_oracle_prefix(K, ctx, parity_log, halo_mode):
    import gc
    gc.collect()
    path = os.path.join(ROOT, "tests", "golden", "oracle_cfg4_cg1024.json")
    g = json.load(open(path))
    n1, world = 1024, 8
    n = n1 ** 3
    assert g["n"] == n and g["nnz"] == 7 * n - 6 * n1 * n1
    free_b, _total = ctx.mem_info()
    need = 175e9 if halo_mode == 0 else 250e9      # CSR 94 GB + codes 7.5 GB + 5 vectors 43 GB (+ the gathered x per rank: 8 x 8.6 GB)
    if free_b < need:
        pytest.skip(f"cfg 4 on one GPU needs {need / 1e9:.0f} GB of free HBM, {free_b / 1e9:.0f} GB are free")
    out = _cfg4_ranks(K, n1, world, g["niter"], g["x_index"], halo_mode)
    href = np.array(g["residuals"])
    o = out[0]
    assert sum(x["nnz"] for x in out) == g["nnz"]
    assert all(np.array_equal(x["hist"], o["hist"]) for x in out), "ranks disagree on the history"
    assert o["niter"] == g["niter"] and o["status"] == g["status"]
    assert len(o["hist"]) == len(href)
    hex_ = np.array(g["residuals_exact_dots"])
    d_oracle = float(g["oracle_vs_exact_dots_max_rel_dev"])
    dev = _rel(o["hist"], href)
    dev_exact = _rel(o["hist"], hex_)
    xs = {}
    for x in out:
        xs.update(x["xs"])
    xv = np.array([xs[k] for k in range(len(g["x_index"]))])
    xg, xe = np.array(g["x_sample"]), np.array(g["x_sample_exact_dots"])
    xdev = float(np.max(np.abs(xv - xg)) / np.max(np.abs(xg)))
    xdev_exact = float(np.max(np.abs(xv - xe)) / np.max(np.abs(xe)))
    parity_log(test="cg_1024_8ranks_vs_oracle", halo_mode=halo_mode, iterations=o["niter"], hist_max_rel=dev, hist_max_rel_vs_exact_dot_oracle=dev_exact,
               documented_oracle_vs_exact_dot_oracle=d_oracle, x_sample_rel=xdev, x_sample_rel_vs_exact_dot_oracle=xdev_exact,
               halo_info_rank0=o["halo"], code_info_rank0=o["code"])
    assert dev_exact <= 1e-12, dev_exact
    assert dev <= d_oracle + 1e-12, (dev, d_oracle)
    assert xdev_exact <= 1e-12 and xdev <= d_oracle + 1e-12, (xdev_exact, xdev)


# ---- cfg 2 against the oracle with EXACT dots (round 4) -------------------------------------------------------------------
# tests/golden/oracle_cfg2_cg512_exact_dots.json (make_scale_golden.py leg 22): the same ko_cg with every dot computed by Dot2
# (ko_set_dot_mode(1)) instead of the documented sequential extended-precision sum.  SpMV and the fma axpys of the HIP path
# are bit-identical to the oracle's, its dots are Dot2 as well (device_reduce.hpp) -- so against THIS history only the order
# of the double-double partial sums differs, which changes a correctly rounded result next to never.  Asserted: the
# 100-iteration prefix AND the full solve to rtol 1e-8 (1225 iterations) within EXACT_TOL of it, same iteration count and
# status -- the gap to the documented oracle (4e-13 / 7.7e-11) is that oracle's own summation error.
EXACT_TOL = 1e-13
# gmres! (45-iteration prefix, full solve of 940 iterations = 31 restarts) and bicgstab! (25 iterations) against the same kind of
# history: measured 0.0 -- bit-identical -- in round 4 for all of them (profiles/r04_parity_log.jsonl), so the north star's 1e-12
# is asserted even for the restarted full solve, whose distance to the DOCUMENTED oracle (4.0e-4, bound 1.2e-3 above) turns out to
# be that oracle's extended-precision summation error amplified by the restarts, not a property of the HIP path.
GMRES_EXACT_TOL = (1e-12, 1e-12)
BICGSTAB_EXACT_TOL = 1e-12


@pytest.mark.parametrize("fused", [2, 0])
def test_cg_512_against_the_exact_dot_oracle(K, ctx, parity_log, fused):
    g = _golden("oracle_cfg2_cg512_exact_dots.json")
    n = 512 ** 3
    A = K.CsrMatrix.stencil(ctx, "poisson", 512)
    b = ctx.empty(n)
    K.kfill_(b, 1.0)
    ws = K.CgWorkspace(ctx, n, n)
    K.cg_(ws, A, b, atol=0.0, rtol=0.0, itmax=100, history=True, fused=fused)
    hp = np.array(g["prefix_residuals"])
    assert len(ws.stats.residuals) == len(hp)
    dev_prefix = _rel(ws.stats.residuals, hp)
    xs = ws.x.to_host()
    xp = np.array(g["prefix_x_sample"])
    xdev_prefix = float(np.max(np.abs(xs[g["x_index"]] - xp)) / np.max(np.abs(xp)))
    K.cg_(ws, A, b, atol=0.0, rtol=1e-8, itmax=n, history=True, fused=fused)
    st = ws.stats
    hf = np.array(g["residuals"])
    m = min(len(hf), len(st.residuals))
    dev_full = _rel(np.asarray(st.residuals)[:m], hf[:m])
    xs = ws.x.to_host()
    xf = np.array(g["x_sample"])
    xdev_full = float(np.max(np.abs(xs[g["x_index"]] - xf)) / np.max(np.abs(xf)))
    parity_log(test="cg_512_vs_exact_dot_oracle", fused=fused, prefix_hist_max_rel=dev_prefix, prefix_x_sample_rel=xdev_prefix,
               full_iterations=st.niter, ref_iterations=g["niter"], full_hist_max_rel=dev_full, full_x_sample_rel=xdev_full,
               bit_identical_history=bool(len(hf) == len(st.residuals) and np.array_equal(np.asarray(st.residuals), hf)))
    assert st.niter == g["niter"] and st.status == g["status"]
    assert dev_prefix <= EXACT_TOL and xdev_prefix <= EXACT_TOL, (dev_prefix, xdev_prefix)
    assert dev_full <= EXACT_TOL and xdev_full <= EXACT_TOL, (dev_full, xdev_full)
    # measured in round 4: 0.0 everywhere -- all 1226 residual norms of the full solve and all 101 of the prefix bit for bit,
    # fused = 2 and the reference's primitive sequence alike (profiles/r04_parity_log.jsonl); the bound above leaves room for
    # the one-ulp difference two faithful dot algorithms may show on some input


def test_gmres_cfg3_against_the_exact_dot_oracle(K, ctx, parity_log):
    """cfg 3 against ko_gmres with Dot2 dots (tests/golden/oracle_cfg3_gmres256_exact_dots.json, leg 23): the 45-iteration prefix
    and the full solve to rtol 1e-8.  The Gram-Schmidt coefficients and the norms are the only reductions of gmres!; with exact
    dots on both sides what separates the two implementations is the order of the double-double partial sums alone."""
    g = _golden("oracle_cfg3_gmres256_exact_dots.json")
    n = 256 ** 3
    A = K.CsrMatrix.stencil(ctx, "kron_unsymmetric", 256)
    ones = ctx.empty(n)
    K.kfill_(ones, 1.0)
    b = ctx.empty(n)
    A.matvec(ones, b)
    ws = K.GmresWorkspace(ctx, n, n, memory=g["memory"])
    K.gmres_(ws, A, b, restart=True, atol=0.0, rtol=0.0, itmax=45, history=True)
    hp = np.array(g["prefix_residuals"])
    assert len(ws.stats.residuals) == len(hp)
    dev_prefix = _rel(ws.stats.residuals, hp)
    K.gmres_(ws, A, b, restart=True, atol=0.0, rtol=1e-8, itmax=n, history=True)
    st = ws.stats
    hf = np.array(g["residuals"])
    m = min(len(hf), len(st.residuals))
    dev_full = _rel(np.asarray(st.residuals)[:m], hf[:m])
    xs = ws.x.to_host()
    xf = np.array(g["x_sample"])
    xdev_full = float(np.max(np.abs(xs[g["x_index"]] - xf)) / np.max(np.abs(xf)))
    parity_log(test="gmres_cfg3_vs_exact_dot_oracle", prefix_hist_max_rel=dev_prefix, full_iterations=st.niter, ref_iterations=g["niter"],
               full_hist_max_rel=dev_full, full_x_sample_rel=xdev_full,
               bit_identical_history=bool(len(hf) == len(st.residuals) and np.array_equal(np.asarray(st.residuals), hf)))
    assert st.niter == g["niter"] and st.status == g["status"]
    assert dev_prefix <= GMRES_EXACT_TOL[0] and dev_full <= GMRES_EXACT_TOL[1], (dev_prefix, dev_full)


@pytest.mark.parametrize("fused", [2, 0])
def test_bicgstab_256_against_the_exact_dot_oracle(K, ctx, parity_log, fused):
    """bicgstab! on cfg 3's operator, 25 iterations, against ko_bicgstab with Dot2 dots (leg 24)."""
    g = _golden("oracle_bicgstab256_exact_dots.json")
    n = 256 ** 3
    A = K.CsrMatrix.stencil(ctx, "kron_unsymmetric", 256)
    ones = ctx.empty(n)
    K.kfill_(ones, 1.0)
    b = ctx.empty(n)
    A.matvec(ones, b)
    ws = K.BicgstabWorkspace(ctx, n, n)
    K.bicgstab_(ws, A, b, atol=0.0, rtol=0.0, itmax=g["niter"], history=True, fused=fused)
    st = ws.stats
    href = np.array(g["residuals"])
    assert st.niter == g["niter"] and st.status == g["status"] and len(st.residuals) == len(href)
    dev = _rel(st.residuals, href)
    parity_log(test="bicgstab_256_vs_exact_dot_oracle", fused=fused, iterations=st.niter, hist_max_rel=dev,
               bit_identical_history=bool(np.array_equal(np.asarray(st.residuals), href)))
    assert dev <= BICGSTAB_EXACT_TOL, dev


# ---- the row-partitioned path at the BASELINE sizes, all solvers (round 4) ------------------------------------------------
# gmres!(30) and bicgstab! on cfg 3's operator over 4 / 8 in-process ranks, cg! at 512^3 over 8 (the strong-scaling layout of
# bench.py --gpus 8), each against the exact-dot oracle history of the GLOBAL system: the per-rank double-double partials are
# all-gathered as (hi, lo) pairs and merged in rank order, so the partition must not cost a bit either.
def _ranks(K, world, hub, body):
    import threading
    out, errs = [None] * world, []

    def run(rank):
        try:
            c = K.Context(0)
            c.comm_init_local(rank, world, hub)
            out[rank] = body(c, rank)
            c.barrier()
            c.close()
        except Exception as e:      # noqa: BLE001
            import traceback
            errs.append((rank, repr(e), traceback.format_exc()))
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    return out


def test_partitioned_solvers_at_full_size_against_the_exact_dot_oracle(K, ctx, parity_log):
    import gc
    gc.collect()
    gg, gb, gc2 = _golden("oracle_cfg3_gmres256_exact_dots.json"), _golden("oracle_bicgstab256_exact_dots.json"), _golden("oracle_cfg2_cg512_exact_dots.json")
    n3 = 256 ** 3

    def unsym(world):
        starts = K.row_partition(n3, world)

        def body(c, rank):
            r0, r1 = starts[rank], starts[rank + 1]
            A = K.CsrMatrix.stencil(c, "kron_unsymmetric", 256, rows=(r0, r1), distributed=True)
            m = r1 - r0
            ones = c.empty(m)
            K.kfill_(ones, 1.0)
            b = c.empty(m)
            A.matvec(ones, b)
            ws = K.GmresWorkspace(c, m, m, memory=30)
            K.gmres_(ws, A, b, restart=True, atol=0.0, rtol=0.0, itmax=45, history=True)
            hg = ws.stats.residuals.copy()
            del ws
            wb = K.BicgstabWorkspace(c, m, m)
            K.bicgstab_(wb, A, b, atol=0.0, rtol=0.0, itmax=25, history=True, fused=2)
            return hg, wb.stats.residuals.copy()
        return body

    for world in (4, 8):
        res = _ranks(K, world, 5150 + world, unsym(world))
        hg, hb = res[0]
        assert all(np.array_equal(r[0], hg) and np.array_equal(r[1], hb) for r in res), "ranks disagree"
        dg, db = _rel(hg, np.array(gg["prefix_residuals"])), _rel(hb, np.array(gb["residuals"]))
        parity_log(test="partitioned_cfg3_vs_exact_dot_oracle", ranks=world, gmres_hist_max_rel=dg, bicgstab_hist_max_rel=db)
        assert dg <= 1e-12 and db <= 1e-12, (world, dg, db)

    n2 = 512 ** 3
    starts = K.row_partition(n2, 8)

    def cg_body(c, rank):
        r0, r1 = starts[rank], starts[rank + 1]
        A = K.CsrMatrix.stencil(c, "poisson", 512, rows=(r0, r1), distributed=True)
        m = r1 - r0
        b = c.empty(m)
        K.kfill_(b, 1.0)
        ws = K.CgWorkspace(c, m, m)
        K.cg_(ws, A, b, atol=0.0, rtol=1e-8, itmax=n2, history=True, fused=2)
        return ws.stats.niter, ws.stats.status, ws.stats.residuals.copy()
    res = _ranks(K, 8, 5199, cg_body)
    niter, status, h = res[0]
    assert all(r[0] == niter and np.array_equal(r[2], h) for r in res)
    hf = np.array(gc2["residuals"])
    assert niter == gc2["niter"] and status == gc2["status"] and len(h) == len(hf)
    dc = _rel(h, hf)
    parity_log(test="partitioned_cfg2_full_solve_vs_exact_dot_oracle", ranks=8, iterations=niter, hist_max_rel=dc,
               bit_identical_history=bool(np.array_equal(h, hf)))
    assert dc <= 1e-12, dc


# ---- the vector solvers on the NON-STENCIL operators at full size (round 5) ------------------------------------------------
# tests/golden/oracle_irregular_cg.json / oracle_irregular_gmres_bicgstab.json (make_scale_golden.py legs 27 / 28): cg! on the 10.5 M-row
# banded + random operator, gmres!(30, restart) and bicgstab! on its nonsymmetric variant with four rows of 3000 further entries --
# products through the LDS stream kernel and the strided vector kernel, not the staged stencil kernels -- against the oracle with the
# documented dots and with exact (Dot2) dots.  Asserted as for cfg 4: <= 1e-12 against the exact-dot history (the HIP dots are Dot2 too;
# two faithful dots may differ by an ulp), and <= d + 1e-12 against the documented history, d = the documented oracle's own distance to
# its exact-dot variant, computed from the two histories of the golden.

def _irregular_rhs(K, ctx, A, n):
    xt = ctx.array(np.cos(np.arange(n) * 1e-3) + 0.5)
    b = ctx.zeros(n)
    A.matvec(xt, b)                                # bit-identical to the oracle's serial product
    return b


def _own_distance(a, b):
    a, b = np.array(a), np.array(b)
    k = min(len(a), len(b))
    return float(np.max(np.abs(a[:k] - b[:k]) / b[:k]))


@pytest.mark.parametrize("fused", [2, 0])
def test_irregular_cg_full_size_against_the_oracles(K, ctx, parity_log, fused):
    path = os.path.join(ROOT, "tests", "golden", "oracle_irregular_cg.json")
    if not os.path.exists(path):
        pytest.skip("golden not generated (make_scale_golden.py 27)")
    g = json.load(open(path))
    n = g["n"]
    A = K.CsrMatrix.banded_random(ctx, n, seed=1)
    assert A.nnz == g["nnz"]
    b = _irregular_rhs(K, ctx, A, n)
    ws = K.CgWorkspace(ctx, n, n)
    K.cg_(ws, A, b, atol=0.0, rtol=0.0, itmax=100, history=True, fused=fused)
    h = ws.stats.residuals
    d = _own_distance(g["prefix_residuals_exact"], g["prefix_residuals"])       # same normalisation as _rel(h, documented)
    dev, dev_exact = _rel(h, np.array(g["prefix_residuals"])), _rel(h, np.array(g["prefix_residuals_exact"]))
    xs = ws.x.to_host()[g["x_index"]]
    xdev_exact = float(np.max(np.abs(xs - np.array(g["prefix_x_sample_exact"]))) / np.max(np.abs(g["prefix_x_sample_exact"])))
    K.cg_(ws, A, b, atol=0.0, rtol=1e-8, itmax=n, history=True, fused=fused)
    st = ws.stats
    hf, he = st.residuals, np.array(g["residuals_exact"])
    full_exact = _rel(hf, he) if len(hf) == len(he) else None
    parity_log(test="irregular_cg_full_size", fused=fused, kernel=A.spmv_kernel_choice, prefix_vs_documented=dev, prefix_vs_exact=dev_exact,
               documented_vs_exact=d, x_sample_vs_exact=xdev_exact, niter=st.niter, oracle_niter=g["niter"], oracle_niter_exact=g["niter_exact"],
               full_vs_exact=full_exact)
    assert dev_exact <= 1e-12, dev_exact
    assert dev <= d + 1e-12, (dev, d)
    assert xdev_exact <= 1e-12, xdev_exact
    assert st.niter == g["niter_exact"] and st.status == g["status_exact"] and st.solved
    assert full_exact is not None and full_exact <= 1e-12, full_exact
    assert abs(st.niter - g["niter"]) <= 2           # the documented oracle's last iterates sit within a fraction of a percent of the threshold


def test_irregular_gmres_bicgstab_full_size_against_the_oracles(K, ctx, parity_log):
    path = os.path.join(ROOT, "tests", "golden", "oracle_irregular_gmres_bicgstab.json")
    if not os.path.exists(path):
        pytest.skip("golden not generated (make_scale_golden.py 28)")
    g = json.load(open(path))
    n = g["n"]
    A = K.CsrMatrix.banded_random(ctx, n, seed=1, unsym=True, dense_rows=4)
    assert A.nnz == g["nnz"]
    b = _irregular_rhs(K, ctx, A, n)
    ws = K.GmresWorkspace(ctx, n, n, memory=30)
    K.gmres_(ws, A, b, restart=True, atol=0.0, rtol=0.0, itmax=45, history=True)
    hg = ws.stats.residuals
    dg = _own_distance(g["gmres_residuals_exact"], g["gmres_residuals"])
    g_dev, g_exact = _rel(hg, np.array(g["gmres_residuals"])), _rel(hg, np.array(g["gmres_residuals_exact"]))
    xg = ws.x.to_host()[g["x_index"]]
    gx_exact = float(np.max(np.abs(xg - np.array(g["gmres_x_sample_exact"]))) / np.max(np.abs(g["gmres_x_sample_exact"])))
    del ws
    wb = K.BicgstabWorkspace(ctx, n, n)
    K.bicgstab_(wb, A, b, atol=0.0, rtol=0.0, itmax=25, history=True)
    hb = wb.stats.residuals
    db = _own_distance(g["bicgstab_residuals_exact"], g["bicgstab_residuals"])
    b_dev, b_exact = _rel(hb, np.array(g["bicgstab_residuals"])), _rel(hb, np.array(g["bicgstab_residuals_exact"]))
    parity_log(test="irregular_gmres_bicgstab_full_size", kernel=A.spmv_kernel_choice, gmres_vs_documented=g_dev, gmres_vs_exact=g_exact,
               gmres_documented_vs_exact=dg, gmres_x_vs_exact=gx_exact, bicgstab_vs_documented=b_dev, bicgstab_vs_exact=b_exact,
               bicgstab_documented_vs_exact=db)
    assert g_exact <= 1e-12 and gx_exact <= 1e-12, (g_exact, gx_exact)
    assert g_dev <= dg + 1e-12, (g_dev, dg)
    assert wb.stats.niter == g["bicgstab_niter_exact"]
    assert b_exact <= 1e-12, b_exact
    assert b_dev <= db + 1e-12, (b_dev, db)
