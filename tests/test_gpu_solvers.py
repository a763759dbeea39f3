"""GPU parity of cg_ / gmres_ / bicgstab_ against the CPU oracle (same seeded inputs, through the C ABI).

Tolerances (fp64, stated per the north star: iteration counts equal, residual norms within a stated
relative tolerance):
  * iteration counts and status strings: EQUAL to the oracle's.
  * residual-norm histories: for every k, |r_k(gpu) - r_k(cpu)| <= HIST_RTOL * r_k(cpu) + HIST_FLOOR * r_0
    with HIST_RTOL = 1e-10 and HIST_FLOOR = 100 eps.  The floor is the rounding of forming b - A x at a
    GMRES restart (cancellation amplifies one-ulp differences in x by ||b|| / ||r_k||); the measured
    deviations (CG: <= 1.3e-13 relative over 157 iterations at 64^3) are logged to
    gpurun_out/parity_log.jsonl and quoted in DESIGN.md.
  * final true residual ||b - A x|| / ||b||: both below the solver tolerance and within 1e-12 of
    each other in absolute terms.
"""
import json
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EPS = np.finfo(float).eps
HIST_RTOL = 1e-10
HIST_FLOOR = 100 * EPS


def _bicgstab_rtol(n1):
    """Tolerance of the GPU-vs-oracle BiCGSTAB history, DERIVED, not fitted: both are double-precision runs of one
    recurrence whose exact (binary128) history is in tests/golden/quad_histories.json together with the CPU oracle's own
    distance d to it (1.1e-11 at 8^3, 1.5e-8 at 16^3: alpha and omega are ratios of cancelling dots).  The HIP path is held
    to 8 d of the exact history (tests/test_gpu_quad_reference.py), so the two can differ by at most (1 + 8) d."""
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "quad_histories.json")))
    d = {c["n1"]: c["oracle_double_max_rel_dev"] for c in g["cases"] if c["solver"] == "bicgstab"}
    return 9.0 * d[n1] + 1e-13


def _upload(K, ctx, A):
    return K.CsrMatrix.from_host(ctx, A.rowptr, A.col, A.val, (A.n, A.n))


def _hist_dev(h_gpu, h_cpu):
    """max_k |dr_k| / (HIST_RTOL r_k + HIST_FLOOR r_0): <= 1 means within the stated tolerance."""
    assert len(h_gpu) == len(h_cpu), (len(h_gpu), len(h_cpu))
    if not len(h_cpu):
        return 0.0
    return float(np.max(np.abs(h_gpu - h_cpu) / (HIST_RTOL * h_cpu + HIST_FLOOR * h_cpu[0])))


def _hist_rel(h_gpu, h_cpu):
    return float(np.max(np.abs(h_gpu - h_cpu) / np.maximum(h_cpu, 1e-300))) if len(h_cpu) else 0.0


# ------------------------------------------------------------------------------ CG

@pytest.mark.parametrize("n1", [16, 32, 64])
@pytest.mark.parametrize("fused", [False, True, 2])
def test_cg_poisson_matches_oracle(K, ctx, oracle, parity_log, n1, fused):
    """cfg 1 (64^3) and the reference's own sparse_laplacian(16) case (test/test_cg.jl:22-28)."""
    A = oracle.poisson3d(n1)
    b = np.ones(A.n)
    ref = oracle.cg(A, b, history=True)
    dA = _upload(K, ctx, A)
    x, st, ws = K.cg(dA, ctx.array(b), history=True, fused=fused)
    assert st.solved and st.status == ref.status == "solution good enough given atol and rtol"
    assert st.niter == ref.niter
    dev = _hist_dev(st.residuals, ref.residuals)
    xh = x.to_host()
    S = A.to_scipy()
    res_gpu = np.linalg.norm(b - S @ xh) / np.linalg.norm(b)
    res_cpu = np.linalg.norm(b - S @ ref.x) / np.linalg.norm(b)
    parity_log(test="cg_poisson", n1=n1, fused=fused, niter=st.niter, hist_tol_units=dev,
               hist_max_rel=_hist_rel(st.residuals, ref.residuals),
               x_max_abs=float(np.max(np.abs(xh - ref.x))), res_gpu=res_gpu, res_cpu=res_cpu)
    assert dev <= 1.0
    assert res_gpu <= 1e-6 and abs(res_gpu - res_cpu) <= 1e-12          # test/test_cg.jl:22-28 bound
    assert np.allclose(xh, ref.x, rtol=0, atol=1e-9 * np.abs(ref.x).max())
    assert ws.nbytes == 4 * 8 * A.n                                      # storage: CG = 4n (test_allocations.jl:41-57)


def test_golden_histories_all_solvers(K, ctx, oracle, parity_log):
    """The committed golden vectors (tests/golden/oracle_histories.json): CG with the package defaults and with the
    benchmark settings of benchmark/benchmarks.jl:14-21 (atol=0, rtol=1e-8, itmax=n), restarted GMRES with and
    without reorthogonalisation, BiCGSTAB."""
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_histories.json")))
    run = {"cg": K.cg, "gmres": K.gmres, "bicgstab": K.bicgstab}
    for case in gold["cases"]:
        A = getattr(oracle, case["matrix"])(case["n1"])
        b = np.ones(A.n) if case["rhs"] == "ones" else A.matvec(np.ones(A.n))
        x, st, _ = run[case["solver"]](_upload(K, ctx, A), ctx.array(b), history=True, **case["kwargs"])
        assert st.niter == case["niter"] and st.status == case["status"], case["name"]
        ref_hist = np.array(case["residuals"])
        dev = _hist_dev(st.residuals, ref_hist)
        rel = _hist_rel(st.residuals, ref_hist)
        parity_log(test="golden", name=case["name"], hist_tol_units=dev, hist_max_rel=rel)
        if case["solver"] == "bicgstab":
            assert rel <= _bicgstab_rtol(case["n1"]), case["name"]
        else:
            assert dev <= 1.0, case["name"]


def test_cg_device_generated_operator_64(K, ctx, oracle):
    A = oracle.poisson3d(64)
    ref = oracle.cg(A, np.ones(A.n), atol=0.0, rtol=1e-8, itmax=A.n, history=True)
    dA = K.CsrMatrix.stencil(ctx, "poisson", 64)
    b = ctx.empty(A.n)
    K.kfill_(b, 1.0)
    x, st, _ = K.cg(dA, b, atol=0.0, rtol=1e-8, itmax=A.n, history=True)
    assert st.niter == ref.niter == 159
    assert _hist_dev(st.residuals, ref.residuals) <= 1.0


def test_cg_device_resident_loop_equals_host_loop(K, ctx, oracle):
    """fused = 2 keeps alpha, beta, pNorm^2 and the stopping tests on the device (csrc/solver_device.hpp).  Same
    double operations in the same order as the host loop => histories, iteration counts, status strings and
    the final x are BIT-identical to fused = 1, for every way the loop can end."""
    A = oracle.poisson3d(24)
    dA = _upload(K, ctx, A)
    rng = np.random.default_rng(7)
    bh = rng.standard_normal(A.n)
    b = ctx.array(bh)
    for kw in (dict(), dict(rtol=1e-12, atol=0.0), dict(itmax=5), dict(itmax=1), dict(atol=0.0, rtol=0.0, itmax=23),
               dict(history=False), dict(x0=ctx.array(rng.standard_normal(A.n)))):
        kw = dict(dict(history=True), **kw)
        x1, st1, _ = K.cg(dA, b, fused=1, **kw)
        x2, st2, _ = K.cg(dA, b, fused=2, **kw)
        assert (st2.niter, st2.status, st2.solved, st2.inconsistent) == (st1.niter, st1.status, st1.solved, st1.inconsistent), kw
        if kw["history"]:
            assert np.array_equal(st2.residuals, st1.residuals), kw
        else:
            assert len(st2.residuals) == 0
        assert np.array_equal(x2.to_host(), x1.to_host()), kw
    # more iterations than the 4-iteration look-ahead chunks and than one history window would need a drain
    ref = oracle.cg(A, bh, history=True)
    x2, st2, ws = K.cg(dA, b, fused=2, history=True)
    assert st2.niter == ref.niter and _hist_dev(st2.residuals, ref.residuals) <= 1.0
    # the r vector of the workspace is the residual of the returned x (nothing ran after the stop)
    r = ws.vector("r").to_host()
    assert np.allclose(r, bh - A.matvec(x2.to_host()), atol=1e-9 * np.linalg.norm(bh))
    # a history longer than the device window: drained in pieces, still identical
    ctx.set_option("hist_window", 8)
    try:
        x5, st5, _ = K.cg(dA, b, fused=2, history=True, rtol=1e-12, atol=0.0)
    finally:
        ctx.set_option("hist_window", 1 << 14)
    x6, st6, _ = K.cg(dA, b, fused=1, history=True, rtol=1e-12, atol=0.0)
    assert st5.niter == st6.niter > 40 and np.array_equal(st5.residuals, st6.residuals)
    assert np.array_equal(x5.to_host(), x6.to_host())
    # zero curvature: singular operator, b in the null space direction gives p.Ap = 0
    Z = oracle.tridiag(6, 0.0, 0.0, 0.0)
    bz = ctx.array(np.ones(6))
    _, s1, _ = K.cg(_upload(K, ctx, Z), bz, fused=1)
    _, s2, _ = K.cg(_upload(K, ctx, Z), bz, fused=2)
    assert s1.status == s2.status == "zero curvature detected" and s2.niter == s1.niter == 0 and s2.inconsistent
    # fall back to the host loop whenever the mode does not apply (preconditioner, callback, linesearch, radius)
    x3, st3, _ = K.cg(dA, b, fused=2, radius=1.0)
    x4, st4, _ = K.cg(dA, b, fused=1, radius=1.0)
    assert st3.status == st4.status == "on trust-region boundary" and np.array_equal(x3.to_host(), x4.to_host())


@pytest.mark.parametrize("n1", [16, 32, 48])
def test_cg_single_reduction_variant_parity_budget(K, ctx, oracle, parity_log, n1):
    """options.variant = 1: Chronopoulos-Gear single-reduction CG (one reduction = one all-reduce per
    iteration).  NOT the reference's recurrence -- its own parity budget against the oracle's cg!: the same solution to
    the requested tolerance, an iteration count within 2, residual histories within 1e-6 relative while above 1e-6 r_0."""
    A = oracle.poisson3d(n1)
    b = np.ones(A.n)
    ref = oracle.cg(A, b, history=True)
    dA = _upload(K, ctx, A)
    x, st, ws = K.cg(dA, ctx.array(b), history=True, variant=1)
    assert st.solved and st.status == ref.status
    assert abs(st.niter - ref.niter) <= 2
    k = min(len(st.residuals), len(ref.residuals))
    big = ref.residuals[:k] >= 1e-6 * ref.residuals[0]
    dev = float(np.max(np.abs(st.residuals[:k] - ref.residuals[:k])[big] / ref.residuals[:k][big]))
    xh = x.to_host()
    res = np.linalg.norm(b - A.matvec(xh)) / np.linalg.norm(b)
    parity_log(test="cg_single_reduction", n1=n1, niter=st.niter, niter_ref=ref.niter, hist_max_rel=dev, res=res)
    assert dev <= 1e-6
    assert res <= 2e-8 * 1.5 or res <= 1.5 * np.linalg.norm(b - A.matvec(ref.x)) / np.linalg.norm(b)
    assert np.allclose(xh, ref.x, rtol=0, atol=1e-7 * np.abs(ref.x).max())
    # ways to stop / refuse
    _, st2, _ = K.cg(dA, ctx.array(b), variant=1, itmax=5)
    assert st2.niter == 5 and not st2.solved and st2.status == "maximum number of iterations exceeded"
    with pytest.raises(K.KhipError):
        K.cg(dA, ctx.array(b), variant=1, radius=1.0)
    with pytest.raises(K.KhipError):
        K.cg(dA, ctx.array(b), variant=1, M=K.Jacobi(dA))
    # compressed operator: same numbers (the SpMV and both dots are bit-identical)
    dC = _upload(K, ctx, A)
    assert dC.compress() > 0
    x3, st3, _ = K.cg(dC, ctx.array(b), history=True, variant=1)
    assert st3.niter == st.niter and np.array_equal(x3.to_host(), xh)


@pytest.mark.parametrize("n1", [16, 32, 48])
def test_cg_pipelined_variant_parity_budget(K, ctx, oracle, parity_log, n1):
    """options.variant = 2: pipelined CG (Ghysels & Vanroose 2014): one reduction per
    iteration and a product that does not depend on it.  NOT the reference's recurrence (the residual and its image are
    recurred, not recomputed) -- its own parity budget against the oracle's cg!: solved to the requested tolerance, iteration
    count within 3, residual histories within 1e-5 relative while above 1e-5 r_0, the TRUE residual within 5x the oracle's."""
    A = oracle.poisson3d(n1)
    b = np.ones(A.n)
    ref = oracle.cg(A, b, history=True)
    dA = _upload(K, ctx, A)
    x, st, ws = K.cg(dA, ctx.array(b), history=True, variant=2)
    assert st.solved and st.status == ref.status
    assert abs(st.niter - ref.niter) <= 3
    k = min(len(st.residuals), len(ref.residuals))
    big = ref.residuals[:k] >= 1e-5 * ref.residuals[0]
    dev = float(np.max(np.abs(st.residuals[:k] - ref.residuals[:k])[big] / ref.residuals[:k][big]))
    xh = x.to_host()
    res = np.linalg.norm(b - A.matvec(xh)) / np.linalg.norm(b)
    res_ref = np.linalg.norm(b - A.matvec(ref.x)) / np.linalg.norm(b)
    parity_log(test="cg_pipelined", n1=n1, niter=st.niter, niter_ref=ref.niter, hist_max_rel=dev, res=res, res_ref=res_ref)
    assert dev <= 1e-5
    assert res <= 5 * res_ref + 1e-12
    assert np.allclose(xh, ref.x, rtol=0, atol=1e-6 * np.abs(ref.x).max())
    _, st2, _ = K.cg(dA, ctx.array(b), variant=2, itmax=5)
    assert st2.niter == 5 and not st2.solved and st2.status == "maximum number of iterations exceeded"
    with pytest.raises(K.KhipError):
        K.cg(dA, ctx.array(b), variant=2, M=K.Jacobi(dA))
    with pytest.raises(K.KhipError):
        K.cg(dA, ctx.array(b), variant=3)
    # warm start (the tolerance is relative to the warm-start residual, as in cg!): same solution
    x0 = ctx.array(ref.x * (1 + 1e-3 * np.cos(np.arange(A.n))))
    x4, st4, _ = K.cg(dA, ctx.array(b), x0=x0, variant=2)
    assert st4.solved and np.allclose(x4.to_host(), ref.x, rtol=0, atol=1e-6 * np.abs(ref.x).max())


@pytest.mark.parametrize("n1,memory,sstep", [(12, 12, 4), (12, 10, 4), (16, 20, 1), (16, 20, 2), (16, 20, 5), (16, 24, 8), (20, 30, 4)])
def test_gmres_sstep_variant_parity_budget(K, ctx, oracle, parity_log, n1, memory, sstep):
    """options.variant = 2 of gmres!: s-step GMRES: blocks of s monomial-basis vectors,
    batched CGS2 + CholeskyQR2, the Hessenberg columns recovered on the host.  The same Krylov spaces as gmres!(restart), so the
    residual estimates agree with the oracle's up to rounding and the restart normalisation (the reference divides by the
    Givens estimate, this variant by the true norm): iteration count within 2, history within 1e-6 while above 1e-7 r_0, the
    TRUE residual within the tolerance."""
    A = oracle.kron_unsymmetric(n1)
    b = A.matvec(np.ones(A.n))
    ref = oracle.gmres(A, b, memory=memory, restart=True, history=True)
    dA = _upload(K, ctx, A)
    ctx.set_option("gmres_sstep", sstep)
    try:
        x, st, _ = K.gmres(dA, ctx.array(b), memory=memory, restart=True, history=True, variant=2)
    finally:
        ctx.set_option("gmres_sstep", 4)
    assert st.solved and st.status == ref.status, st.status
    assert abs(st.niter - ref.niter) <= 2, (st.niter, ref.niter)
    k = min(len(st.residuals), len(ref.residuals))
    big = ref.residuals[:k] >= 1e-7 * ref.residuals[0]
    dev = float(np.max(np.abs(st.residuals[:k] - ref.residuals[:k])[big] / ref.residuals[:k][big]))
    xh = x.to_host()
    res = np.linalg.norm(b - A.matvec(xh)) / np.linalg.norm(b)
    parity_log(test="gmres_sstep", n1=n1, memory=memory, s=sstep, niter=st.niter, niter_ref=ref.niter, hist_max_rel=dev, res=res)
    assert dev <= 1e-6, dev
    assert res <= 3e-8
    assert np.allclose(xh, 1.0, atol=1e-6)
    # refusals
    with pytest.raises(K.KhipError):
        K.gmres(dA, ctx.array(b), memory=memory, restart=False, variant=2)
    with pytest.raises(K.KhipError):
        K.gmres(dA, ctx.array(b), memory=memory, restart=True, variant=2, M=K.Jacobi(dA))
    _, st2, _ = K.gmres(dA, ctx.array(b), memory=memory, restart=True, variant=2, itmax=7)
    assert st2.niter == 7 and not st2.solved and st2.status == "maximum number of iterations exceeded"


def test_cg_edge_cases(K, ctx, oracle):
    A = oracle.tridiag(10, -1.0, 4.0, -1.0)                               # symmetric_definite(10)
    bh = A.matvec(np.arange(1.0, 11.0))
    dA, b = _upload(K, ctx, A), ctx.array(bh)
    for fused in (False, True, 2):
        x, st, _ = K.cg(dA, b, itmax=10, fused=fused)
        assert st.solved and np.linalg.norm(bh - A.matvec(x.to_host())) / np.linalg.norm(bh) <= 1e-6
    # zero right-hand side (test/test_cg.jl)
    x, st, _ = K.cg(dA, ctx.zeros(10))
    assert st.niter == 0 and st.status == "x is a zero-residual solution" and np.all(x.to_host() == 0)
    # itmax
    x, st, _ = K.cg(dA, b, itmax=2)
    assert (not st.solved) and st.niter == 2 and st.status == "maximum number of iterations exceeded"
    # warm start: same iterates as the oracle's warm start
    x0 = np.linspace(0.5, 9.5, 10)
    ref = oracle.cg(A, bh, x0=x0, history=True)
    x, st, _ = K.cg(dA, b, x0=ctx.array(x0), history=True)
    assert st.niter == ref.niter and np.allclose(x.to_host(), ref.x, atol=1e-12)
    # nonpositive curvature with linesearch (negative definite operator)
    Aneg = oracle.tridiag(10, 1.0, -4.0, 1.0)
    ref = oracle.cg(Aneg, bh, linesearch=True)
    x, st, _ = K.cg(_upload(K, ctx, Aneg), b, linesearch=True)
    assert st.niter == ref.niter == 0 and st.indefinite and st.npcCount == 1 and st.status == "nonpositive curvature"
    assert np.array_equal(x.to_host(), ref.x)
    # trust region
    ref = oracle.cg(A, bh, radius=1.0)
    x, st, _ = K.cg(dA, b, radius=1.0)
    assert st.status == ref.status == "on trust-region boundary" and st.niter == ref.niter
    assert math.isclose(np.linalg.norm(x.to_host()), 1.0, rel_tol=1e-10)
    assert np.allclose(x.to_host(), ref.x, atol=1e-13)
    # operator that is not positive definite -> error, message as the reference's
    with pytest.raises(K.KhipError) as e:
        K.cg(_upload(K, ctx, Aneg), b, M=lambda r, z: K.kscalcopy_(10, z, -1.0, r) and None)
    # callback(workspace)::Bool requesting exit after 3 iterations
    seen = []

    def cb(ws):
        seen.append(K.knorm(10, ws.vector("r")))
        return len(seen) >= 3
    x, st, _ = K.cg(dA, b, callback=cb, atol=0.0, rtol=0.0)
    assert st.niter == 3 and st.status == "user-requested exit" and len(seen) == 3


def test_cg_jacobi_preconditioner_callable(K, ctx, oracle):
    """M given as a device callable (operator contract docs/src/matrix_free.md:32-34); the Poisson diagonal is 6."""
    A = oracle.poisson3d(12)
    bh = np.ones(A.n)
    ref = oracle.cg(A, bh, M=lambda r: r / 6.0, history=True)
    x, st, ws = K.cg(_upload(K, ctx, A), ctx.array(bh), M=lambda r, z: K.kdivcopy_(A.n, z, r, 6.0), history=True)
    assert st.niter == ref.niter and _hist_dev(st.residuals, ref.residuals) <= 1.0
    assert ws.nbytes == 5 * 8 * A.n      # z allocated lazily (src/cg.jl:142)


def test_jacobi_preconditioner_native(K, ctx, oracle, parity_log):
    """Built-in Jacobi operator (khip_jacobi_create: z = r ./ diag(A)) on an SPD matrix with a strongly varying
    diagonal; the oracle applies M = Diagonal(1 ./ diag(A)) as the reference's examples do
    (test/test_gmres.jl:105-128).  Also checks the primitives khip_vmul / khip_vdiv / khip_csr_diagonal."""
    import scipy.sparse as sp
    rng = np.random.default_rng(12)
    P = oracle.poisson3d(10).to_scipy()
    dvar = np.exp(rng.uniform(0, 6, P.shape[0]))
    S = (P + sp.diags(dvar)).tocsr()
    S.sort_indices()
    n = S.shape[0]
    d = S.diagonal()
    bh = S @ np.ones(n)
    dA = K.CsrMatrix.from_scipy(ctx, S)
    assert np.array_equal(dA.diagonal().to_host(), d)
    a, b2 = rng.standard_normal(n), rng.standard_normal(n) + 3.0
    w = ctx.empty(n)
    assert np.array_equal(K.kvmul_(n, w, ctx.array(a), ctx.array(b2)).to_host(), a * b2)
    assert np.array_equal(K.kvdiv_(n, w, ctx.array(a), ctx.array(b2)).to_host(), a / b2)
    ref = oracle.cg(lambda v: S @ v, bh, M=lambda r: r / d, history=True)
    ref0 = oracle.cg(lambda v: S @ v, bh, history=True)
    x, st, _ = K.cg(dA, ctx.array(bh), M=K.Jacobi(dA), history=True)
    assert st.solved and st.niter == ref.niter < ref0.niter        # preconditioning pays off
    dev = _hist_dev(st.residuals, ref.residuals)
    parity_log(test="cg_jacobi_native", niter=st.niter, niter_unpreconditioned=ref0.niter, hist_tol_units=dev)
    assert dev <= 1.0
    # the same operator as right preconditioner of GMRES and BiCGSTAB
    refg = oracle.gmres(lambda v: S @ v, bh, N=lambda r: r / d, memory=20, history=True)
    x, stg, _ = K.gmres(dA, ctx.array(bh), N=K.Jacobi(dA), memory=20, history=True)
    assert stg.niter == refg.niter and _hist_dev(stg.residuals, refg.residuals) <= 1.0
    refb = oracle.bicgstab(lambda v: S @ v, bh, M=lambda r: r / d, history=True)
    x, stb, _ = K.bicgstab(dA, ctx.array(bh), M=K.Jacobi(dA), history=True)
    assert stb.niter == refb.niter and _hist_rel(stb.residuals, refb.residuals) <= _bicgstab_rtol(16)


# ------------------------------------------------------------------------------ GMRES

@pytest.mark.parametrize("n1", [8, 16])
@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("kw", [dict(), dict(restart=True), dict(restart=True, reorthogonalization=True)])
def test_gmres_kron_unsymmetric_matches_oracle(K, ctx, oracle, parity_log, n1, fused, kw):
    """cfg 3's operator at oracle-sized grids (kron_unsymmetric, test/test_utils.jl:160-169)."""
    A = oracle.kron_unsymmetric(n1)
    bh = A.matvec(np.ones(A.n))
    ref = oracle.gmres(A, bh, memory=10, history=True, **kw)
    x, st, ws = K.gmres(_upload(K, ctx, A), ctx.array(bh), memory=10, history=True, fused=fused, **kw)
    assert st.solved and st.status == ref.status and st.niter == ref.niter
    dev = _hist_dev(st.residuals, ref.residuals)
    parity_log(test="gmres_kron", n1=n1, fused=fused, kw=kw, niter=st.niter, hist_tol_units=dev,
               hist_max_rel=_hist_rel(st.residuals, ref.residuals))
    assert dev <= 1.0
    S = A.to_scipy()
    assert np.linalg.norm(bh - S @ x.to_host()) / np.linalg.norm(bh) <= 1e-6       # test/test_gmres.jl:93-129
    assert np.allclose(x.to_host(), ref.x, atol=1e-9)


def test_gmres_memory30_restart_cfg3_shape(K, ctx, oracle, parity_log):
    A = oracle.kron_unsymmetric(20)
    bh = A.matvec(np.ones(A.n))
    ref = oracle.gmres(A, bh, memory=30, restart=True, history=True, atol=1e-10, rtol=1e-10)
    x, st, _ = K.gmres(_upload(K, ctx, A), ctx.array(bh), memory=30, restart=True, history=True, atol=1e-10, rtol=1e-10)
    assert st.niter == ref.niter and st.status == ref.status
    dev = _hist_dev(st.residuals, ref.residuals)
    parity_log(test="gmres_mem30", niter=st.niter, hist_tol_units=dev, hist_max_rel=_hist_rel(st.residuals, ref.residuals))
    assert dev <= 1.0


def test_gmres_look_ahead_equals_plain_sequence(K, ctx, oracle):
    """fused = 2 for gmres! (M = N = I, CSR operator): V[k+1] = q / ||q|| (device scalar) and the next product are enqueued
    before the host reads the Hessenberg column -- same operations on the same values, so histories, iteration counts
    and x are bit-identical to fused = 1, with and without restart, when the loop ends on convergence or on itmax."""
    A = oracle.kron_unsymmetric(10)
    dA = _upload(K, ctx, A)
    rng = np.random.default_rng(21)
    bh = A.matvec(rng.standard_normal(A.n))
    b = ctx.array(bh)
    for kw in (dict(memory=10, restart=True), dict(memory=10), dict(memory=20), dict(memory=5, restart=True, itmax=13),
               dict(memory=30, rtol=1e-12, atol=0.0), dict(memory=8, restart=True, history=False), dict(memory=4, itmax=9)):
        kw = dict(dict(history=True), **kw)
        x1, st1, _ = K.gmres(dA, b, fused=1, **kw)
        x2, st2, _ = K.gmres(dA, b, fused=2, **kw)
        assert (st2.niter, st2.status, st2.solved, st2.inconsistent) == (st1.niter, st1.status, st1.solved, st1.inconsistent), kw
        if kw["history"]:
            assert np.array_equal(st2.residuals, st1.residuals), kw
        assert np.array_equal(x2.to_host(), x1.to_host()), kw
    ref = oracle.gmres(A, bh, memory=10, restart=True, history=True)
    _, st, _ = K.gmres(dA, b, memory=10, restart=True, history=True, fused=2)
    assert st.niter == ref.niter


def test_gmres_edge_cases(K, ctx, oracle):
    A = oracle.kron_unsymmetric(6)
    bh = A.matvec(np.ones(A.n))
    dA = _upload(K, ctx, A)
    x, st, _ = K.gmres(dA, ctx.zeros(A.n))
    assert st.niter == 0 and st.status == "x is a zero-residual solution"
    # Jacobi preconditioners (left, right) as device callables; kron_unsymmetric has diagonal 12
    for side in ("M", "N"):
        refkw = {side: (lambda v: v / 12.0)}
        devkw = {side: (lambda v, out: K.kdivcopy_(A.n, out, v, 12.0))}
        ref = oracle.gmres(A, bh, memory=10, restart=True, history=True, **refkw)
        x, st, _ = K.gmres(dA, ctx.array(bh), memory=10, restart=True, history=True, **devkw)
        assert st.niter == ref.niter and _hist_dev(st.residuals, ref.residuals) <= 1.0
    # basis growth when restart = false and memory is small (src/gmres.jl:244-252,319-324)
    ref = oracle.gmres(A, bh, memory=3, history=True)
    x, st, ws = K.gmres(dA, ctx.array(bh), memory=3, history=True)
    assert st.niter == ref.niter > 3 and _hist_dev(st.residuals, ref.residuals) <= 1.0
    # warm start
    x0 = np.full(A.n, 0.9)
    ref = oracle.gmres(A, bh, x0=x0, memory=10, restart=True)
    x, st, _ = K.gmres(dA, ctx.array(bh), x0=ctx.array(x0), memory=10, restart=True)
    assert st.niter == ref.niter and np.allclose(x.to_host(), ref.x, atol=1e-10)
    # itmax
    x, st, _ = K.gmres(dA, ctx.array(bh), memory=10, itmax=4)
    assert st.niter == 4 and st.status == "maximum number of iterations exceeded"


# ------------------------------------------------------------------------------ BiCGSTAB

@pytest.mark.parametrize("n1", [8, 16])
@pytest.mark.parametrize("fused", [False, True])
def test_bicgstab_matches_oracle(K, ctx, oracle, parity_log, n1, fused):
    A = oracle.kron_unsymmetric(n1)
    bh = A.matvec(np.ones(A.n))
    ref = oracle.bicgstab(A, bh, history=True)
    x, st, ws = K.bicgstab(_upload(K, ctx, A), ctx.array(bh), history=True, fused=fused)
    assert st.solved and st.status == ref.status and st.niter == ref.niter
    dev = _hist_dev(st.residuals, ref.residuals)
    parity_log(test="bicgstab_kron", n1=n1, fused=fused, niter=st.niter, hist_tol_units=dev,
               hist_max_rel=_hist_rel(st.residuals, ref.residuals))
    assert _hist_rel(st.residuals, ref.residuals) <= _bicgstab_rtol(n1)          # 9.5e-11 at 8^3, 1.4e-7 at 16^3
    S = A.to_scipy()
    assert np.linalg.norm(bh - S @ x.to_host()) / np.linalg.norm(bh) <= 1e-6       # test/test_bicgstab.jl:39-45
    assert ws.nbytes == 6 * 8 * A.n                                                # storage 6n


def test_bicgstab_device_resident_loop_equals_host_loop(K, ctx, oracle):
    """fused = 2 for bicgstab! (M = N = I, CSR operator): rho, alpha, omega, beta and the stopping tests live on
    the device (csrc/solver_device.hpp); histories, iteration counts, status strings and x are bit-identical to
    fused = 1 however the loop ends."""
    A = oracle.kron_unsymmetric(10)
    dA = _upload(K, ctx, A)
    rng = np.random.default_rng(11)
    bh = A.matvec(rng.standard_normal(A.n))
    b = ctx.array(bh)
    for kw in (dict(), dict(rtol=1e-12, atol=0.0), dict(itmax=3), dict(itmax=1), dict(atol=0.0, rtol=0.0, itmax=17),
               dict(history=False), dict(x0=ctx.array(rng.standard_normal(A.n))), dict(c=ctx.array(rng.standard_normal(A.n)))):
        kw = dict(dict(history=True), **kw)
        x1, st1, _ = K.bicgstab(dA, b, fused=1, **kw)
        x2, st2, _ = K.bicgstab(dA, b, fused=2, **kw)
        assert (st2.niter, st2.status, st2.solved) == (st1.niter, st1.status, st1.solved), kw
        if kw["history"]:
            assert np.array_equal(st2.residuals, st1.residuals), kw
        assert np.array_equal(x2.to_host(), x1.to_host()), kw
    ref = oracle.bicgstab(A, bh, history=True)
    x2, st2, _ = K.bicgstab(dA, b, fused=2, history=True)
    assert st2.niter == ref.niter and np.max(np.abs(st2.residuals - ref.residuals) / ref.residuals) <= 1e-9   # random b, 10^3: measured 1e-11
    ctx.set_option("hist_window", 8)
    try:
        x3, st3, _ = K.bicgstab(dA, b, fused=2, history=True, rtol=1e-13, atol=0.0)
    finally:
        ctx.set_option("hist_window", 1 << 14)
    x4, st4, _ = K.bicgstab(dA, b, fused=1, history=True, rtol=1e-13, atol=0.0)
    assert st3.niter == st4.niter > 8 and np.array_equal(st3.residuals, st4.residuals) and np.array_equal(x3.to_host(), x4.to_host())
    # preconditioned or callback runs fall back to the host loop
    P = K.Jacobi(dA)
    x5, st5, _ = K.bicgstab(dA, b, M=P, fused=2)
    x6, st6, _ = K.bicgstab(dA, b, M=P, fused=1)
    assert st5.niter == st6.niter and np.array_equal(x5.to_host(), x6.to_host())


def test_bicgstab_edge_cases(K, ctx, oracle):
    A = oracle.kron_unsymmetric(6)
    bh = A.matvec(np.ones(A.n))
    dA = _upload(K, ctx, A)
    x, st, _ = K.bicgstab(dA, ctx.zeros(A.n))
    assert st.niter == 0 and st.status == "x is a zero-residual solution"
    x, st, _ = K.bicgstab(dA, ctx.array(bh), c=ctx.zeros(A.n))
    assert st.status == "Breakdown bᴴc = 0" and not st.solved
    ref = oracle.bicgstab(A, bh, M=lambda v: v / 12.0, N=lambda v: v.copy(), history=True)
    x, st, _ = K.bicgstab(dA, ctx.array(bh), M=lambda v, o: K.kdivcopy_(A.n, o, v, 12.0),
                          N=lambda v, o: K.kcopy_(A.n, o, v), history=True)
    assert st.niter == ref.niter and st.solved
    x0 = np.full(A.n, 0.9)
    ref = oracle.bicgstab(A, bh, x0=x0)
    x, st, _ = K.bicgstab(dA, ctx.array(bh), x0=ctx.array(x0))
    assert st.niter == ref.niter and np.allclose(x.to_host(), ref.x, atol=1e-9)


# ------------------------------------------------------------------ full-size properties (cfg 2)

def test_cg_full_size_512_properties(K, ctx, parity_log):
    """cfg 2: 40 CG iterations at 512^3.  Size-independent checks: (1) the residual norm from the
    recurrence equals the recomputed ||b - A x|| (x, r stay consistent), (2) the first residual norms
    equal the 64^3 ... no: equal sqrt(n) at k = 0, (3) fused and unfused paths give the same history."""
    n1 = 512
    n = n1 ** 3
    A = K.CsrMatrix.stencil(ctx, "poisson", n1)
    b = ctx.empty(n)
    K.kfill_(b, 1.0)
    ws = K.CgWorkspace(ctx, n, n)
    K.cg_(ws, A, b, atol=0.0, rtol=0.0, itmax=40, history=True, fused=True)
    st = ws.stats
    assert st.niter == 40 and st.status == "maximum number of iterations exceeded"
    assert st.residuals[0] == math.sqrt(n)
    hist_fused = st.residuals.copy()
    t = ctx.empty(n)
    A.matvec(ws.x, t)
    K.kaxpby_(n, 1.0, b, -1.0, t)                 # t = b - A x
    true_res = K.knorm(n, t)
    assert abs(true_res - hist_fused[-1]) <= 1e-10 * hist_fused[0]
    K.cg_(ws, A, b, atol=0.0, rtol=0.0, itmax=40, history=True, fused=False)
    dev = float(np.max(np.abs(ws.stats.residuals - hist_fused) / hist_fused))
    parity_log(test="cg_512_fused_vs_unfused", hist_max_rel=dev, true_res_gap=abs(true_res - hist_fused[-1]))
    assert dev <= 1e-11


def test_no_device_memory_leak_over_object_lifecycles(K, oracle):
    """Create / use / drop every kind of object repeatedly on a fresh context: free device memory must come back
    (workspaces incl. their device-resident loop state, CSR handles with templates / transposes / panel halos,
    preconditioners with their captured graphs, panels)."""
    import gc
    ctx = K.Context(0)
    A_cpu = oracle.poisson3d(12)
    B_cpu = oracle.kron_unsymmetric(8)

    def cycle():
        A = K.CsrMatrix.from_host(ctx, A_cpu.rowptr, A_cpu.col, A_cpu.val, (A_cpu.n, A_cpu.n))
        Bm = K.CsrMatrix.from_host(ctx, B_cpu.rowptr, B_cpu.col, B_cpu.val, (B_cpu.n, B_cpu.n))
        b = ctx.array(np.ones(A_cpu.n))
        for kw in (dict(fused=0), dict(fused=1), dict(fused=2), dict(variant=1), dict(fused=2, history=True)):
            K.cg(A, b, **kw)
        P = K.Ilu0(A)
        K.cg(A, b, M=P)
        J = K.Jacobi(A)
        K.cg(A, b, M=J)
        At = Bm.transpose()
        Bm.compress()
        bb = ctx.array(B_cpu.matvec(np.ones(B_cpu.n)))
        K.bicgstab(Bm, bb, fused=2, history=True)
        K.gmres(Bm, bb, memory=10, restart=True)
        K.gmres(At, bb, memory=10)
        rng = np.random.default_rng(0)
        K.block_gmres(Bm, rng.standard_normal((B_cpu.n, 4)), memory=4)
        del A, Bm, b, P, J, At, bb
        gc.collect()
        ctx.sync()

    cycle()                                  # first cycle grows the context's own scratch (reduction partials, panels)
    free0, _ = ctx.mem_info()
    for _ in range(5):
        cycle()
    free1, _ = ctx.mem_info()
    ctx.close()
    assert free0 - free1 <= 8 << 20, (free0, free1)      # allow allocator granularity, not a per-cycle leak


# ------------------------------------------------------------------ product helpers and storage formulas

def test_product_to_boundary_known_answers(K, ctx):
    """test/test_aux.jl:81-95 against the to_boundary the trust-region cg! runs (csrc/solvers.cpp; dots on the device)."""
    import ctypes as C
    n = 5
    x = np.ones(n)
    d = np.ones(n)
    d[0:n:2] = -1

    def tb(xv, dv, radius, flip=False):
        s1, s2 = C.c_double(), C.c_double()
        dx, dd = ctx.array(xv), ctx.array(dv)
        rc = K.lib().khip_test_to_boundary(ctx._h, n, dx.ptr, dd.ptr, radius, int(flip), C.byref(s1), C.byref(s2))
        if rc != 0:
            raise ValueError(K.lib().khip_last_error().decode())
        return s1.value, s2.value
    for bad in (-1.0, 0.5):                                  # "radius must be positive", "outside of the trust region"
        with pytest.raises(ValueError):
            tb(x, d, bad)
    with pytest.raises(ValueError):                          # "zero direction"
        tb(x, np.zeros(n), 1.0)
    r = tb(x, d, 5.0)
    assert math.isclose(max(r), 2.209975124224178, rel_tol=1e-14)
    assert math.isclose(min(r), -1.8099751242241782, rel_tol=1e-14)
    r = tb(x, d, 5.0, flip=True)
    assert math.isclose(max(r), 1.8099751242241782, rel_tol=1e-14)
    assert math.isclose(min(r), -2.209975124224178, rel_tol=1e-14)


@pytest.mark.parametrize("mem", [5, 30])
def test_gmres_storage_formula(K, ctx, oracle, mem):
    """test/test_allocations.jl:251-270: GMRES needs 2 n-vectors (x, w), mem n-vectors (V), 3 mem-vectors (c, s, z) and a
    packed triangle of mem (mem + 1) / 2 -- exactly that, plus the (mem + 1)-entry landing buffer of the look-ahead fetch;
    a solve allocates nothing more (restart = true), Δx / p / q appear only with warm start / preconditioners."""
    A = oracle.kron_unsymmetric(8)
    n = A.n
    dA = _upload(K, ctx, A)
    b = ctx.array(A.matvec(np.ones(n)))
    ws = K.GmresWorkspace(ctx, n, n, memory=mem)
    formula = 8 * (2 * n + n * mem + 2 * mem + mem * (mem + 1) // 2) + 8 * mem
    assert ws.nbytes == formula + 8 * (mem + 1)
    K.gmres_(ws, dA, b, restart=True)
    assert ws.stats.solved and ws.nbytes == formula + 8 * (mem + 1) + 8 * n      # + Δx: allocate_if(restart, ...), src/gmres.jl:141
    K.gmres_(ws, dA, b, restart=True)
    assert ws.nbytes == formula + 8 * (mem + 1) + 8 * n                           # the in-place solve allocates nothing
    K.gmres_(ws, dA, b, restart=True, N=lambda v, o: K.kcopy_(n, o, v))
    assert ws.nbytes == formula + 8 * (mem + 1) + 2 * 8 * n                       # + p (right preconditioner), src/gmres.jl:143


def test_block_gmres_storage_formula(K, ctx, oracle):
    """test/test_allocations.jl:734-761: 2 (n p) blocks X, W; C (p p); D (2p p); mem tau (p), V (n p), Z (p p), H (2p p);
    mem (mem + 1) / 2 R (p p).  This implementation holds in addition the row-major panel copy of B and 2 mem + 1 staging
    blocks (p p) of the fused sweeps, reported separately; a restarted solve adds ΔX (n p) once, nothing per call."""
    A = oracle.kron_unsymmetric(8)
    n, p, mem = A.n, 4, 6
    dA = _upload(K, ctx, A)
    S = A.to_scipy()
    t = (np.arange(n) + 1.0) / n
    B = S @ np.stack([t ** j for j in range(p)], axis=1)
    ws = K.BlockGmresWorkspace(ctx, n, n, p, memory=mem)
    formula = 8 * (2 * n * p + p * p + 2 * p * p + mem * p + mem * n * p + mem * p * p + (mem * (mem + 1) // 2) * p * p + mem * 2 * p * p)
    # a workspace on the caller's panels (the default of the Python mirror, as julia/KrylovHIP) reads B in place: no panel copy
    extra = (0 if ws.adopted else 8 * n * p) + 8 * (2 * mem + 1) * p * p
    assert ws.nbytes_extra == extra and ws.nbytes == formula + extra
    wo = K.BlockGmresWorkspace(ctx, n, n, p, memory=mem, adopt=False)
    assert wo.nbytes_extra == 8 * n * p + 8 * (2 * mem + 1) * p * p and wo.nbytes == formula + wo.nbytes_extra
    Bd = ctx.array(np.asfortranarray(B).ravel(order="F"))
    K.block_gmres_(ws, dA, Bd, itmax=mem)                                          # as the reference's test: itmax = mem, no basis growth
    assert ws.stats.niter == mem and ws.nbytes == formula + extra                  # the in-place solve allocated nothing
    K.block_gmres_(ws, dA, Bd, restart=True)
    assert ws.nbytes == formula + extra + 8 * n * p                                # + ΔX, src/block_gmres.jl:145
    K.block_gmres_(ws, dA, Bd, restart=True)
    assert ws.nbytes == formula + extra + 8 * n * p


@pytest.mark.parametrize("n1,kw", [(8, dict()), (16, dict(restart=True)), (12, dict(restart=True, memory=30)), (10, dict(memory=4))])
def test_gmres_cgs2_variant_parity_budget(K, ctx, oracle, parity_log, n1, kw):
    """gmres!(variant = 1): classical Gram-Schmidt applied twice (CGS2) instead of the reference's modified Gram-Schmidt
    cascade (src/gmres.jl:259-271) -- an opt-in, communication-reducing form, NOT the reference's recurrence.
    Its parity budget: same solution to the requested tolerance, iteration count within 1 of the oracle's gmres!, residual
    history within 1e-6 relative while the residual is above 1e-6 r_0.  Measured: identical counts, histories within 1e-9."""
    A = oracle.kron_unsymmetric(n1)
    bh = A.matvec(np.ones(A.n))
    kw = dict(dict(memory=10), **kw)
    ref = oracle.gmres(A, bh, history=True, **kw)
    dA = _upload(K, ctx, A)
    x, st, _ = K.gmres(dA, ctx.array(bh), history=True, variant=1, **kw)
    assert st.solved and abs(st.niter - ref.niter) <= 1
    k = min(len(st.residuals), len(ref.residuals))
    h, hr = st.residuals[:k], ref.residuals[:k]
    big = hr > 1e-6 * hr[0]
    dev = float(np.max(np.abs(h[big] - hr[big]) / hr[big]))
    parity_log(test="gmres_cgs2", n1=n1, kw=kw, niter=st.niter, niter_oracle=ref.niter, hist_max_rel_above_1em6=dev)
    assert dev <= 1e-6
    S = A.to_scipy()
    assert np.linalg.norm(bh - S @ x.to_host()) / np.linalg.norm(bh) <= 1e-6
    with pytest.raises(K.KhipError):
        K.gmres(dA, ctx.array(bh), variant=7)


# ---- options.verbose and stats.allocation_timer --------------------------------------------------

def test_verbose_prints_the_reference_log_and_changes_nothing(K, ctx, oracle, capfd):
    """verbose > 0: the reference's per-iteration rows on stdout (src/cg.jl:132,182-183,224,267-269; src/gmres.jl:131,191-192,
    315,364; src/bicgstab.jl:135,193-194,255-257; src/block_gmres.jl:120,181-182,297,340), one every `verbose` iterations
    (kdisplay, src/krylov_utils.jl:301); same iteration count and the same residual history as the silent solve."""
    import re
    n1 = 16
    A = K.CsrMatrix.stencil(ctx, "poisson", n1)
    n = n1 ** 3
    b = ctx.empty(n)
    K.kfill_(b, 1.0)
    _, st0, _ = K.cg(A, b, history=True)
    capfd.readouterr()
    _, st1, _ = K.cg(A, b, history=True, verbose=10)
    def log_of(title):          # the solver's log inside whatever else the process wrote to fd 1 meanwhile (RCCL prints a banner there)
        text = capfd.readouterr().out
        assert title in text, text[-400:]
        return text[text.index(title):]

    out = log_of("CG: system")
    assert st1.niter == st0.niter and np.array_equal(st1.residuals, st0.residuals)
    lines = out.split("\n")
    assert lines[0] == f"CG: system of {n} equations in {n} variables"
    assert lines[1].split() == ["k", "‖r‖", "pAp", "α", "σ", "timer"] and lines[1].startswith("    k      ‖r‖       pAp")
    rows = [l for l in lines[2:] if l.strip()]
    assert [int(r.split()[0]) for r in rows] == list(range(0, st0.niter + 1, 10))
    assert re.fullmatch(r"    0  6\.4e\+01   1\.5e\+03   2\.7e\+00   2\.7e\+00  \d+\.\d\ds", rows[0]), rows[0]   # ‖r0‖ = 64, pAp = 6 * 16^2 = 1536 (p = ones: only the boundary rows have a nonzero sum), α = σ = 4096 / 1536
    for r in rows[:-1]:
        f = r.split()
        assert len(f) == 6 and f[5].endswith("s")
    assert out.endswith("\n\n")
    # gmres!, bicgstab!, block_gmres!: title, header, rows every `verbose` iterations, unchanged results
    Au = K.CsrMatrix.stencil(ctx, "kron_unsymmetric", 10)
    nu = 1000
    ones = ctx.empty(nu)
    K.kfill_(ones, 1.0)
    bu = Au.matvec(ones)
    _, g0, _ = K.gmres(Au, bu, memory=10, restart=True, history=True)
    capfd.readouterr()
    _, g1, _ = K.gmres(Au, bu, memory=10, restart=True, history=True, verbose=5)
    out = log_of("GMRES: system").split("\n")
    assert g1.niter == g0.niter and np.array_equal(g1.residuals, g0.residuals)
    assert out[0] == f"GMRES: system of size {nu}" and out[1].split() == ["pass", "k", "‖rₖ‖", "hₖ₊₁.ₖ", "timer"]
    assert out[2].split()[:3] == ["0", "0", "%.1e" % g0.residuals[0]] and "✗ ✗ ✗ ✗" in out[2]
    ks = [int(l.split()[1]) for l in out[3:] if l.strip()]
    assert ks == [k for k in range(5, g0.niter + 1, 5)]
    _, s0, _ = K.bicgstab(Au, bu, history=True)
    capfd.readouterr()
    _, s1, _ = K.bicgstab(Au, bu, history=True, verbose=4)
    out = log_of("BICGSTAB: system").split("\n")
    assert s1.niter == s0.niter and np.array_equal(s1.residuals, s0.residuals)
    assert out[0] == f"BICGSTAB: system of size {nu}" and out[1].split() == ["k", "‖rₖ‖", "|αₖ|", "|ωₖ|", "timer"]
    assert out[2].split()[:4] == ["0", "%.1e" % s0.residuals[0], "1.0e+00", "1.0e+00"]
    B = np.random.default_rng(3).standard_normal((nu, 4))
    capfd.readouterr()
    X1, b1, _ = K.block_gmres(Au, B, memory=5, ctx=ctx, history=True, verbose=3)
    out = log_of("BLOCK-GMRES: system").split("\n")
    X0, b0, _ = K.block_gmres(Au, B, memory=5, ctx=ctx, history=True)
    assert b1.niter == b0.niter and np.array_equal(np.array(b1.residuals), np.array(b0.residuals)) and np.array_equal(X1, X0)
    assert out[0] == f"BLOCK-GMRES: system of size {nu} with 4 right-hand sides" and out[1].split() == ["pass", "k", "‖Rₖ‖", "timer"]


def test_verbose_log_goes_to_the_callers_iostream(K, ctx, capfd, tmp_path):
    """options.log_fd = the reference's `iostream` keyword (src/cg.jl:24,182-183): the same rows, written to the caller's file
    descriptor instead of stdout; variants with another recurrence have no such rows and refuse verbose."""
    n1 = 12
    n = n1 ** 3
    A = K.CsrMatrix.stencil(ctx, "poisson", n1)
    b = ctx.empty(n)
    K.kfill_(b, 1.0)
    capfd.readouterr()
    _, st0, _ = K.cg(A, b, history=True, verbose=5)
    on_stdout = capfd.readouterr().out
    on_stdout = on_stdout[on_stdout.index("CG: system"):]
    path = tmp_path / "cg.log"
    with open(path, "wb") as f:
        _, st1, _ = K.cg(A, b, history=True, verbose=5, log_fd=f.fileno())
    assert "CG: system" not in capfd.readouterr().out                  # nothing on stdout this time
    text = path.read_text()
    strip = lambda t: [" ".join(l.split()[:-1]) for l in t.split("\n")]   # drop the timer column
    assert strip(text) == strip(on_stdout) and text.startswith(f"CG: system of {n} equations in {n} variables\n")
    assert st1.niter == st0.niter and np.array_equal(st1.residuals, st0.residuals)
    Au = K.CsrMatrix.stencil(ctx, "kron_unsymmetric", 8)
    ones = ctx.empty(512)
    K.kfill_(ones, 1.0)
    bu = Au.matvec(ones)
    with open(tmp_path / "rest.log", "wb") as f:
        K.gmres(Au, bu, memory=10, restart=True, verbose=3, log_fd=f.fileno())
        K.bicgstab(Au, bu, verbose=3, log_fd=f.fileno())
        K.block_gmres(Au, np.random.default_rng(5).standard_normal((512, 2)), memory=4, ctx=ctx, verbose=2, log_fd=f.fileno())
    rest = (tmp_path / "rest.log").read_text()
    assert rest.index("GMRES: system of size 512") < rest.index("BICGSTAB: system of size 512") < rest.index("BLOCK-GMRES: system of size 512 with 2")
    assert "system of size" not in capfd.readouterr().out
    for variant in (1, 2):
        with pytest.raises(K.KhipError) as e:
            K.cg(A, b, variant=variant, verbose=1)
        assert "variant = 0" in str(e.value)


def test_allocation_timer_counts_creation_and_lazy_allocations(K, ctx):
    """stats.allocation_timer (src/krylov_workspaces.jl:288-289; allocate_if, src/krylov_utils.jl:281-288): set when the
    workspace allocates its vectors, increased by the lazy ones (z for a preconditioned cg!, Δx for a warm start), not by
    a solve that allocates nothing."""
    n1 = 20
    n = n1 ** 3
    A = K.CsrMatrix.stencil(ctx, "poisson", n1)
    b = ctx.empty(n)
    K.kfill_(b, 1.0)
    ws = K.CgWorkspace(ctx, n, n)
    t_create = ws.stats.allocation_timer
    assert t_create > 0.0
    K.cg_(ws, A, b)
    assert ws.stats.allocation_timer == t_create                       # the 4 n vectors were there: nothing allocated
    K.cg_(ws, A, b, M=K.Jacobi(A))
    assert ws.stats.allocation_timer > t_create                        # z allocated on first use (src/cg.jl:142)
    wsb = K.BlockGmresWorkspace(ctx, n, n, 4, memory=3)
    assert wsb.stats.allocation_timer > 0.0
