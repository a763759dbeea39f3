"""The distributed code path on ONE GPU: a 1-rank RCCL communicator exercises comm_init, the halo-plan
construction on the device (collect / sort / remap kernels), the [owned | ghost] SpMV variant, the
all-gathered (hi, lo) dot and the interior / boundary split -- everything except the actual peer
traffic, which needs more than one GPU (covered on CPU by tests/test_dist_gloo.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dctx(K):
    c = K.Context(0)
    c.comm_init(0, 1, K.Context.comm_unique_id())
    yield c
    c.close()


def test_single_rank_communicator_matches_plain_path(K, ctx, dctx, oracle):
    n1 = 24
    A_cpu = oracle.poisson3d(n1)
    n = A_cpu.n
    Ad = K.CsrMatrix.stencil(dctx, "poisson", n1, rows=(0, n), distributed=True)
    Ap = K.CsrMatrix.stencil(ctx, "poisson", n1)
    x = np.linspace(-1, 1, n) ** 3
    yd = Ad.matvec(dctx.array(x)).to_host()
    assert np.array_equal(yd, Ap.matvec(ctx.array(x)).to_host())
    assert np.array_equal(yd, A_cpu.matvec(x))
    # dot through the all-gather path == plain path
    a, b = dctx.array(x), dctx.array(x[::-1].copy())
    assert K.kdot(n, a, b) == K.kdot(n, ctx.array(x), ctx.array(x[::-1].copy()))
    # CG: identical history
    bd = dctx.empty(n); K.kfill_(bd, 1.0)
    bp = ctx.empty(n); K.kfill_(bp, 1.0)
    _, st_d, _ = K.cg(Ad, bd, history=True)
    _, st_p, _ = K.cg(Ap, bp, history=True)
    ref = oracle.cg(A_cpu, np.ones(n), history=True)
    assert st_d.niter == st_p.niter == ref.niter
    assert np.array_equal(st_d.residuals, st_p.residuals)
    dctx.barrier()


def test_row_slab_with_ghost_columns_single_process(K, dctx, oracle):
    """A middle slab [r0, r1) of the grid as a 'distributed' operator needs ghost planes on both sides; with one
    rank nobody can own them, so creation must fail loudly (partition does not cover the operator)."""
    n1 = 12
    n = n1 ** 3
    with pytest.raises(K.KhipError):
        K.CsrMatrix.stencil(dctx, "poisson", n1, rows=(n // 4, n // 2), distributed=True)


# ---------------------------------------------------------------------------------------------------
# Several ranks on ONE GPU through the in-process communicator (khip_comm_init_local): every rank is a
# context driven by its own host thread.  Exercises the device-side halo-plan construction with real
# ghost columns, the [owned | ghost] kernels, the interior/boundary split with fused dots and the
# all-reduced reductions; only the three RCCL transport calls are replaced by device-to-device copies.
import threading


def _run_ranks(K, world, hub_id, body):
    results, errors = [None] * world, []

    def worker(rank):
        try:
            c = K.Context(0)
            c.comm_init_local(rank, world, hub_id)
            results[rank] = body(c, rank)
            c.barrier()
            c.close()
        except Exception as e:  # pragma: no cover
            import traceback
            errors.append(f"rank {rank}: {e}\n{traceback.format_exc()}")
    ts = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not errors, errors
    assert all(not t.is_alive() for t in ts), "a rank is stuck (collective mismatch)"
    return results


@pytest.mark.parametrize("world,n1", [(2, 16), (4, 16), (3, 15), (8, 16)])
def test_distributed_spmv_and_cg_local_ranks(K, oracle, world, n1):
    A_cpu = oracle.poisson3d(n1)
    n = A_cpu.n
    x = np.linspace(-1, 1, n) ** 3 + 0.25
    y_ref = A_cpu.matvec(x)
    ref = oracle.cg(A_cpu, np.ones(n), history=True)
    starts = K.row_partition(n, world)

    def body(c, rank):
        r0, r1 = starts[rank], starts[rank + 1]
        A = K.CsrMatrix.stencil(c, "poisson", n1, rows=(r0, r1), distributed=True)
        out = {}
        for overlap in (1, 0):
            c.set_option("overlap_halo", overlap)
            y = A.matvec(c.array(x[r0:r1])).to_host()
            out[f"y{overlap}"] = y
            yv = c.empty(r1 - r0)
            out[f"d{overlap}"] = K.spmv_dot(A, c.array(x[r0:r1]), yv)
        c.set_option("overlap_halo", 1)
        b = c.empty(r1 - r0)
        K.kfill_(b, 1.0)
        for fused in (2, True, False):
            xs, st, _ = K.cg(A, b, history=True, fused=fused)
            out[f"cg{int(fused)}"] = (st.niter, st.residuals, xs.to_host(), st.status)
        xs, st, _ = K.cg(A, b, history=True, variant=1)                    # single-reduction variant: one all-reduce per iteration
        out["cgv1"] = (st.niter, st.residuals, xs.to_host(), st.solved)
        return out

    res = _run_ranks(K, world, 100 + world * 10 + n1, body)
    d_ref = oracle.dot(x, y_ref)
    for rank, out in enumerate(res):
        r0, r1 = starts[rank], starts[rank + 1]
        for overlap in (1, 0):
            assert np.array_equal(out[f"y{overlap}"], y_ref[r0:r1]), (rank, overlap)        # bit-identical to the oracle
            assert abs(out[f"d{overlap}"] - d_ref) <= 4 * np.finfo(float).eps * abs(d_ref) + 1e-16 * np.abs(x * y_ref).sum()
        for fused in (2, 1, 0):
            niter, hist, xs, status = out[f"cg{fused}"]
            assert niter == ref.niter and status == ref.status
            assert np.max(np.abs(hist - ref.residuals) / ref.residuals) <= 1e-10
            assert np.allclose(xs, ref.x[r0:r1], atol=1e-10)
        # every rank computed the bit-identical scalars
        assert np.array_equal(out["cg1"][1], res[0]["cg1"][1])
        # single-reduction variant: converges to the same solution, identical scalars on every rank
        assert out["cgv1"][3] and abs(out["cgv1"][0] - ref.niter) <= 2
        assert np.allclose(out["cgv1"][2], ref.x[r0:r1], atol=1e-7 * np.abs(ref.x).max())
        assert np.array_equal(out["cgv1"][1], res[0]["cgv1"][1])
        # device-resident scalars (fused = 2): the same histories and iterates as the host-scalar loop
        assert np.array_equal(out["cg2"][1], out["cg1"][1]) and np.array_equal(out["cg2"][2], out["cg1"][2])


def test_distributed_gmres_bicgstab_local_ranks(K, oracle):
    world, n1 = 2, 10
    A_cpu = oracle.kron_unsymmetric(n1)
    n = A_cpu.n
    bh = A_cpu.matvec(np.ones(n))
    ref_g = oracle.gmres(A_cpu, bh, memory=10, restart=True, history=True)
    ref_b = oracle.bicgstab(A_cpu, bh, history=True)
    starts = K.row_partition(n, world)

    def body(c, rank):
        r0, r1 = starts[rank], starts[rank + 1]
        A = K.CsrMatrix.stencil(c, "kron_unsymmetric", n1, rows=(r0, r1), distributed=True)
        b = c.array(bh[r0:r1])
        _, stg, _ = K.gmres(A, b, memory=10, restart=True, history=True)
        _, stc, _ = K.gmres(A, b, memory=10, restart=True, history=True, variant=1)       # CGS2: three all-reduces per inner iteration
        assert stc.solved and abs(stc.niter - stg.niter) <= 1
        k = min(len(stc.residuals), len(stg.residuals))
        big = stg.residuals[:k] > 1e-6 * stg.residuals[0]
        assert np.max(np.abs(stc.residuals[:k][big] - stg.residuals[:k][big]) / stg.residuals[:k][big]) <= 1e-6
        xb1, stb, _ = K.bicgstab(A, b, history=True, fused=1)
        xb2, stb2, _ = K.bicgstab(A, b, history=True, fused=2)          # scalars and stopping tests on the device
        assert stb2.niter == stb.niter and stb2.status == stb.status
        assert np.array_equal(stb2.residuals, stb.residuals) and np.array_equal(xb2.to_host(), xb1.to_host())
        return stg.niter, stg.residuals, stb.niter, stb.residuals

    for rank, (gi, gh, bi, bhist) in enumerate(_run_ranks(K, world, 777, body)):
        assert gi == ref_g.niter and bi == ref_b.niter
        assert np.max(np.abs(gh - ref_g.residuals) / (1e-10 * ref_g.residuals + 100 * np.finfo(float).eps * ref_g.residuals[0])) <= 1.0
        assert np.max(np.abs(bhist - ref_b.residuals) / ref_b.residuals) <= 1e-7


def test_distributed_block_gmres_local_ranks(K, oracle):
    """block_gmres! on a row-partitioned operator: SpMM with an exchanged panel halo, the p x p blocks of the panel
    products summed over ranks, CholeskyQR2 on the distributed panel.  3 ranks on one GPU against the oracle's solve
    of the global system (same iteration count, residual history, solution rows)."""
    n1, p, world = 10, 4, 3
    A_cpu = oracle.kron_unsymmetric(n1)
    n = A_cpu.n
    S = A_cpu.to_scipy()
    t = (np.arange(n) + 1.0) / n
    Xt = np.stack([t ** j for j in range(p)], axis=1)
    B = S @ Xt
    ref = oracle.block_gmres(A_cpu, B, memory=8, history=True)
    starts = K.row_partition(n, world)

    def body(c, rank):
        r0, r1 = starts[rank], starts[rank + 1]
        A = K.CsrMatrix.stencil(c, "kron_unsymmetric", n1, rows=(r0, r1), distributed=True)
        # distributed SpMM against the oracle's columns
        Y = K.Panel(c, r1 - r0, p)
        K.spmm_(A, K.Panel.from_host(c, Xt[r0:r1]), Y)
        X, st, _ = K.block_gmres(A, B[r0:r1], memory=8, history=True, ctx=c)
        return dict(spmm=Y.to_host(), X=X, niter=st.niter, solved=st.solved, hist=st.residuals.copy(), status=st.status)

    res = _run_ranks(K, world, 90910, body)
    for rank, out in enumerate(res):
        r0, r1 = starts[rank], starts[rank + 1]
        assert np.array_equal(out["spmm"], B[r0:r1] if False else np.stack(
            [A_cpu.matvec(np.ascontiguousarray(Xt[:, j]))[r0:r1] for j in range(p)], axis=1))
        assert out["solved"] and out["niter"] == ref.niter and out["status"] == ref.status
        assert len(out["hist"]) == len(ref.residuals)
        assert np.max(np.abs(out["hist"] - ref.residuals) / (1e-8 * ref.residuals + 100 * np.finfo(float).eps * ref.residuals[0])) <= 1.0
        assert np.allclose(out["X"], ref.x[r0:r1], atol=1e-8 * np.abs(ref.x).max())
        assert np.array_equal(out["hist"], res[0]["hist"])            # identical on every rank


@pytest.mark.parametrize("kind", ["ill-conditioned", "dependent", "tsqr", "tsqr dependent"])
def test_distributed_panel_qr_same_factors_on_all_ranks(K, kind):
    """The panel QR on unequal row slabs (333 / 333 / 334 rows) when it leaves the plain CholeskyQR2 route: the shifted pass
    (a column 1e-9 away from another) and the deflation of dependent columns.  Every rank must apply the same p x p factors
    to its rows -- the shift depends on the row count of the WHOLE panel: R identical on the ranks, the stacked Q orthonormal,
    Q R = A."""
    world, n, p = 3, 1000, 6
    rng = np.random.default_rng(8)
    A = rng.standard_normal((n, p))
    if kind in ("ill-conditioned", "tsqr"):                   # "tsqr...": R by TSQR, one triangle per rank, finished on the host
        A[:, 5] = A[:, 0] + 1e-9 * rng.standard_normal(n)
    else:
        A = np.repeat(A, 300, axis=0)[: 300 * n]              # tall enough for the shifted pass to flatten the dependent columns
        A = A + 0.0
        A[:, 4] = A[:, 1]
        A[:, 2] = 0.0
    starts = K.row_partition(A.shape[0], world)

    def body(c, rank):
        r0, r1 = starts[rank], starts[rank + 1]
        Q = K.Panel.from_host(c, A[r0:r1])
        c.set_option("panel_qr_tsqr", 1 if kind.startswith("tsqr") else 0)
        R = K.panel_qr_(Q)
        return dict(R=R, Q=Q.to_host())

    res = _run_ranks(K, world, 31337, body)
    for out in res[1:]:
        assert np.array_equal(out["R"], res[0]["R"])
    Qs, R = np.concatenate([o["Q"] for o in res], axis=0), res[0]["R"]
    assert np.max(np.abs(Qs.T @ Qs - np.eye(p))) <= 1e-9
    assert np.max(np.abs(Qs @ R - A)) <= 1e-9 * np.sqrt(A.shape[0])


@pytest.mark.parametrize("n,p,world", [(20, 6, 4), (1000, 16, 3), (9, 4, 3)])
def test_distributed_panel_qr_is_lapacks_also_when_rank0_owns_few_rows(K, n, p, world):
    """householder!(Q, R, tau) on row slabs equals LAPACK's geqrf + orgqr without sign normalisation -- the signs and tau come
    from the top p x p block of the panel, which spans several ranks when rank 0 owns fewer than p rows (20 rows over 4 ranks,
    p = 6: rows 0-4 | 5 of rank 1)."""
    import scipy.linalg as sl
    rng = np.random.default_rng(7 * n + p)
    A = rng.standard_normal((n, p)) @ (np.eye(p) + 0.3 * rng.standard_normal((p, p)))
    starts = K.row_partition(n, world)

    def body(c, rank):
        r0, r1 = starts[rank], starts[rank + 1]
        Q = K.Panel.from_host(c, A[r0:r1])
        R, tau = K.panel_qr_tau_(Q)
        return dict(R=R, tau=tau, Q=Q.to_host())

    res = _run_ranks(K, world, 5150 + n, body)
    (_, tau_l), _ = sl.qr(A, mode="raw")
    Ql, Rl = sl.qr(A, mode="economic")
    Qs = np.concatenate([o["Q"] for o in res], axis=0)
    for o in res:
        assert np.array_equal(o["R"], res[0]["R"]) and np.array_equal(o["tau"], res[0]["tau"])
    R, tau = res[0]["R"], res[0]["tau"]
    assert np.array_equal(np.sign(np.diag(R)), np.sign(np.diag(Rl)))
    assert np.allclose(Qs, Ql, atol=1e-10) and np.allclose(R, Rl, atol=1e-10 * np.abs(Rl).max())
    assert np.allclose(tau, tau_l, atol=1e-10)


def test_distributed_block_gmres_with_dependent_right_hand_sides(K, oracle):
    """The deflating panel QR on row slabs: an equal and a zero right-hand side column, 3 ranks.  The Gram matrices are
    rank-summed, so every rank deflates the same columns; solved, the oracle's status, an iteration count within 2 of the
    oracle's global solve, identical histories on the ranks."""
    n1, world = 10, 3
    A_cpu = oracle.stencil27_unsym(n1)
    n = A_cpu.n
    S = A_cpu.to_scipy()
    B = np.random.default_rng(0).standard_normal((n, 6))
    B[:, 3] = B[:, 1]
    B[:, 5] = 0.0
    ref = oracle.block_gmres(A_cpu, B, memory=3, restart=True, rtol=1e-8, atol=0.0, history=True, itmax=200)
    starts = K.row_partition(n, world)

    def body(c, rank):
        r0, r1 = starts[rank], starts[rank + 1]
        A = K.CsrMatrix.stencil(c, "stencil27", n1, rows=(r0, r1), distributed=True)
        X, st, _ = K.block_gmres(A, B[r0:r1], memory=3, restart=True, rtol=1e-8, atol=0.0, history=True, itmax=200, ctx=c)
        return dict(X=np.asarray(X), niter=st.niter, solved=st.solved, hist=st.residuals.copy(), status=st.status)

    res = _run_ranks(K, world, 424242, body)
    X = np.concatenate([out["X"] for out in res], axis=0)
    for out in res:
        assert out["solved"] and out["status"] == ref.status and abs(out["niter"] - ref.niter) <= 2, (out["niter"], ref.niter)
        assert np.array_equal(out["hist"], res[0]["hist"])
    assert np.linalg.norm(B - S @ X, axis=0).max() <= 10.0 * np.linalg.norm(B - S @ ref.x, axis=0).max() + 1e-12


def test_distributed_krylov_processes_local_ranks(K, oracle):
    """arnoldi / hermitian_lanczos on a row-partitioned operator (3 in-process ranks): every rank gets the same H / T
    as the oracle's serial run, and the row slabs of V stack to the oracle's basis."""
    import oracle_processes as P
    world, n1, k = 3, 9, 12
    A_cpu, S_cpu = oracle.kron_unsymmetric(n1), oracle.poisson3d(n1)
    n = A_cpu.n
    bh = np.random.default_rng(3).random(n)
    Vr, beta_r, Hr = P.arnoldi(A_cpu.matvec, bh, k, reorthogonalization=True)
    Wr, gamma_r, Tr = P.hermitian_lanczos(S_cpu.matvec, bh, k)
    starts = K.row_partition(n, world)

    def body(c, rank):
        r0, r1 = starts[rank], starts[rank + 1]
        b = c.array(bh[r0:r1])
        A = K.CsrMatrix.stencil(c, "kron_unsymmetric", n1, rows=(r0, r1), distributed=True)
        V, beta, H = K.arnoldi(A, b, k, reorthogonalization=True)
        S = K.CsrMatrix.stencil(c, "poisson", n1, rows=(r0, r1), distributed=True)
        W, gamma, T = K.hermitian_lanczos(S, b, k)
        return beta, H, V.to_host(), gamma, T.data.copy(), W.to_host()

    res = _run_ranks(K, world, 424242, body)
    eps = np.finfo(float).eps
    for rank, (beta, H, Vloc, gamma, Tnz, Wloc) in enumerate(res):
        r0, r1 = starts[rank], starts[rank + 1]
        assert abs(beta - beta_r) <= 4 * eps * beta_r and abs(gamma - gamma_r) <= 4 * eps * gamma_r
        assert np.max(np.abs(H - Hr)) <= 1e-10 * np.max(np.abs(Hr))
        assert np.max(np.abs(Tnz - Tr)) <= 1e-10 * np.max(np.abs(Tr))
        assert np.max(np.abs(Vloc - Vr[r0:r1])) <= 1e-8 and np.max(np.abs(Wloc - Wr[r0:r1])) <= 1e-8
        assert np.array_equal(H, res[0][1]) and np.array_equal(Tnz, res[0][4])      # identical on every rank


@pytest.mark.parametrize("p,slices", [(16, 0), (8, 0), (32, -1), (32, 1)])
def test_distributed_spmm_window_local_ranks(K, oracle, p, slices):
    """SpMM on row slabs with ghost panel rows: the tile and window kernels (owned / ghost select, DIST instantiations; the tile
    kernel with 2, 4 and 8 lanes per row and over 16-column slices of 32-column panels) give the same bits as the
    direct-gather kernel and as the single-GPU product."""
    world, n1 = 3, 12
    A_cpu = oracle.stencil27_unsym(n1)
    n = A_cpu.n
    Xh = np.random.default_rng(2).standard_normal((n, p))
    ref = np.stack([A_cpu.matvec(np.ascontiguousarray(Xh[:, j])) for j in range(p)], axis=1)
    starts = K.row_partition(n, world)

    def body(c, rank):
        r0, r1 = starts[rank], starts[rank + 1]
        A = K.CsrMatrix.stencil(c, "stencil27", n1, rows=(r0, r1), distributed=True)
        X = K.Panel.from_host(c, Xh[r0:r1])
        out = []
        for tile, window in ((2, 1), (0, 1), (0, 0)):     # spmm_tile.hip's DIST instantiations, then the window and direct kernels
            c.set_option("spmm_tile", tile)
            c.set_option("spmm_tile_slices", slices)
            c.set_option("spmm_window", window)
            Y = K.Panel(c, r1 - r0, p)
            K.spmm_(A, X, Y)
            out.append(Y.to_host())
        out.append(A.tile_info)
        return out

    for rank, (Yt, Yw, Yd, info) in enumerate(_run_ranks(K, world, 515151, body)):
        r0, r1 = starts[rank], starts[rank + 1]
        assert info["state"] == 1, info
        assert np.array_equal(Yt, Yd) and np.array_equal(Yw, Yd) and np.array_equal(Yw, ref[r0:r1])


@pytest.mark.parametrize("world,n1", [(2, 12), (3, 10), (4, 13)])
def test_gather_mode_equals_neighbour_exchange_local_ranks(K, oracle, world, n1):
    """The two ways a distributed handle fetches the remote part of x (csrc/comm.cpp, docs/src/custom_workspaces.md:517-521
    and :583-586): neighbour exchange of the referenced entries (halo_mode = 1) and all-gather of x (halo_mode = 2, the
    general fallback BASELINE's north star names).  Same y bit for bit, same CG history, same SpMM; unequal slices
    (13^3 rows over 4 ranks) exercise the padded stride of the gather buffer; halo_mode = 0 picks the gather as soon as a
    rank would fetch more than halo_gather_pct % of its own size."""
    A_cpu = oracle.kron_unsymmetric(n1)
    n = A_cpu.n
    rng = np.random.default_rng(5)
    x = rng.standard_normal(n)
    y_ref = A_cpu.matvec(x)
    p = 4
    Xh = rng.standard_normal((n, p))
    Y_ref = np.stack([A_cpu.matvec(np.ascontiguousarray(Xh[:, j])) for j in range(p)], axis=1)
    bh = A_cpu.matvec(np.ones(n))
    ref = oracle.bicgstab(A_cpu, bh, history=True)
    starts = K.row_partition(n, world)

    def body(c, rank):
        r0, r1 = starts[rank], starts[rank + 1]
        out = {}
        for mode in (1, 2):
            c.set_option("halo_mode", mode)
            A = K.CsrMatrix.stencil(c, "kron_unsymmetric", n1, rows=(r0, r1), distributed=True)
            out[f"info{mode}"] = A.halo_info
            out[f"y{mode}"] = A.matvec(c.array(x[r0:r1])).to_host()
            Y = K.Panel(c, r1 - r0, p)
            K.spmm_(A, K.Panel.from_host(c, Xh[r0:r1]), Y)
            out[f"Y{mode}"] = Y.to_host()
            _, st, _ = K.bicgstab(A, c.array(bh[r0:r1]), history=True)
            out[f"h{mode}"] = (st.niter, st.residuals.copy())
        c.set_option("halo_mode", 0)
        c.set_option("halo_gather_pct", 1)                    # any ghost plane is more than 1 % of a slab here
        A = K.CsrMatrix.stencil(c, "kron_unsymmetric", n1, rows=(r0, r1), distributed=True)
        out["auto_low"] = A.halo_info[0]
        c.set_option("halo_gather_pct", 100000)
        A = K.CsrMatrix.stencil(c, "kron_unsymmetric", n1, rows=(r0, r1), distributed=True)
        out["auto_high"] = A.halo_info[0]
        return out

    res = _run_ranks(K, world, 31000 + world * 100 + n1, body)
    maxm = max(starts[r + 1] - starts[r] for r in range(world))
    for rank, out in enumerate(res):
        r0, r1 = starts[rank], starts[rank + 1]
        assert out["info1"][0] == 0 and out["info2"] == (1, world * maxm, r1 - r0)
        for mode in (1, 2):
            assert np.array_equal(out[f"y{mode}"], y_ref[r0:r1]), (rank, mode)
            assert np.array_equal(out[f"Y{mode}"], Y_ref[r0:r1]), (rank, mode)
            assert out[f"h{mode}"][0] == ref.niter
        assert np.array_equal(out["h1"][1], out["h2"][1])
        assert out["auto_low"] == 1 and out["auto_high"] == 0


def test_time_limit_is_agreed_on_by_all_ranks(K, oracle):
    """the wall clock is the one stopping test that is not derived from all-reduced scalars.  With a
    communicator attached it is decided collectively, so every rank leaves the loop at the same iteration with the same
    status -- for the host loops (fused 0 / 1), the device-resident loop (fused = 2), gmres! and bicgstab!."""
    world, n1 = 2, 16
    n = n1 ** 3
    starts = K.row_partition(n, world)

    def body(c, rank):
        r0, r1 = starts[rank], starts[rank + 1]
        A = K.CsrMatrix.stencil(c, "poisson", n1, rows=(r0, r1), distributed=True)
        b = c.empty(r1 - r0)
        K.kfill_(b, 1.0)
        out = []
        for fused in (2, 1, 0):
            _, st, _ = K.cg(A, b, atol=0.0, rtol=0.0, itmax=10 ** 6, timemax=1e-5, fused=fused)
            out.append((st.niter, st.status))
        _, st, _ = K.gmres(A, b, atol=0.0, rtol=0.0, itmax=10 ** 6, timemax=1e-5, memory=5, restart=True)
        out.append((st.niter, st.status))
        _, st, _ = K.bicgstab(A, b, atol=0.0, rtol=0.0, itmax=10 ** 6, timemax=1e-5)
        out.append((st.niter, st.status))
        return out

    res = _run_ranks(K, world, 60606, body)
    assert res[0] == res[1]
    assert all(status == "time limit exceeded" and niter < 1000 for niter, status in res[0])


def test_create_dist_failure_on_one_rank_fails_on_all_local_ranks(K, oracle):
    """a shard that is rejected on ONE rank before the halo plan's first collective (here: a column index outside
    [0, n), caught by the device-side validation) used to leave the other ranks blocked in the all-gather.  Now the failure
    travels in the plan's status word and every rank raises."""
    world, n1 = 3, 8
    A_cpu = oracle.poisson3d(n1)
    n = A_cpu.n
    starts = K.row_partition(n, world)

    def body(c, rank):
        r0, r1 = starts[rank], starts[rank + 1]
        sl = A_cpu.row_slice(r0, r1)
        col = sl.col.copy()
        if rank == 1:
            col[3] = n + 5                                  # invalid on this rank only
        try:
            K.CsrMatrix.from_host(c, sl.rowptr, col, sl.val, (n, n), dist_rows=(r0, r1), n_global=n)
            return "created"
        except K.KhipError as e:
            return "error: " + str(e)

    res = _run_ranks(K, world, 626262, body)
    assert all(r.startswith("error") for r in res), res
    assert "column index" in res[1] or "csr" in res[1], res[1]


@pytest.mark.parametrize("world,mode", [(3, 1), (3, 2), (2, 0), (1, 0)])
def test_distributed_transpose_local_ranks(K, oracle, world, mode):
    """khip_csr_transpose of a row-partitioned handle (the two-sided processes need mul!(y, A', x),
    src/krylov_processes.jl:133-222): A' comes back with the same partition, y = A' x is bit-identical to the serial loop over
    the columns of A (the single-GPU transpose's order), in neighbour-exchange and in gather mode; (A')' x == A x; and
    bilq-style use: the nonhermitian Lanczos process runs on (A, A') of the partitioned operator."""
    n1 = 9
    A_cpu = oracle.kron_unsymmetric(n1)
    n = A_cpu.n
    S = A_cpu.to_scipy().tocsr()
    St = S.T.tocsr()
    St.sort_indices()
    rng = np.random.default_rng(4)
    x = rng.standard_normal(n)
    yt_ref = np.zeros(n)
    for j in range(n):
        acc = 0.0
        for q in range(St.indptr[j], St.indptr[j + 1]):
            acc = acc + St.data[q] * x[St.indices[q]]
        yt_ref[j] = acc
    y_ref = A_cpu.matvec(x)
    starts = K.row_partition(n, world)

    def body(c, rank):
        r0, r1 = starts[rank], starts[rank + 1]
        c.set_option("halo_mode", mode)
        A = K.CsrMatrix.stencil(c, "kron_unsymmetric", n1, rows=(r0, r1), distributed=True)
        At = A.transpose()
        xs = c.array(x[r0:r1])
        yt = At.matvec(xs).to_host()
        Att = At.transpose()
        ytt = Att.matvec(xs).to_host()
        return dict(yt=yt, ytt=ytt, y=A.matvec(xs).to_host(), nnz=(A.nnz, At.nnz), shape=At.shape)

    res = _run_ranks(K, world, 717100 + 10 * world + mode, body)
    assert sum(r["nnz"][1] for r in res) == A_cpu.nnz
    for rank, out in enumerate(res):
        r0, r1 = starts[rank], starts[rank + 1]
        assert np.array_equal(out["yt"], yt_ref[r0:r1]), (rank, "A' x")
        assert np.array_equal(out["ytt"], y_ref[r0:r1]) and np.array_equal(out["y"], y_ref[r0:r1]), (rank, "(A')' x")


def test_pipelined_cg_on_partitioned_operators(K, ctx, dctx, oracle):
    """options.variant = 2 (pipelined CG) on row-partitioned operators: with a real (one-rank) RCCL communicator the
    reduction's all-gather, the cross-rank combine and the scalar epilogue run on the COMMUNICATION stream while the next
    product runs on the main stream (comm_allreduce_dd_device_begin / _end); with three in-process ranks the in-order form.
    Same iteration count as the plain single-GPU run of the variant, histories within 1e-10 of it, within the variant's
    budget of the oracle's cg!."""
    n1 = 20
    A_cpu = oracle.poisson3d(n1)
    n = A_cpu.n
    ref = oracle.cg(A_cpu, np.ones(n), history=True)
    Ap = K.CsrMatrix.stencil(ctx, "poisson", n1)
    bp = ctx.empty(n); K.kfill_(bp, 1.0)
    xp, stp, _ = K.cg(Ap, bp, history=True, variant=2)
    assert stp.solved and abs(stp.niter - ref.niter) <= 3
    Ad = K.CsrMatrix.stencil(dctx, "poisson", n1, rows=(0, n), distributed=True)
    bd = dctx.empty(n); K.kfill_(bd, 1.0)
    xd, std_, _ = K.cg(Ad, bd, history=True, variant=2)
    assert std_.niter == stp.niter and np.max(np.abs(std_.residuals - stp.residuals) / stp.residuals) <= 1e-10
    assert np.allclose(xd.to_host(), ref.x, rtol=0, atol=1e-6 * np.abs(ref.x).max())
    dctx.barrier()
    world = 3
    starts = K.row_partition(n, world)

    def body(c, rank):
        r0, r1 = starts[rank], starts[rank + 1]
        A = K.CsrMatrix.stencil(c, "poisson", n1, rows=(r0, r1), distributed=True)
        b = c.empty(r1 - r0); K.kfill_(b, 1.0)
        x, st, _ = K.cg(A, b, history=True, variant=2)
        return st.niter, st.residuals.copy(), x.to_host(), st.solved

    res = _run_ranks(K, world, 828282, body)
    for rank, (niter, hist, xs, solved) in enumerate(res):
        r0, r1 = starts[rank], starts[rank + 1]
        assert solved and niter == stp.niter and np.max(np.abs(hist - stp.residuals) / stp.residuals) <= 1e-10
        assert np.array_equal(hist, res[0][1])
        assert np.allclose(xs, ref.x[r0:r1], rtol=0, atol=1e-6 * np.abs(ref.x).max())


def test_sstep_gmres_on_partitioned_operators(K, ctx, oracle):
    """gmres! variant 2 (s-step) on three in-process ranks: the batched Gram-Schmidt and Gram reductions go through the device
    all-reduce 64 scalars at a time; same iteration count and history (1e-9) as the single-GPU run of the variant, identical on
    every rank."""
    n1, memory = 12, 12
    A_cpu = oracle.kron_unsymmetric(n1)
    n = A_cpu.n
    bh = A_cpu.matvec(np.ones(n))
    Ap = K.CsrMatrix.stencil(ctx, "kron_unsymmetric", n1)
    _, stp, _ = K.gmres(Ap, ctx.array(bh), memory=memory, restart=True, history=True, variant=2)
    assert stp.solved
    world = 3
    starts = K.row_partition(n, world)

    def body(c, rank):
        r0, r1 = starts[rank], starts[rank + 1]
        A = K.CsrMatrix.stencil(c, "kron_unsymmetric", n1, rows=(r0, r1), distributed=True)
        x, st, _ = K.gmres(A, c.array(bh[r0:r1]), memory=memory, restart=True, history=True, variant=2)
        return st.niter, st.residuals.copy(), x.to_host(), st.solved

    res = _run_ranks(K, world, 939393, body)
    for rank, (niter, hist, xs, solved) in enumerate(res):
        assert solved and niter == stp.niter and np.max(np.abs(hist - stp.residuals) / stp.residuals) <= 1e-9
        assert np.array_equal(hist, res[0][1])
        assert np.allclose(xs, 1.0, atol=1e-6)


@pytest.mark.parametrize("kernel", [2, 3])
def test_device_loops_with_the_two_reduction_fallback_on_partitioned_operators(K, ctx, oracle, kernel):
    """with a forced ordered / vector SpMV kernel (spmv_kernel = 2 / 3) the product cannot carry the second
    reduction, so spmv_any issues two reductions and a stand-alone scalar epilogue -- which under a communicator belongs to
    the cross-rank combine alone (it was applied twice: iter advanced by two, alpha and gamma corrupted, ranks could diverge).
    Single-reduction CG (variant 1) and bicgstab!(fused = 2) on three in-process ranks against the same runs on one GPU."""
    n1 = 14
    A_cpu = oracle.poisson3d(n1)
    n = A_cpu.n
    ref = oracle.cg(A_cpu, np.ones(n), history=True)
    B_cpu = oracle.kron_unsymmetric(n1)
    bb = B_cpu.matvec(np.ones(n))
    refb = oracle.bicgstab(B_cpu, bb, history=True)
    prev = ctx.get_option("spmv_kernel")
    ctx.set_option("spmv_kernel", kernel)
    try:
        Ap = K.CsrMatrix.stencil(ctx, "poisson", n1)
        bp = ctx.empty(n); K.kfill_(bp, 1.0)
        _, stp, _ = K.cg(Ap, bp, history=True, variant=1)
        Bp = K.CsrMatrix.stencil(ctx, "kron_unsymmetric", n1)
        _, stb, _ = K.bicgstab(Bp, ctx.array(bb), history=True, fused=2)
    finally:
        ctx.set_option("spmv_kernel", prev)
    assert stp.solved and abs(stp.niter - ref.niter) <= 2 and stb.solved and stb.niter == refb.niter
    world = 3
    starts = K.row_partition(n, world)

    def body(c, rank):
        c.set_option("spmv_kernel", kernel)
        r0, r1 = starts[rank], starts[rank + 1]
        A = K.CsrMatrix.stencil(c, "poisson", n1, rows=(r0, r1), distributed=True)
        b = c.empty(r1 - r0); K.kfill_(b, 1.0)
        x, st, _ = K.cg(A, b, history=True, variant=1)
        B = K.CsrMatrix.stencil(c, "kron_unsymmetric", n1, rows=(r0, r1), distributed=True)
        xb, sb, _ = K.bicgstab(B, c.array(bb[r0:r1]), history=True, fused=2)
        return st.niter, st.residuals.copy(), x.to_host(), st.solved, sb.niter, sb.residuals.copy(), xb.to_host(), sb.solved

    res = _run_ranks(K, world, 929200 + kernel, body)
    for rank, (niter, hist, xs, solved, nb, hb, xb, sb) in enumerate(res):
        r0, r1 = starts[rank], starts[rank + 1]
        assert solved and niter == stp.niter and len(hist) == len(stp.residuals)
        assert np.max(np.abs(hist - stp.residuals) / stp.residuals) <= 1e-10
        assert np.array_equal(hist, res[0][1])
        assert np.allclose(xs, ref.x[r0:r1], rtol=0, atol=1e-6 * np.abs(ref.x).max())
        assert sb and nb == stb.niter and np.max(np.abs(hb - stb.residuals) / stb.residuals) <= 1e-7
        assert np.array_equal(hb, res[0][5])
        assert np.allclose(xb, refb.x[r0:r1], rtol=0, atol=1e-6 * np.abs(refb.x).max())
