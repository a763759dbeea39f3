"""GPU parity of the panel kernels (MFMA f64) and block_gmres_ against numpy / the CPU oracle.

The panel QR on the device is CholeskyQR2 with LAPACK's Householder signs and tau recovered from the top p x p block
(csrc/block.cpp): Q, R and tau are compared with LAPACK's geqrf / orgqr directly, no sign fix-up.  block-GMRES parity is
stated on iteration counts, residual norm histories and the solution.  Tolerance on the history: |dr_k| <= 1e-8 r_k + floor * r_0 with floor = 100 eps
without restart (measured: 1e-13 relative) and floor = 1e-10 with restart: the restart recomputes B - A X,
and X carries the cond(R_k)-amplified difference between the two panel-QR variants (measured 1.3e-11 r_0)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
EPS = np.finfo(float).eps


@pytest.mark.parametrize("n,p", [(1, 1), (16, 16), (1000, 16), (4097, 3), (33333, 16), (2048, 32), (5000, 20)])
def test_panel_layout_and_products(K, ctx, n, p):
    rng = np.random.default_rng(n + p)
    V, Q = rng.standard_normal((n, p)), rng.standard_normal((n, p))
    dV, dQ = K.Panel.from_host(ctx, V), K.Panel.from_host(ctx, Q)
    assert np.array_equal(dV.to_host(), V)                              # layout round trip is exact
    Psi = K.panel_gemm_tn(dV, dQ)
    ref = V.T @ Q
    scale = np.abs(V).T @ np.abs(Q)
    assert np.all(np.abs(Psi - ref) <= 4 * np.sqrt(n) * EPS * scale + 1e-300)
    M = rng.standard_normal((p, p))
    K.panel_gemm_nn_(-1.0, dV, M, 1.0, dQ)                               # Q -= V M
    ref2 = Q - V @ M
    assert np.allclose(dQ.to_host(), ref2, rtol=0, atol=64 * EPS * (np.abs(Q) + np.abs(V) @ np.abs(M)).max())
    K.panel_gemm_nn_(2.0, dV, M, 0.0, dQ)                                # beta = 0 ignores the old Q
    assert np.allclose(dQ.to_host(), 2.0 * V @ M, rtol=0, atol=64 * EPS * (np.abs(V) @ np.abs(M)).max())
    assert abs(K.panel_norm(dV) - np.linalg.norm(V)) <= 4 * EPS * np.linalg.norm(V)
    # MFMA operand check with an ASYMMETRIC second factor (catches row/column swaps)
    E = np.zeros((n, p)); E[: min(n, p), : min(n, p)] = np.eye(min(n, p))
    dE = K.Panel.from_host(ctx, E)
    assert np.allclose(K.panel_gemm_tn(dE, dQ), (2.0 * V @ M)[:p, :][: min(n, p)].T.T if False else E.T @ (2.0 * V @ M), atol=1e-12)


@pytest.mark.parametrize("n,p", [(64, 4), (1000, 16), (20000, 16), (3000, 7), (4096, 32), (16, 16), (33, 1)])
def test_panel_qr(K, ctx, n, p):
    """householder!(Q, R, tau) (src/block_krylov_utils.jl:201-208): Q, R and tau equal LAPACK's geqrf + orgqr -- compared
    WITHOUT any sign normalisation."""
    import scipy.linalg as sl
    rng = np.random.default_rng(7 * n + p)
    A = rng.standard_normal((n, p)) @ (np.eye(p) + 0.3 * rng.standard_normal((p, p)))
    dQ = K.Panel.from_host(ctx, A)
    R, tau = K.panel_qr_tau_(dQ)
    Qh = dQ.to_host()
    assert np.allclose(np.tril(R, -1), 0)
    assert np.allclose(Qh.T @ Qh, np.eye(p), atol=1e-13)
    assert np.allclose(Qh @ R, A, atol=1e-12 * np.abs(A).max() * p)
    (_, tau_l), _ = sl.qr(A, mode="raw")
    Ql, Rl = sl.qr(A, mode="economic")
    assert np.array_equal(np.sign(np.diag(R)), np.sign(np.diag(Rl)))
    assert np.allclose(Qh, Ql, atol=1e-10) and np.allclose(R, Rl, atol=1e-10 * np.abs(Rl).max())
    assert np.allclose(tau, tau_l, atol=1e-10) and np.all(((tau >= 1.0) & (tau <= 2.0)) | (tau == 0.0))   # tau = 0: last column of a square block
    # the positive-diagonal factor stays available
    ctx.set_option("panel_signs", 0)
    try:
        dQ2 = K.Panel.from_host(ctx, A)
        R2 = K.panel_qr_(dQ2)
        S = np.sign(np.diag(Rl))
        assert np.all(np.diag(R2) > 0) and np.allclose(dQ2.to_host(), Ql * S, atol=1e-10)
    finally:
        ctx.set_option("panel_signs", 1)


@pytest.mark.parametrize("n,p,cond", [(5000, 16, 1e2), (20000, 16, 1e10), (20000, 16, 1e14), (3000, 7, 1e12), (4096, 32, 1e9), (300, 16, 1e6)])
def test_panel_qr_by_tsqr(K, ctx, n, p, cond):
    """option panel_qr_tsqr = 1 (TSQR): the R factor comes from block Householder QRs reduced over a tree of
    triangles instead of the Cholesky of the Gram matrix -- no conditioning limit, no shifted pass.  Q orthonormal to 1e-13 and
    Q R = A to 1e-13 ||A|| up to cond 1e14 (where CholeskyQR2 needs its shifted pass or gives up); R, tau, Q equal LAPACK's to
    cond * eps; on well-conditioned panels also equal to the default path's."""
    import scipy.linalg as sl
    rng = np.random.default_rng(n + p)
    U = np.linalg.qr(rng.standard_normal((n, p)))[0]
    W = np.linalg.qr(rng.standard_normal((p, p)))[0]
    A = (U * np.logspace(0, -np.log10(cond), p)) @ W.T
    ctx.set_option("panel_qr_tsqr", 1)
    try:
        dQ = K.Panel.from_host(ctx, A)
        R, tau = K.panel_qr_tau_(dQ)
        Qh = dQ.to_host()
    finally:
        ctx.set_option("panel_qr_tsqr", 0)
    assert np.allclose(np.tril(R, -1), 0)
    assert np.abs(Qh.T @ Qh - np.eye(p)).max() <= 1e-13
    assert np.abs(Qh @ R - A).max() <= 1e-13 * np.abs(A).max() * p
    (_, tau_l), _ = sl.qr(A, mode="raw")
    Ql, Rl = sl.qr(A, mode="economic")
    tol = max(1e-12, 50 * cond * np.finfo(float).eps)
    assert np.array_equal(np.sign(np.diag(R)), np.sign(np.diag(Rl)))
    assert np.allclose(R, Rl, atol=tol * np.abs(Rl).max())
    if cond <= 1e10:
        assert np.allclose(Qh, Ql, atol=tol) and np.allclose(tau, tau_l, atol=tol)
    if cond <= 1e6:
        dQ2 = K.Panel.from_host(ctx, A)
        R2, tau2 = K.panel_qr_tau_(dQ2)
        assert np.allclose(R2, R, atol=1e-9 * np.abs(R).max()) and np.allclose(dQ2.to_host(), Qh, atol=1e-9)


def test_block_gmres_with_tsqr_panels_same_histories(K, ctx, oracle):
    A = oracle.stencil27_unsym(12)
    S = A.to_scipy()
    B, _ = _rhs(S, A.n, 16)
    dA = K.CsrMatrix.from_host(ctx, A.rowptr, A.col, A.val, (A.n, A.n))
    res = []
    for tsqr in (0, 1):
        ctx.set_option("panel_qr_tsqr", tsqr)
        X, st, _ = K.block_gmres(dA, B, memory=5, ctx=ctx, history=True, restart=True, itmax=20)
        res.append((X, st.niter, np.array(st.residuals)))
    ctx.set_option("panel_qr_tsqr", 0)
    assert res[0][1] == res[1][1]
    # two correct QRs round differently; restarted block-GMRES at toy sizes amplifies that (the oracle's own
    # double-precision history is 1e-3 from the exact one on such a case): first cycle to 1e-11, the whole history to 1e-4
    dev = np.abs(res[0][2] - res[1][2]) / res[0][2]
    assert dev[:6].max() <= 1e-11 and dev.max() <= 1e-4, dev
    assert np.allclose(res[0][0], res[1][0], atol=1e-5 * np.abs(res[0][0]).max())


def test_panel_qr_unit_columns_take_dlarfg_tau_zero(K, ctx):
    """a reduced column that is exactly +-e_j (identity columns, an already triangular panel) has a zero
    sub-column: DLARFG returns tau = 0 and keeps the sign of the pivot, and so does the panel QR -- same R and tau as geqrf."""
    import scipy.linalg as sl
    n, p = 500, 8
    for A in (np.eye(n, p), np.vstack([np.triu(np.random.default_rng(1).standard_normal((p, p))) + 3 * np.diag([1, -1, 1, 1, -1, 1, -1, 1]),
                                        np.zeros((n - p, p))])):
        dQ = K.Panel.from_host(ctx, A)
        R, tau = K.panel_qr_tau_(dQ)
        Qh = dQ.to_host()
        (_, tau_l), _ = sl.qr(A, mode="raw")
        Ql, Rl = sl.qr(A, mode="economic")
        assert np.all(tau_l == 0.0) and np.all(tau == 0.0), (tau, tau_l)
        assert np.allclose(R, Rl, atol=1e-13 * np.abs(Rl).max()) and np.allclose(Qh, Ql, atol=1e-13)
        assert np.allclose(Qh @ R, A, atol=1e-13 * np.abs(A).max() * p)


@pytest.mark.parametrize("eps_col", [1e-6, 1e-9, 1e-12])
def test_panel_qr_ill_conditioned_takes_the_shifted_pass(K, ctx, eps_col):
    """cond(A) up to ~1e12: CholeskyQR2 alone is unsafe (cond^2 > 1/eps); the shifted first pass (shifted
    CholeskyQR3) keeps the whole factorisation on the device and still delivers an orthonormal Q and A = Q R."""
    rng = np.random.default_rng(3)
    A = rng.standard_normal((500, 6))
    A[:, 5] = A[:, 0] + eps_col * rng.standard_normal(500)
    dQ = K.Panel.from_host(ctx, A)
    R = K.panel_qr_(dQ)
    Qh = dQ.to_host()
    assert np.allclose(np.tril(R, -1), 0) and np.all(np.diag(R) != 0)
    assert np.allclose(Qh.T @ Qh, np.eye(6), atol=1e-10)
    assert np.allclose(Qh @ R, A, atol=1e-10)


def test_panel_qr_rank_deficient_block(K, ctx):
    """An exactly dependent column behaves as with LAPACK's Householder QR: A = Q R still holds, the dependent
    direction shows up as a (relatively) tiny diagonal entry of R and an arbitrary unit column of Q (here the shifted
    pass alone copes; taller panels: test_panel_qr_dependent_columns_get_stand_in_directions).  A zero block is an error."""
    rng = np.random.default_rng(4)
    A = rng.standard_normal((300, 4))
    A[:, 3] = 2.0 * A[:, 1]
    dQ = K.Panel.from_host(ctx, A)
    R = K.panel_qr_(dQ)
    Qh = dQ.to_host()
    assert np.allclose(Qh @ R, A, atol=1e-9) and np.allclose(Qh.T @ Qh, np.eye(4), atol=1e-8)
    assert abs(R[3, 3]) <= 1e-5 * abs(R[1, 1])
    with pytest.raises(K.KhipError):
        K.panel_qr_(K.Panel.from_host(ctx, np.zeros((64, 3))))


@pytest.mark.parametrize("tsqr", [0, 1])
def test_panel_qr_dependent_columns_get_stand_in_directions(K, ctx, tsqr):
    """Tall panels whose dependent columns survive even the shifted pass (its shift grows with n): an equal column, a linear
    combination, a zero column.  As with LAPACK's Householder QR (src/block_gmres.jl:250-283 calls householder! on whatever
    block it gets) the factorisation goes on: A = Q R, Q orthonormal, R upper triangular with a zero where the dependent
    column is, and a unit vector in Q there."""
    rng = np.random.default_rng(41)
    n, p = 200_000, 8
    A = rng.standard_normal((n, p))
    A[:, 3] = A[:, 1]
    A[:, 5] = 2.0 * A[:, 0] - 0.5 * A[:, 2]
    A[:, 6] = 0.0
    ctx.set_option("panel_qr_tsqr", tsqr)
    try:
        dQ = K.Panel.from_host(ctx, A)
        R = K.panel_qr_(dQ)
        Qh = dQ.to_host()
    finally:
        ctx.set_option("panel_qr_tsqr", 0)
    assert np.allclose(np.tril(R, -1), 0)
    assert np.max(np.abs(Qh.T @ Qh - np.eye(p))) <= 1e-10
    assert np.max(np.abs(Qh @ R - A)) <= 1e-9 * np.sqrt(n)
    d = np.abs(np.diag(R))
    assert np.all(d[[3, 5, 6]] <= 1e-6 * d[0]) and np.all(d[[0, 1, 2, 4, 7]] >= 1e-3 * d[0])


@pytest.mark.parametrize("case", ["equal columns", "linear combination", "zero column", "equal + zero, memory 3"])
def test_block_gmres_with_dependent_right_hand_sides(K, ctx, oracle, case):
    """block_gmres! on right-hand sides without full column rank: the reference's Householder QR of the residual block does not
    care and the solve converges (the oracle, which restates it, takes 20-24 iterations here).  Same on the device: solved, the
    same status, an iteration count within 2 of the oracle's, true residuals as small as the oracle's."""
    A = oracle.stencil27_unsym(10)
    S = A.to_scipy()
    n = A.n
    rng = np.random.default_rng(0)
    B = rng.standard_normal((n, 6))
    memory = 5
    if case == "equal columns":
        B[:, 3] = B[:, 1]
    elif case == "linear combination":
        B[:, 5] = 2.0 * B[:, 0] - B[:, 2]
    elif case == "zero column":
        B[:, 4] = 0.0
    else:
        B[:, 3] = B[:, 1]
        B[:, 5] = 0.0
        memory = 3
    ref = oracle.block_gmres(A, B, memory=memory, restart=True, rtol=1e-8, atol=0.0, history=True, itmax=200)
    dA = K.CsrMatrix.from_host(ctx, A.rowptr, A.col, A.val, (n, n))
    X, st, ws = K.block_gmres(dA, B, memory=memory, restart=True, rtol=1e-8, atol=0.0, history=True, itmax=200)
    Xh = np.asarray(X)
    assert st.solved and ref.solved and st.status == ref.status
    assert abs(st.niter - ref.niter) <= 2, (st.niter, ref.niter)
    res_gpu = np.linalg.norm(B - S @ Xh, axis=0).max()
    res_ref = np.linalg.norm(B - S @ ref.x, axis=0).max()
    assert res_gpu <= 10.0 * res_ref + 1e-12, (res_gpu, res_ref)
    assert np.max(np.abs(Xh - ref.x)) <= 1e-5 * np.max(np.abs(ref.x))


def test_spmm_panel(K, ctx, oracle):
    A = oracle.stencil27_unsym(10)
    rng = np.random.default_rng(5)
    X = rng.standard_normal((A.n, 16))
    dA = K.CsrMatrix.from_host(ctx, A.rowptr, A.col, A.val, (A.n, A.n))
    dX, dY = K.Panel.from_host(ctx, X), K.Panel(ctx, A.n, 16)
    K.spmm_(dA, dX, dY)
    ref = np.stack([A.matvec(np.ascontiguousarray(X[:, j])) for j in range(16)], axis=1)
    assert np.array_equal(dY.to_host(), ref)


def _rhs(S, n, p):
    """B = A * X_true as in interfaces/test/C/test_block.c:62-70.  That file uses X_true[i, j] = ((i+1)/n)^j with
    p = 3; for p = 16 those monomials are numerically rank deficient (cond ~ 1e16, outside the reference's
    contract "B must have full column rank", docs/src/interfaces/reference.md:236), so wider blocks use a
    well-conditioned trigonometric family instead."""
    t = (np.arange(n) + 1.0) / n
    if p <= 4:
        Xt = np.stack([t ** j for j in range(p)], axis=1)
    else:
        Xt = np.stack([np.cos(j * np.pi * t) + 0.1 * j for j in range(p)], axis=1)
    return S @ Xt, Xt


@pytest.mark.parametrize("n1,p,kw", [(6, 3, {}), (8, 4, {}), (8, 4, dict(restart=True)), (8, 4, dict(reorthogonalization=True)),
                                     (10, 16, dict(restart=True)), (12, 16, {})])
def test_block_gmres_matches_oracle(K, ctx, oracle, parity_log, n1, p, kw):
    A = oracle.kron_unsymmetric(n1)
    S = A.to_scipy()
    B, Xt = _rhs(S, A.n, p)
    ref = oracle.block_gmres(A, B, memory=8, history=True, **kw)
    dA = K.CsrMatrix.from_host(ctx, A.rowptr, A.col, A.val, (A.n, A.n))
    X, st, ws = K.block_gmres(dA, B, memory=8, history=True, **kw)
    assert st.solved and st.status == ref.status and st.niter == ref.niter
    h, hr = st.residuals, ref.residuals
    assert len(h) == len(hr)
    floor = 1e-10 if kw.get("restart") else 100 * EPS
    dev = float(np.max(np.abs(h - hr) / (1e-8 * hr + floor * hr[0])))
    parity_log(test="block_gmres", n1=n1, p=p, kw=kw, niter=st.niter, hist_tol_units=dev,
               hist_max_rel=float(np.max(np.abs(h - hr) / hr)), x_max_abs=float(np.abs(X - ref.x).max()))
    assert dev <= 1.0
    assert np.linalg.norm(B - S @ X) / np.linalg.norm(B) <= 1e-6            # interfaces/test/C/test_block.c bound
    assert np.abs(X - Xt).max() <= 1e-5


def test_block_gmres_warm_start_and_itmax(K, ctx, oracle):
    A = oracle.kron_unsymmetric(6)
    S = A.to_scipy()
    B, Xt = _rhs(S, A.n, 3)
    dA = K.CsrMatrix.from_host(ctx, A.rowptr, A.col, A.val, (A.n, A.n))
    X, st, _ = K.block_gmres(dA, B, memory=4, itmax=2)
    assert st.niter == 2 and st.status == "maximum number of iterations exceeded"
    X0 = 0.9 * Xt
    ref = oracle.block_gmres(A, B, X0=X0, memory=8)
    X, st, _ = K.block_gmres(dA, B, X0=X0, memory=8)
    assert st.solved and st.niter == ref.niter and np.abs(X - ref.x).max() <= 1e-8


def test_block_gmres_preconditioners(K, ctx, oracle):
    """Left / right Jacobi as device callbacks on row-major panels (interfaces/test/C/test_block.c Jacobi cases);
    kron_unsymmetric has the constant diagonal 12."""
    A = oracle.kron_unsymmetric(7)
    S = A.to_scipy()
    B, Xt = _rhs(S, A.n, 4)
    dA = K.CsrMatrix.from_host(ctx, A.rowptr, A.col, A.val, (A.n, A.n))

    def jac(X, Y):
        K.kdivcopy_(len(X.buf), Y.buf, X.buf, 12.0)
    for side in ("M", "N"):
        ref = oracle.block_gmres(A, B, memory=8, history=True, **{side: (lambda X: X / 12.0)})
        X, st, _ = K.block_gmres(dA, B, memory=8, history=True, **{side: jac})
        assert st.solved and st.niter == ref.niter
        assert np.max(np.abs(st.residuals - ref.residuals) / (1e-8 * ref.residuals + 100 * EPS * ref.residuals[0])) <= 1.0
        assert np.abs(X - Xt).max() <= 1e-5
    # operator given as a callback instead of a CSR handle
    X, st, _ = K.block_gmres(lambda Xp, Yp: K.spmm_(dA, Xp, Yp), B, memory=8, ctx=ctx)
    assert st.solved and np.abs(X - Xt).max() <= 1e-5


# ---- SpMM with the panel-row window in LDS (csr_aux.hip, spmm_window_kernel) ------------------------------------

def _spmm_both(K, ctx, dA, X, p):
    """Y with the window kernel (default) and with the direct-gather kernel, as host arrays."""
    dX = K.Panel.from_host(ctx, X)
    out = []
    for window in (1, 0):
        ctx.set_option("spmm_window", window)
        dY = K.Panel(ctx, dA.m, p)
        K.spmm_(dA, dX, dY)
        out.append(dY.to_host())
    ctx.set_option("spmm_window", 1)
    return out


@pytest.mark.parametrize("p", [2, 4, 6, 8, 12, 16, 24, 32])
def test_spmm_window_bit_identical_to_columnwise_spmv(K, ctx, oracle, p):
    """27-point operator (the rows of a group share panel rows: window path) -- Y == the direct-gather kernel == p SpMVs."""
    A = oracle.stencil27_unsym(14)
    X = np.random.default_rng(p).standard_normal((A.n, p))
    dA = K.CsrMatrix.from_host(ctx, A.rowptr, A.col, A.val, (A.n, A.n))
    Yw, Yd = _spmm_both(K, ctx, dA, X, p)
    ref = np.stack([A.matvec(np.ascontiguousarray(X[:, j])) for j in range(p)], axis=1)
    assert np.array_equal(Yw, Yd) and np.array_equal(Yw, ref)


def test_spmm_window_mixed_groups_and_fallbacks(K, ctx, oracle):
    """(a) a banded operator with a few dense rows: their groups exceed the window and take the direct path inside the
    window kernel; (b) a scattered operator: no locality, the handle keeps the direct-gather kernel; (c) rectangular;
    (d) rows of very different length in one wave (non-uniform path) incl. empty rows."""
    import scipy.sparse as sp
    rng = np.random.default_rng(3)
    n, p = 6000, 16

    def check(S, tag):
        S = S.tocsr()
        S.sort_indices()
        dA = K.CsrMatrix.from_host(ctx, S.indptr.astype(np.int64), S.indices.astype(np.int32), S.data.copy(), S.shape)
        X = rng.standard_normal((S.shape[1], p))
        Yw, Yd = _spmm_both(K, ctx, dA, X, p)
        ref = np.zeros((S.shape[0], p))
        for i in range(S.shape[0]):                      # serial row loops, column by column = the order of the kernels
            for q in range(S.indptr[i], S.indptr[i + 1]):
                ref[i] = ref[i] + S.data[q] * X[S.indices[q]]
        assert np.array_equal(Yw, Yd), tag
        assert np.array_equal(Yw, ref), tag

    band = sp.diags([rng.standard_normal(n - abs(k)) for k in range(-6, 7)], list(range(-6, 7)), format="lil")
    for r in (100, 101, 3333):                           # dense rows: > window capacity in panel rows and in nonzeros
        band[r, :] = rng.standard_normal(n) * (rng.random(n) < 0.3)
    check(band, "band + dense rows")
    check(sp.random(n, n, density=0.004, random_state=7) + sp.eye(n), "scattered")
    check(sp.diags([np.ones(900), 2 * np.ones(900), 3 * np.ones(900)], [0, 40, 1], shape=(900, 1400)), "rectangular")
    ragged = sp.lil_matrix((n, n))
    for i in range(n):
        k = (i * 7) % 23                                 # 0..22 entries, changing from row to row
        cols = np.unique(np.clip(i + np.arange(k) - k // 2, 0, n - 1))
        if cols.size:
            ragged[i, cols] = rng.standard_normal(cols.size)
    check(ragged, "ragged rows")


# ---- SpMM with 16 columns: wave-private windows filled by LDS-DMA, grid-tile row groups (spmm_tile.hip) ----------

def _spmm_three(K, ctx, dA, X, slices=0):
    """Y (8, 16, 32, 64 columns) with the tile kernel, the window kernel and the direct-gather kernel, as host arrays.
    slices: panels of 32 columns and more as 16-column slices (1), in one launch (-1), as the library decides (0)."""
    p = X.shape[1]
    if p <= 32:
        dX = K.Panel.from_host(ctx, X)
    else:                                                    # wider than the panel kernels go: the row-major buffer directly
        dX = K.Panel(ctx, X.shape[0], p)
        h = np.zeros((K.panel_rows(X.shape[0]), p))
        h[:X.shape[0]] = X
        dX.buf.copy_from_host(h.ravel())
    out = []
    for tile, window in ((2, 1), (0, 1), (0, 0)):            # 2: the tile kernel at every width it has
        ctx.set_option("spmm_tile", tile)
        ctx.set_option("spmm_tile_slices", slices)
        ctx.set_option("spmm_window", window)
        dY = K.Panel(ctx, dA.m, p)
        K.spmm_(dA, dX, dY)
        out.append(dY.to_host() if p <= 32 else dY.buf.to_host().reshape(-1, p)[:dA.m].copy())
    ctx.set_option("spmm_tile", 1)
    ctx.set_option("spmm_tile_slices", 0)
    ctx.set_option("spmm_window", 1)
    return out


def _serial_spmm(S, X):
    ref = np.zeros((S.shape[0], X.shape[1]))
    for i in range(S.shape[0]):                          # serial row loops, column by column = the order of the kernels
        for q in range(S.indptr[i], S.indptr[i + 1]):
            ref[i] = ref[i] + S.data[q] * X[S.indices[q]]
    return ref


@pytest.mark.parametrize("kind,dims", [("stencil27", (14, 14, 14)), ("stencil27", (21, 13, 10)), ("poisson", (33, 33, 33)),
                                       ("kron_unsymmetric", (18, 18, 18)), ("poisson", (40, 40, 1))])
def test_spmm_tile_grid_operators_bit_identical(K, ctx, oracle, kind, dims):
    """Structured-grid operators: the 32-row groups are grid tiles (tile_info.grid_tiles), extents that are no multiple of
    the tile (partial tiles), a single plane (8 x 4 x 1 tiles).  Y == window kernel == direct kernel == serial loop."""
    n1, n2, n3 = dims
    dA = K.CsrMatrix.stencil(ctx, kind, n1, n2, n3)
    X = np.random.default_rng(n1).standard_normal((dA.n, 16))
    Yt, Yw, Yd = _spmm_three(K, ctx, dA, X)
    info = dA.tile_info
    assert info["state"] == 1 and info["grid_tiles"] == 1 and info["direct_groups"] == 0, info
    assert np.array_equal(Yt, Yd) and np.array_equal(Yw, Yd)
    A = None
    if kind == "stencil27" and n1 == n2 == n3:
        A = oracle.stencil27_unsym(n1)
    elif kind == "poisson":
        A = oracle.poisson3d(n1, n2, n3)
    elif kind == "kron_unsymmetric":
        A = oracle.kron_unsymmetric(n1)
    if A is not None:
        ref = np.stack([A.matvec(np.ascontiguousarray(X[:, j])) for j in range(16)], axis=1)
        assert np.array_equal(Yt, ref)
    if kind == "poisson" and n3 > 1:
        # 7-point tile: far fewer than the 6 x 6 x 4 box; half a window more where the sliding windows look one group ahead
        assert info["window"] <= (6 * 4 * 4 + 8) * 3 // 2, info


@pytest.mark.parametrize("p", [8, 16, 32])
def test_spmm_tile_sliding_and_double_buffered_windows_bit_identical(K, ctx, oracle, p):
    """Round 4's two window schemes of the tile kernel, forced on and off: sliding windows (spmm_tile_slide: a wave walks a run of
    groups and copies only the panel rows its window lacks; run lengths 1 = whole grid lines, 3, 7; grids with partial tiles, a
    single plane, and an operator without a grid) and double-buffered windows (spmm_tile_dbuf: the copies of the next group
    land under this group's products).  Y equals the direct-gather kernel's bit for bit in every combination."""
    saved = {k: ctx.get_option(k) for k in ("spmm_tile", "spmm_tile_slide", "spmm_tile_dbuf", "spmm_window", "spmm_tile_shape", "spmm_tile_pair", "spmm_tile_ahead")}
    rng = np.random.default_rng(31 + p)
    makers = [lambda: K.CsrMatrix.stencil(ctx, "stencil27", 13, 10, 9), lambda: K.CsrMatrix.stencil(ctx, "poisson", 21, 12, 11),
              lambda: K.CsrMatrix.stencil(ctx, "poisson", 37, 29, 1), lambda: K.CsrMatrix.banded_random(ctx, 6000, seed=3)]
    try:
        for make in makers:
            ctx.set_option("spmm_tile", 0); ctx.set_option("spmm_window", 0)
            dA = make()
            X = rng.standard_normal((dA.n, p))
            dX = K.Panel.from_host(ctx, X)
            dY = K.Panel(ctx, dA.m, p)
            K.spmm_(dA, dX, dY)
            ref = dY.to_host()
            ctx.set_option("spmm_tile", 2)
            # (slide, dbuf, shape, pair): pair = two waves per window (spmm_tile2_kernel, p >= 16; with and without sliding windows);
            # the last rows are the defaults (-1: by rule)
            # ahead (round 6): the sliding windows' look-ahead -- the next group's new panel rows are copied BEFORE this group's
            # products, into slots neither group reads (two-wave kernel with runs; 0 = the one-window loop of round 4)
            for slide, dbuf, shape, pair, ahead in ((0, 0, 0, 0, 0), (0, 1, 0, 0, 0), (1, 0, 0, 0, 0), (3, 0, 0, 0, 1), (7, 0, 4, 0, 0), (1, 1, 0, 0, 1), (0, 1, 4, 0, 0),
                                                    (0, 0, 0, 1, 1), (1, 0, 0, 1, 0), (1, 0, 0, 1, 1), (3, 0, 4, 1, 0), (3, 0, 4, 1, 1), (2, 0, 0, 1, 1), (27, 0, 0, 1, 0),
                                                    (27, 0, 0, 1, 1), (-1, -1, 0, 1, -1), (-1, -1, 0, 0, -1)):
                ctx.set_option("spmm_tile_pair", pair)
                ctx.set_option("spmm_tile_ahead", ahead)
                ctx.set_option("spmm_tile_slide", slide); ctx.set_option("spmm_tile_dbuf", dbuf); ctx.set_option("spmm_tile_shape", shape)
                dB = make()                               # the group records are built per handle, at its first product
                dY2 = K.Panel(ctx, dB.m, p)
                K.spmm_(dB, dX, dY2)
                assert dB.tile_info["state"] == 1, (slide, dbuf, shape, pair, ahead, dB.tile_info)
                assert np.array_equal(dY2.to_host(), ref), (slide, dbuf, shape, pair, ahead)
                K.spmm_(dB, dX, dY2)                      # a second product on the same records
                assert np.array_equal(dY2.to_host(), ref), (slide, dbuf, shape, pair, ahead, "second product")
    finally:
        for k, v in saved.items():
            ctx.set_option(k, v)


def test_spmm_tile_general_operators_and_fallbacks(K, ctx, oracle):
    """No grid: (a) band + seeded long-range columns (identity order, window from the histogram of distinct columns);
    (b) band with dense rows (their groups are flagged: direct path inside the tile kernel); (c) scattered (no reuse: the
    handle keeps the other kernels, state -1); (d) rectangular; (e) ragged rows incl. empty ones, with Inf / NaN in X --
    the masked path must not pick up entries past the end of a row."""
    import scipy.sparse as sp
    rng = np.random.default_rng(11)
    n = 6000

    def check(S, tag, want_state=None, special=False):
        S = S.tocsr()
        S.sort_indices()
        dA = K.CsrMatrix.from_host(ctx, S.indptr.astype(np.int64), S.indices.astype(np.int32), S.data.copy(), S.shape)
        X = rng.standard_normal((S.shape[1], 16))
        if special:
            X[::97, 3] = np.inf
            X[5::131, 9] = np.nan
        Yt, Yw, Yd = _spmm_three(K, ctx, dA, X)
        info = dA.tile_info
        if want_state is not None:
            assert info["state"] == want_state, (tag, info)
        with np.errstate(invalid="ignore", over="ignore"):
            ref = _serial_spmm(S, X)
        assert np.array_equal(Yt, Yd, equal_nan=True) and np.array_equal(Yw, Yd, equal_nan=True), tag
        assert np.array_equal(Yt, ref, equal_nan=True), tag
        return info

    band = sp.diags([rng.standard_normal(n - abs(k)) for k in range(-10, 11)], list(range(-10, 11)), format="lil")
    far = sp.random(n, n, density=3.0 / n, random_state=5, format="lil")
    info = check((band + far).tocsr(), "band + random", want_state=1)
    assert info["grid_tiles"] == 0 and info["window"] >= 64, info
    band2 = sp.diags([rng.standard_normal(n - abs(k)) for k in range(-6, 7)], list(range(-6, 7)), format="lil")
    for r in (100, 101, 3333):
        band2[r, :] = rng.standard_normal(n) * (rng.random(n) < 0.3)
    info = check(band2, "band + dense rows", want_state=1)
    assert info["direct_groups"] >= 2, info
    check(sp.random(n, n, density=0.004, random_state=7) + sp.eye(n), "scattered", want_state=-1)
    check(sp.diags([np.ones(900), 2 * np.ones(900), 3 * np.ones(900)], [0, 40, 1], shape=(900, 1400)), "rectangular")
    ragged = sp.lil_matrix((n, n))
    for i in range(n):
        k = (i * 7) % 23
        cols = np.unique(np.clip(i + np.arange(k) - k // 2, 0, n - 1))
        if cols.size:
            ragged[i, cols] = rng.standard_normal(cols.size)
    check(ragged, "ragged rows", want_state=1, special=True)
    # exactly 32 and 33 entries in a row: the longest row of the register path, and the first one past it
    edge = sp.lil_matrix((400, 400))
    for i in range(400):
        k = 32 if i % 2 == 0 else (33 if i == 201 else 5)
        cols = (i + np.arange(k)) % 400
        edge[i, cols] = rng.standard_normal(k)
    info = check(edge, "rows of 32 / 33 entries", want_state=1)
    assert info["direct_groups"] >= 1, info


def test_spmm_tile_fuzz_small_and_odd_shapes(K, ctx):
    """Seeded fuzz over shapes the tile kernel's bookkeeping could trip on: fewer rows than one group, one row, a row count that
    is no multiple of 32, empty rows (also at the end), a single column, every entry in one column, banded patterns that do
    and do not reach the reuse threshold, 2-D grids of awkward extents.  Whatever kernel the handle ends up with (tile_info
    state 1 or -1), Y equals the direct kernel's and the serial loop's, bit for bit."""
    import scipy.sparse as sp
    rng = np.random.default_rng(2024)
    cases = []
    for m, n, dens in [(1, 1, 1.0), (1, 50, 0.5), (5, 5, 0.6), (31, 31, 0.3), (33, 40, 0.2), (64, 64, 0.05), (95, 200, 0.04), (129, 17, 0.5),
                       (300, 300, 0.01), (1000, 1000, 0.003)]:
        S = sp.random(m, n, density=dens, random_state=int(rng.integers(1 << 30)), format="lil")
        if m > 4:
            S[m - 1, :] = 0                                   # empty last row
            S[m // 2, :] = 0                                  # and one in the middle
        cases.append((f"random {m}x{n}", S.tocsr()))
    cases.append(("single column", sp.csr_matrix((np.ones(70), (np.arange(70), np.zeros(70, dtype=int))), shape=(70, 9))))
    for w in (1, 3, 9):
        k = 500
        cases.append((f"band {w}", sp.diags([rng.standard_normal(k - abs(d)) for d in range(-w, w + 1)], list(range(-w, w + 1)), format="csr")))
    for a, b in ((7, 5), (4, 4), (13, 3), (50, 9)):          # 5-point grids of awkward extents (8 x 4 x 1 tiles, partial ones)
        T = lambda q: sp.diags([-np.ones(q - 1), 2 * np.ones(q), -np.ones(q - 1)], [-1, 0, 1])
        cases.append((f"grid {a}x{b}", (sp.kron(sp.eye(b), T(a)) + sp.kron(T(b), sp.eye(a))).tocsr()))
    for tag, S in cases:
        S = S.tocsr()
        S.eliminate_zeros()
        S.sort_indices()
        m, n = S.shape
        dA = K.CsrMatrix.from_host(ctx, S.indptr.astype(np.int64), S.indices.astype(np.int32), S.data.astype(np.float64), (m, n))
        X = rng.standard_normal((n, 16))
        Yt, Yw, Yd = _spmm_three(K, ctx, dA, X)
        assert np.array_equal(Yt, Yd) and np.array_equal(Yw, Yd), (tag, dA.tile_info)
        assert np.array_equal(Yt, _serial_spmm(S, X)), (tag, dA.tile_info)


@pytest.mark.parametrize("p,slices", [(8, 0), (32, -1), (32, 1), (64, 1)])
def test_spmm_tile_other_widths_bit_identical(K, ctx, oracle, p, slices):
    """The tile kernel with 2 and 8 lanes per row (p = 8, 32 in one launch) and the 16-column kernel over column slices of
    wider panels (p = 32, 64): grid operators, band + long-range columns, band + dense rows (flagged groups: the direct kernel
    at that width / slice), ragged rows with Inf / NaN in X, odd small shapes.  Same records as p = 16; Y == window kernel ==
    direct kernel == serial loop."""
    import scipy.sparse as sp
    rng = np.random.default_rng(100 + p)
    for kind, dims in (("stencil27", (13, 11, 9)), ("poisson", (20, 17, 6)), ("poisson", (37, 9, 1))):
        dA = K.CsrMatrix.stencil(ctx, kind, *dims)
        X = rng.standard_normal((dA.n, p))
        Yt, Yw, Yd = _spmm_three(K, ctx, dA, X, slices)
        assert dA.tile_info["state"] == 1 and dA.tile_info["grid_tiles"] == 1
        assert np.array_equal(Yt, Yd) and np.array_equal(Yw, Yd), (kind, dims)
        if kind == "poisson":
            A = oracle.poisson3d(*dims)
            assert np.array_equal(Yt, np.stack([A.matvec(np.ascontiguousarray(X[:, j])) for j in range(p)], axis=1))
    n = 3000
    band = sp.diags([rng.standard_normal(n - abs(k)) for k in range(-10, 11)], list(range(-10, 11)), format="lil")
    far = sp.random(n, n, density=3.0 / n, random_state=5, format="lil")
    dense = sp.diags([rng.standard_normal(n - abs(k)) for k in range(-6, 7)], list(range(-6, 7)), format="lil")
    for r in (100, 101, 2222):
        dense[r, :] = rng.standard_normal(n) * (rng.random(n) < 0.3)
    ragged = sp.lil_matrix((n, n))
    for i in range(n):
        k = (i * 7) % 23
        cols = np.unique((i + np.arange(k) * 3 - k) % n)
        if cols.size:
            ragged[i, cols] = rng.standard_normal(cols.size)
    edge = sp.lil_matrix((400, 400))                          # rows of exactly 32 entries (every entry step of a pass) and one of 33
    for i in range(400):
        k = 32 if i % 2 == 0 else (33 if i == 201 else 5)
        edge[i, (i + np.arange(k)) % 400] = rng.standard_normal(k)
    small = sp.random(95, 200, density=0.04, random_state=3, format="lil")
    for tag, S, special in (("band + random", band + far, False), ("dense rows", dense, False), ("ragged", ragged, True), ("rows of 32 / 33", edge, False),
                            ("small", small, False)):
        S = S.tocsr()
        S.sort_indices()
        dA = K.CsrMatrix.from_host(ctx, S.indptr.astype(np.int64), S.indices.astype(np.int32), S.data.copy(), S.shape)
        X = rng.standard_normal((S.shape[1], p))
        if special:
            X[::97, 3] = np.inf
            X[5::131, p - 1] = np.nan
        Yt, Yw, Yd = _spmm_three(K, ctx, dA, X, slices)
        if tag == "dense rows":
            assert dA.tile_info["state"] == 1 and dA.tile_info["direct_groups"] >= 2, dA.tile_info
        with np.errstate(invalid="ignore", over="ignore"):
            ref = _serial_spmm(S, X)
        assert np.array_equal(Yt, Yd, equal_nan=True) and np.array_equal(Yw, Yd, equal_nan=True), tag
        assert np.array_equal(Yt, ref, equal_nan=True), tag


def test_block_gmres_same_history_with_and_without_tile_kernel(K, ctx, oracle):
    A = oracle.stencil27_unsym(12)
    S = A.to_scipy()
    B, _ = _rhs(S, A.n, 16)
    dA = K.CsrMatrix.from_host(ctx, A.rowptr, A.col, A.val, (A.n, A.n))
    res = []
    for tile in (1, 0):
        ctx.set_option("spmm_tile", tile)
        X, st, _ = K.block_gmres(dA, B, memory=5, ctx=ctx, history=True, restart=True, itmax=20)
        res.append((X, st.niter, np.array(st.residuals)))
    ctx.set_option("spmm_tile", 1)
    assert dA.tile_info["state"] == 1
    assert res[0][1] == res[1][1] and np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][0], res[1][0])


def test_block_gmres_same_history_with_and_without_window(K, ctx, oracle):
    A = oracle.stencil27_unsym(12)
    S = A.to_scipy()
    B, _ = _rhs(S, A.n, 8)
    dA = K.CsrMatrix.from_host(ctx, A.rowptr, A.col, A.val, (A.n, A.n))
    res = []
    for window in (1, 0):
        ctx.set_option("spmm_window", window)
        X, st, _ = K.block_gmres(dA, B, memory=6, ctx=ctx, history=True, restart=True, itmax=30)
        res.append((X, st.niter, np.array(st.residuals)))
    ctx.set_option("spmm_window", 1)
    assert res[0][1] == res[1][1] and np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][0], res[1][0])


# ---- block Gram-Schmidt sweep with the fused apply-and-project kernel (panel.hip, panel_nn_tn_kernel) -----------

@pytest.mark.parametrize("n,p,k", [(1000, 16, 1), (1000, 16, 4), (4099, 16, 3), (777, 8, 5), (5000, 32, 3), (300, 5, 2), (257, 24, 2)])
def test_panel_mgs_fused_bit_identical_to_two_kernel_sequence(K, ctx, n, p, k):
    """khip_panel_mgs (src/block_gmres.jl:244-247 and the reorthogonalisation pass :250-256): Q and every Psi block
    from the fused sweep equal the khip_panel_gemm_tn / khip_panel_gemm_nn sequence bit for bit."""
    rng = np.random.default_rng(n + p + k)
    Vall = np.linalg.qr(rng.standard_normal((n, p * k)))[0]
    Vh = [np.ascontiguousarray(Vall[:, i * p:(i + 1) * p]) for i in range(k)]
    Qh = rng.standard_normal((n, p))
    res = []
    for fuse in (1, 0):
        ctx.set_option("panel_fuse", fuse)
        V = [K.Panel.from_host(ctx, v) for v in Vh]
        Q = K.Panel.from_host(ctx, Qh)
        blocks = K.panel_mgs_(V, Q)
        again = K.panel_mgs_(V, Q, accumulate_into=blocks)
        res.append((Q.to_host(), np.array(blocks), np.array(again)))
    ctx.set_option("panel_fuse", 1)
    # the explicit primitive sequence
    V = [K.Panel.from_host(ctx, v) for v in Vh]
    Q = K.Panel.from_host(ctx, Qh)
    prim = []
    for i in range(k):
        psi = K.panel_gemm_tn(V[i], Q)
        K.panel_gemm_nn_(-1.0, V[i], psi, 1.0, Q)
        prim.append(psi)
    assert all(np.array_equal(a, b) for a, b in zip(res[0], res[1]))
    assert np.array_equal(res[0][1], np.array(prim))
    assert max(np.abs(v.T @ res[0][0]).max() for v in Vh) < 1e-13          # two sweeps: orthogonal to working precision
    assert np.allclose(np.array(prim), np.array([v.T @ Qh for v in Vh]), atol=1e-12)


@pytest.mark.parametrize("kw,p", [(dict(restart=True), 4), (dict(reorthogonalization=True), 4), (dict(), 4), (dict(restart=True), 20)])
def test_block_gmres_same_with_and_without_fused_sweep(K, ctx, oracle, kw, p):
    """panel_fuse = 1 (fused Gram-Schmidt sweep, fused QR round, one-pass solution update) vs 0 (one kernel per
    product): same iteration count, history and solution bit for bit; p = 20 takes the 2 x 2-tile instantiations."""
    A = oracle.stencil27_unsym(10)
    B, _ = _rhs(A.to_scipy(), A.n, p)
    dA = K.CsrMatrix.from_host(ctx, A.rowptr, A.col, A.val, (A.n, A.n))
    res = []
    for fuse in (1, 0):
        ctx.set_option("panel_fuse", fuse)
        X, st, _ = K.block_gmres(dA, B, memory=6, ctx=ctx, history=True, itmax=25, **kw)
        res.append((X, st.niter, st.status, np.array(st.residuals)))
    ctx.set_option("panel_fuse", 1)
    assert res[0][1] == res[1][1] and res[0][2] == res[1][2]
    assert np.array_equal(res[0][3], res[1][3]) and np.array_equal(res[0][0], res[1][0])


@pytest.mark.parametrize("n,p,cond", [(1000, 16, 1e2), (4099, 16, 1e10), (300, 5, 10.0), (5000, 32, 1e3)])
def test_panel_qr_same_with_and_without_fused_round(K, ctx, n, p, cond):
    """CholeskyQR2: the in-place scaling of round 1 fused with the Gram matrix of round 2 (panel_scale_gram) gives the
    same Q and R as the separate kernels, also after a shifted first pass (cond 1e10)."""
    rng = np.random.default_rng(n + p)
    U = np.linalg.qr(rng.standard_normal((n, p)))[0]
    W = np.linalg.qr(rng.standard_normal((p, p)))[0]
    A = U @ np.diag(np.logspace(0, -np.log10(cond), p)) @ W.T
    res = []
    for fuse in (1, 0):
        ctx.set_option("panel_fuse", fuse)
        Q = K.Panel.from_host(ctx, A)
        R = K.panel_qr_(Q)
        res.append((Q.to_host(), R))
    ctx.set_option("panel_fuse", 1)
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    Qh, R = res[0]
    assert np.allclose(Qh @ R, A, atol=1e-12 * np.abs(A).max() * cond ** 0 + 1e-13) and np.allclose(Qh.T @ Qh, np.eye(p), atol=1e-10)


def test_block_arnoldi_step_blocks_equal_the_oracles(K, ctx, oracle):
    """One block-Arnoldi step of block_gmres! issued primitive by primitive (src/block_gmres.jl:211-212, :242-247, :259):
    the basis panels AND the small blocks -- Z_1 (R factor of the residual block), Psi_1 = V_1' W, C (R factor of the
    orthogonalised W) -- equal the ones the oracle's Householder path (ko_householder = geqrf + orgqr) produces, entry by
    entry and with the same signs: the device QR is not merely an equivalent factorisation."""
    import ctypes as C
    A = oracle.kron_unsymmetric(9)
    n, p = A.n, 16
    S = A.to_scipy()
    B, _ = _rhs(S, n, p)
    L = oracle.lib()
    dp = C.POINTER(C.c_double)

    def householder(M):
        Q = np.asfortranarray(M.copy())
        R = np.zeros((p, p), order="F")
        tau = np.zeros(p)
        L.ko_householder(n, p, Q.ctypes.data_as(dp), R.ctypes.data_as(dp), tau.ctypes.data_as(dp), 0)
        return Q, R, tau
    V1, Z1, tau1 = householder(B)
    W = np.stack([A.matvec(np.ascontiguousarray(V1[:, j])) for j in range(p)], axis=1)
    Psi = V1.T @ W
    W2 = W - V1 @ Psi
    Q2, Cb, tau2 = householder(W2)

    dA = K.CsrMatrix.from_host(ctx, A.rowptr, A.col, A.val, (n, n))
    dV = K.Panel.from_host(ctx, B)
    Z1g, tau1g = K.panel_qr_tau_(dV)
    dW = K.Panel(ctx, n, p)
    K.spmm_(dA, dV, dW)
    Psig = K.panel_gemm_tn(dV, dW)
    K.panel_gemm_nn_(-1.0, dV, Psig, 1.0, dW)
    Cg, tau2g = K.panel_qr_tau_(dW)

    def close(X, Y, tol):
        return float(np.max(np.abs(X - Y))) <= tol * max(1.0, float(np.max(np.abs(Y))))
    assert close(dV.to_host(), V1, 1e-11) and close(Z1g, Z1, 1e-11) and close(tau1g, tau1, 1e-11)
    assert close(Psig, Psi, 1e-10)
    assert close(dW.to_host(), Q2, 1e-9) and close(Cg, Cb, 1e-9) and close(tau2g, tau2, 1e-9)


@pytest.mark.parametrize("n,p,k", [(16, 16, 1), (1000, 16, 5), (33333, 16, 3), (5000, 4, 6), (4096, 32, 3), (3000, 7, 2), (20000, 16, 24)])
def test_panel_multi_nn_equals_the_product_sequence(K, ctx, n, p, k):
    """X += sum V_i Y_i (src/block_gmres.jl:324-326) in one pass: bit-identical to the k mul!(Xr, V[i], Y[i], 1, 1) calls,
    with the factor blocks re-read per tile (panel_multi_tiles = 0) or held in LDS with 1, 4 or 8 tiles per wave."""
    rng = np.random.default_rng(n + p + k)
    Vh = [rng.standard_normal((n, p)) for _ in range(k)]
    Yh = [rng.standard_normal((p, p)) for _ in range(k)]
    X0 = rng.standard_normal((n, p))
    Vs = [K.Panel.from_host(ctx, v) for v in Vh]
    ref = K.Panel.from_host(ctx, X0)
    for v, y in zip(Vs, Yh):
        K.panel_gemm_nn_(1.0, v, y, 1.0, ref)
    ref_h = ref.to_host()
    saved = ctx.get_option("panel_multi_tiles")
    try:
        for tiles in (0, 1, 4, 8):
            ctx.set_option("panel_multi_tiles", tiles)
            X = K.Panel.from_host(ctx, X0)
            K.panel_multi_nn_(Vs, Yh, 1.0, X)
            assert np.array_equal(X.to_host(), ref_h), (n, p, k, tiles)
        # beta = 0 ignores the old X (the first product of a restart-free solve)
        X = K.Panel.from_host(ctx, np.full((n, p), np.nan))
        K.panel_multi_nn_(Vs, Yh, 0.0, X)
        ref0 = K.Panel.from_host(ctx, X0)
        K.panel_gemm_nn_(1.0, Vs[0], Yh[0], 0.0, ref0)
        for v, y in zip(Vs[1:], Yh[1:]):
            K.panel_gemm_nn_(1.0, v, y, 1.0, ref0)
        assert np.array_equal(X.to_host(), ref0.to_host())
    finally:
        ctx.set_option("panel_multi_tiles", saved)
