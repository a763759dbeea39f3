"""GPU parity of the six Krylov processes (khip_hermitian_lanczos / khip_nonhermitian_lanczos / khip_arnoldi /
khip_golub_kahan / khip_saunders_simon_yip / khip_montoison_orban)
against the CPU oracle (oracle/oracle_processes.py) through the C ABI, plus the reference's own assertions
(test/test_processes.jl:31-190,194-234, shared with the oracle pins through tests/process_checks.py) evaluated on
the device results.

Tolerances (fp64): the processes are short recurrences whose rounding differences grow with the step count, so the
parity bound is stated per quantity for k = 20 steps:
  * beta (= ||b||):                         |d| <= 4 eps beta
  * T / H / L entries:                      |d| <= COEF_RTOL * max|entry|     with COEF_RTOL = 1e-10
  * basis columns:                          ||dV_j||_inf <= BASIS_TOL         with BASIS_TOL = 1e-8 (unit-norm columns)
The measured deviations are logged to gpurun_out/parity_log.jsonl.
"""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EPS = np.finfo(float).eps
COEF_RTOL = 1e-10
BASIS_TOL = 1e-8


def _approx(a, b):          # Julia's `≈` for arrays
    return np.linalg.norm(a - b) <= math.sqrt(EPS) * max(np.linalg.norm(a), np.linalg.norm(b))


def _upload(K, ctx, S):
    S = S.tocsr()
    S.sort_indices()
    return K.CsrMatrix.from_host(ctx, S.indptr.astype(np.int64), S.indices.astype(np.int32), S.data.copy(), S.shape)


def _serial_matvec(S):
    """y = S x with every row accumulated left to right (the order of the device kernel and of ko_spmv)."""
    S = S.tocsr()
    S.sort_indices()
    return lambda x: S @ x          # scipy's csr_matvec is the same serial loop per row


def _spd_sparse(n, seed):
    import scipy.sparse as sp
    R = sp.random(n, n, density=8.0 / n, random_state=seed, format="csr")
    return (R + R.T + sp.diags(np.full(n, 12.0))).tocsr()


@pytest.mark.parametrize("reorth", [False, True])
@pytest.mark.parametrize("n", [500, 20011])
def test_hermitian_lanczos_matches_oracle(K, ctx, oracle, parity_log, n, reorth):
    import oracle_processes as P
    k, s = 20, 5
    S = _spd_sparse(n, 3)
    b = np.random.default_rng(n).random(n)
    Vr, br, nzr = P.hermitian_lanczos(_serial_matvec(S), b, k, reorthogonalization=reorth)
    dA = _upload(K, ctx, S)
    V, beta1, T = K.hermitian_lanczos(dA, ctx.array(b), k, reorthogonalization=reorth)
    Vh, Th = V.to_host(), T.toarray()
    assert V.shape == (n, k + 1) and T.shape == (k + 1, k)
    # the reference's assertions (test/test_processes.jl:39-41)
    assert np.linalg.norm(Vh[:, :s].T @ Vh[:, :s] - np.eye(s)) <= 1e-4
    assert _approx(beta1 * Vh[:, 0], b)
    assert _approx(S @ Vh[:, :k], Vh @ Th)
    # parity with the oracle
    assert abs(beta1 - br) <= 4 * EPS * br
    dcoef = float(np.max(np.abs(T.data - nzr)) / np.max(np.abs(nzr)))
    dbasis = float(np.max(np.abs(Vh - Vr)))
    parity_log(test="hermitian_lanczos", n=n, k=k, reorth=reorth, coef_rel=dcoef, basis_abs=dbasis)
    assert dcoef <= COEF_RTOL and dbasis <= BASIS_TOL


@pytest.mark.parametrize("reorth", [False, True])
def test_arnoldi_matches_oracle(K, ctx, oracle, parity_log, reorth):
    import oracle_processes as P
    k, s = 20, 5
    A = oracle.kron_unsymmetric(12)                   # 1728 rows, unsymmetric
    S = A.to_scipy()
    n = A.n
    b = np.random.default_rng(7).random(n)
    Vr, br, Hr = P.arnoldi(A.matvec, b, k, reorthogonalization=reorth)
    dA = K.CsrMatrix.from_host(ctx, A.rowptr, A.col, A.val, (n, n))
    V, beta, H = K.arnoldi(dA, ctx.array(b), k, reorthogonalization=reorth)
    Vh = V.to_host()
    assert H.shape == (k + 1, k) and np.all(np.tril(H, -2) == 0.0)
    assert np.linalg.norm(Vh[:, :s].T @ Vh[:, :s] - np.eye(s)) <= 1e-4          # test/test_processes.jl:82-84
    assert _approx(beta * Vh[:, 0], b)
    assert _approx(S @ Vh[:, :k], Vh @ H)
    assert abs(beta - br) <= 4 * EPS * br
    dcoef = float(np.max(np.abs(H - Hr)) / np.max(np.abs(Hr)))
    dbasis = float(np.max(np.abs(Vh - Vr)))
    parity_log(test="arnoldi", n=n, k=k, reorth=reorth, coef_rel=dcoef, basis_abs=dbasis)
    assert dcoef <= COEF_RTOL and dbasis <= BASIS_TOL
    if reorth:                                         # full reorthogonalisation: the whole basis is orthonormal
        assert np.linalg.norm(Vh.T @ Vh - np.eye(k + 1)) <= 1e-12


def test_arnoldi_with_callback_operator_is_the_same(K, ctx, oracle):
    """A as an apply callback (kmul! on a user operator) gives bit-identical output to the CSR handle."""
    A = oracle.kron_unsymmetric(8)
    n, k = A.n, 6
    b = ctx.array(np.random.default_rng(1).random(n))
    dA = K.CsrMatrix.from_host(ctx, A.rowptr, A.col, A.val, (n, n))
    V1, b1, H1 = K.arnoldi(dA, b, k)
    V2, b2, H2 = K.arnoldi(lambda x, y: dA.matvec(x, y), b, k)
    assert b1 == b2 and np.array_equal(H1, H2) and np.array_equal(V1.to_host(), V2.to_host())


def test_golub_kahan_matches_oracle(K, ctx, oracle, parity_log):
    import oracle_processes as P
    import scipy.sparse as sp
    m, n, k, s = 250, 500, 20, 5                       # the reference's sizes (test/test_processes.jl:19-21)
    S = sp.random(m, n, density=0.05, random_state=11, format="csr")
    S.sort_indices()
    St = S.T.tocsr()
    St.sort_indices()
    b = np.random.default_rng(5).random(m)
    Vr, Ur, br, nzr = P.golub_kahan(_serial_matvec(S), _serial_matvec(St), b, n, k)
    dA = _upload(K, ctx, S)
    V, U, beta1, L = K.golub_kahan(dA, ctx.array(b), k)
    Vh, Uh, Lh = V.to_host(), U.to_host(), L.toarray()
    B = Lh[:k + 1, :k]
    assert V.shape == (n, k + 1) and U.shape == (m, k + 1) and L.shape == (k + 1, k + 1)
    # test/test_processes.jl:104-110
    assert np.linalg.norm(Vh[:, :s].T @ Vh[:, :s] - np.eye(s)) <= 1e-4
    assert np.linalg.norm(Uh[:, :s].T @ Uh[:, :s] - np.eye(s)) <= 1e-4
    assert _approx(beta1 * Uh[:, 0], b)
    assert _approx(S @ Vh[:, :k], Uh @ B)
    assert _approx(S.T @ Uh, Vh @ Lh.T)
    assert _approx(S.T @ (S @ Vh[:, :k]), Vh @ Lh.T @ B)
    assert _approx(S @ (S.T @ Uh[:, :k]), Uh @ B @ Lh[:k, :k].T)
    dcoef = float(np.max(np.abs(L.data - nzr)) / np.max(np.abs(nzr)))
    dbasis = float(max(np.max(np.abs(Vh - Vr)), np.max(np.abs(Uh - Ur))))
    parity_log(test="golub_kahan", m=m, n=n, k=k, coef_rel=dcoef, basis_abs=dbasis)
    assert abs(beta1 - br) <= 4 * EPS * br and dcoef <= COEF_RTOL and dbasis <= BASIS_TOL


def test_processes_exact_breakdown(K, ctx):
    """test/test_processes.jl:194-218: A0 = I (2 x 2), b0 = 0 -> the reference's error text, or zero vectors with
    allow_breakdown; plus an invariant subspace after one step."""
    I2 = K.CsrMatrix.from_host(ctx, np.array([0, 1, 2], dtype=np.int64), np.array([0, 1], dtype=np.int32),
                               np.ones(2), (2, 2))
    b0 = ctx.zeros(2)
    with pytest.raises(K.KhipError, match="Exact breakdown β₁ == 0."):
        K.hermitian_lanczos(I2, b0, 2)
    V, beta, T = K.hermitian_lanczos(I2, b0, 2, allow_breakdown=True)
    assert beta == 0.0 and not V.to_host().any() and not T.toarray().any()
    with pytest.raises(K.KhipError, match="Exact breakdown β == 0."):
        K.arnoldi(I2, b0, 2)
    V, beta, H = K.arnoldi(I2, b0, 2, allow_breakdown=True)
    assert beta == 0.0 and not V.to_host().any() and not H.any()
    with pytest.raises(K.KhipError, match="Exact breakdown β₁ == 0."):
        K.golub_kahan(I2, b0, 2)
    K.golub_kahan(I2, b0, 2, allow_breakdown=True)
    e1 = ctx.array(np.array([1.0, 0.0]))
    with pytest.raises(K.KhipError, match="Exact breakdown βᵢ₊₁ == 0 at iteration i = 1."):
        K.hermitian_lanczos(I2, e1, 2)
    with pytest.raises(K.KhipError, match="Exact breakdown Hᵢ₊₁.ᵢ == 0 at iteration i = 1."):
        K.arnoldi(I2, e1, 2)
    V, beta, H = K.arnoldi(I2, e1, 2, allow_breakdown=True)
    assert beta == 1.0 and np.array_equal(V.to_host()[:, 0], [1.0, 0.0]) and not V.to_host()[:, 1:].any()
    assert H[0, 0] == 1.0 and H[1, 0] == 0.0


def test_process_argument_checks(K, ctx):
    import ctypes as C
    I2 = K.CsrMatrix.from_host(ctx, np.array([0, 1, 2], dtype=np.int64), np.array([0, 1], dtype=np.int32),
                               np.ones(2), (2, 2))
    b = ctx.array(np.ones(2))
    keep = []
    op = K._make_operator(ctx, I2, 2, keep)
    V = K.DeviceMatrix(ctx, 2, 3)
    beta, H = C.c_double(), np.zeros(6)
    bad = K.lib().khip_arnoldi(ctx._h, op, 2, b.ptr, 2, 0, 0, V.ptr + 8, V.ld, C.byref(beta), H.ctypes.data_as(K.c_double_p))
    assert bad == -1 and b"16-byte aligned" in K.lib().khip_last_error()
    bad = K.lib().khip_arnoldi(ctx._h, op, 2, b.ptr, 2, 0, 0, V.ptr, 1, C.byref(beta), H.ctypes.data_as(K.c_double_p))
    assert bad == -1
    bad = K.lib().khip_arnoldi(ctx._h, op, 2, b.ptr, 0, 0, 0, V.ptr, V.ld, C.byref(beta), H.ctypes.data_as(K.c_double_p))
    assert bad == -1


# ---------------------------------------------------------------------------------- two-sided processes

def _rand_sparse(m, n, seed, density=0.05, shift=0.0):
    import scipy.sparse as sp
    S = sp.random(m, n, density=density, random_state=seed, format="csr")
    if shift:
        S = S + shift * sp.eye(m, n, format="csr")
    S = S.tocsr()
    S.sort_indices()
    return S


def _T(S):
    St = S.T.tocsr()
    St.sort_indices()
    return St


def test_nonhermitian_lanczos_matches_oracle(K, ctx, oracle, parity_log):
    import oracle_processes as P
    import process_checks as pc
    n, k = 500, 20
    S = _rand_sparse(n, n, 21, shift=4.0)
    rng = np.random.default_rng(8)
    b, c = rng.random(n), rng.random(n)
    Vr, b1r, ntr, Ur, g1r, nhr = P.nonhermitian_lanczos(_serial_matvec(S), _serial_matvec(_T(S)), b, c, k)
    V, b1, T, U, g1, Tt = K.nonhermitian_lanczos(_upload(K, ctx, S), ctx.array(b), ctx.array(c), k)
    Vh, Uh = V.to_host(), U.to_host()
    pc.check_nonhermitian_lanczos(S.toarray(), b, c, k, Vh, b1, T.toarray(), Uh, g1, Tt.toarray())   # test/test_processes.jl:59-65
    dcoef = float(max(np.max(np.abs(T.data - ntr)), np.max(np.abs(Tt.data - nhr))) / np.max(np.abs(ntr)))
    dbasis = float(max(np.max(np.abs(Vh - Vr)), np.max(np.abs(Uh - Ur))))
    parity_log(test="nonhermitian_lanczos", n=n, k=k, coef_rel=dcoef, basis_abs=dbasis)
    assert abs(b1 - b1r) <= 4 * EPS * b1r and abs(g1 - g1r) <= 4 * EPS * abs(g1r)
    assert dcoef <= COEF_RTOL and dbasis <= BASIS_TOL


def test_saunders_simon_yip_matches_oracle(K, ctx, oracle, parity_log):
    import oracle_processes as P
    import process_checks as pc
    m, n, k = 250, 500, 20
    S = _rand_sparse(m, n, 22)
    rng = np.random.default_rng(9)
    b, c = rng.random(m), rng.random(n)
    Vr, b1r, ntr, Ur, g1r, nhr = P.saunders_simon_yip(_serial_matvec(S), _serial_matvec(_T(S)), b, c, k)
    V, b1, T, U, g1, Tt = K.saunders_simon_yip(_upload(K, ctx, S), ctx.array(b), ctx.array(c), k)
    Vh, Uh = V.to_host(), U.to_host()
    assert V.shape == (m, k + 1) and U.shape == (n, k + 1)
    pc.check_saunders_simon_yip(S.toarray(), b, c, k, Vh, b1, T.toarray(), Uh, g1, Tt.toarray())     # test/test_processes.jl:126-142
    dcoef = float(max(np.max(np.abs(T.data - ntr)), np.max(np.abs(Tt.data - nhr))) / np.max(np.abs(ntr)))
    dbasis = float(max(np.max(np.abs(Vh - Vr)), np.max(np.abs(Uh - Ur))))
    parity_log(test="saunders_simon_yip", m=m, n=n, k=k, coef_rel=dcoef, basis_abs=dbasis)
    assert abs(b1 - b1r) <= 4 * EPS * b1r and abs(g1 - g1r) <= 4 * EPS * g1r
    assert dcoef <= COEF_RTOL and dbasis <= BASIS_TOL


@pytest.mark.parametrize("reorth", [False, True])
def test_montoison_orban_matches_oracle(K, ctx, oracle, parity_log, reorth):
    import oracle_processes as P
    import process_checks as pc
    m, n, k = 250, 500, 20
    SA, SB = _rand_sparse(m, n, 23), _rand_sparse(n, m, 24)
    rng = np.random.default_rng(10)
    b, c = rng.random(m), rng.random(n)
    Vr, br, Hr, Ur, gr, Fr = P.montoison_orban(_serial_matvec(SA), _serial_matvec(SB), b, c, k, reorthogonalization=reorth)
    V, beta, H, U, gamma, F = K.montoison_orban(_upload(K, ctx, SA), _upload(K, ctx, SB), ctx.array(b), ctx.array(c), k,
                                                reorthogonalization=reorth)
    Vh, Uh = V.to_host(), U.to_host()
    pc.check_montoison_orban(SA.toarray(), SB.toarray(), b, c, k, Vh, beta, H, Uh, gamma, F)         # test/test_processes.jl:155-171
    dcoef = float(max(np.max(np.abs(H - Hr)) / np.max(np.abs(Hr)), np.max(np.abs(F - Fr)) / np.max(np.abs(Fr))))
    dbasis = float(max(np.max(np.abs(Vh - Vr)), np.max(np.abs(Uh - Ur))))
    parity_log(test="montoison_orban", m=m, n=n, k=k, reorth=reorth, coef_rel=dcoef, basis_abs=dbasis)
    assert abs(beta - br) <= 4 * EPS * br and abs(gamma - gr) <= 4 * EPS * gr
    assert dcoef <= COEF_RTOL and dbasis <= BASIS_TOL


def test_two_sided_exact_breakdowns(K, ctx):
    """test/test_processes.jl:205-234: A0 = I and the reference's ssy_mo_breakdown{,2,3} matrices
    (test/test_utils.jl:396-420) -- same error text at the same iteration."""
    import process_checks as pc
    import scipy.sparse as sp

    def dev(M):
        return _upload(K, ctx, sp.csr_matrix(M))

    def raises(msg, fn, *a, **kw):
        with pytest.raises(K.KhipError) as e:
            fn(*a, **kw)
        assert str(e.value).endswith(msg), str(e.value)

    A0, b0, c0 = dev(np.eye(2)), ctx.zeros(2), ctx.array(np.ones(2))
    (M1, b1, c1), (M2, b2, c2), (M3, b3, c3) = pc.ssy_mo_breakdown(), pc.ssy_mo_breakdown2(), pc.ssy_mo_breakdown3()
    A1, A2, A3 = dev(M1), dev(M2), dev(M3)
    A1t, A2t, A3t = dev(M1.T), dev(M2.T), dev(M3.T)
    d = ctx.array
    raises("Exact breakdown β₁γ₁ == 0.", K.nonhermitian_lanczos, A0, b0, c0, 2)
    V, beta, T, U, gamma, Tt = K.nonhermitian_lanczos(A0, b0, c0, 2, allow_breakdown=True)
    assert beta == 0.0 and gamma == 0.0 and not V.to_host().any() and not U.to_host().any() and not T.toarray().any()
    raises("Exact breakdown β₁ == 0.", K.saunders_simon_yip, A0, b0, c0, 2)
    K.saunders_simon_yip(A0, b0, c0, 2, allow_breakdown=True)
    raises("Exact breakdown γ₁ᴴ == 0.", K.saunders_simon_yip, A0, c0, b0, 2)
    K.saunders_simon_yip(A0, c0, b0, 2, allow_breakdown=True)
    raises("Exact breakdown βᵢ₊₁ == 0 at iteration i = 1.", K.saunders_simon_yip, A1, d(b1), d(c1), 1)
    raises("Exact breakdown βᵢ₊₁ == 0 at iteration i = 2.", K.saunders_simon_yip, A2, d(b2), d(c2), 2)
    raises("Exact breakdown γᵢ₊₁ == 0 at iteration i = 2.", K.saunders_simon_yip, A3, d(b3), d(c3), 2)
    raises("Exact breakdown β == 0.", K.montoison_orban, A0, A0, b0, c0, 2)
    K.montoison_orban(A0, A0, b0, c0, 2, allow_breakdown=True)
    raises("Exact breakdown γ == 0.", K.montoison_orban, A0, A0, c0, b0, 2)
    K.montoison_orban(A0, A0, c0, b0, 2, allow_breakdown=True)
    raises("Exact breakdown Hᵢ₊₁.ᵢ == 0 at iteration i = 1.", K.montoison_orban, A1, A1t, d(b1), d(c1), 1)
    raises("Exact breakdown Hᵢ₊₁.ᵢ == 0 at iteration i = 2.", K.montoison_orban, A2, A2t, d(b2), d(c2), 2)
    raises("Exact breakdown Fᵢ₊₁.ᵢ == 0 at iteration i = 2.", K.montoison_orban, A3, A3t, d(b3), d(c3), 2)
