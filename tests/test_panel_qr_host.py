"""Host-only checks of the p x p steps of the panel QR (csrc/block.cpp: deflating_chol, householder_r_host, householder_signs)
through the test exports khip_test_*: no device needed.  What they must deliver is what LAPACK's dgeqrf / dorgqr deliver
(src/block_krylov_utils.jl:201-208 householder! = kgeqrf! + korgqr!), checked against numpy / scipy."""
import ctypes as C

import numpy as np
import pytest
import scipy.linalg as sla


def _lib():
    import krylov_jl_amd as K
    return K.lib()


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _deflating_chol(G, tol=1e-7, detect=True, preset=0):
    p = G.shape[0]
    Gc = np.array(G, dtype=np.float64, order="F")
    R = np.zeros((p, p), order="F")
    ok, mask = C.c_int(), C.c_uint()
    rc = _lib().khip_test_deflating_chol(p, _dp(Gc), tol, 1 if detect else 0, preset, _dp(R), C.byref(ok), C.byref(mask))
    assert rc == 0
    return mask.value, R, bool(ok.value)


def test_deflating_chol_full_rank_is_the_cholesky_factor():
    rng = np.random.default_rng(1)
    A = rng.standard_normal((400, 7))
    G = A.T @ A
    mask, R, ok = _deflating_chol(G)
    assert ok and mask == 0
    assert np.allclose(R, np.linalg.cholesky(G).T, rtol=1e-12, atol=1e-12)
    assert np.allclose(np.tril(R, -1), 0)


def test_deflating_chol_leaves_out_dependent_columns():
    """An equal column, a linear combination, a zero column: exactly those are reported; the factor of the OTHER columns is the
    Cholesky factor of their Gram matrix; a left-out column carries its coefficients along the kept ones, R_jj = 1, zero row --
    so that A Rhat^-1 has orthonormal kept columns and rounding dust in the others."""
    rng = np.random.default_rng(2)
    n, p = 2000, 8
    A = rng.standard_normal((n, p))
    A[:, 3] = A[:, 1]
    A[:, 5] = 2.0 * A[:, 0] - 0.5 * A[:, 2]
    A[:, 6] = 0.0
    mask, R, ok = _deflating_chol(A.T @ A)
    assert ok and mask == (1 << 3) | (1 << 5) | (1 << 6)
    keep = [0, 1, 2, 4, 7]
    Rk = np.linalg.cholesky(A[:, keep].T @ A[:, keep]).T
    assert np.allclose(R[np.ix_(keep, keep)], Rk, rtol=1e-10, atol=1e-10)
    for j in (3, 5, 6):
        assert R[j, j] == 1.0 and np.all(R[j, j + 1:] == 0.0)
    Q = A @ np.linalg.inv(R)
    assert np.max(np.abs(Q[:, keep].T @ Q[:, keep] - np.eye(len(keep)))) <= 1e-10
    assert np.max(np.abs(Q[:, [3, 5, 6]])) <= 1e-10 * np.sqrt(n)


def test_deflating_chol_with_a_preset_set_reports_other_vanishing_pivots():
    rng = np.random.default_rng(3)
    A = rng.standard_normal((300, 5))
    A[:, 2] = A[:, 0]
    A[:, 4] = A[:, 1]
    mask, R, ok = _deflating_chol(A.T @ A, tol=0.0, detect=False, preset=1 << 2)
    assert mask == 1 << 2
    # column 4 equals column 1 and is not in the preset set: its pivot is rounding noise (it may even come out negative)
    assert (not ok) or abs(R[4, 4]) <= 1e-5
    mask, R, ok = _deflating_chol(A.T @ A, tol=0.0, detect=False, preset=(1 << 2) | (1 << 4))
    assert ok and mask == (1 << 2) | (1 << 4)
    mask, R, ok = _deflating_chol(np.zeros((3, 3)))
    assert not ok


@pytest.mark.parametrize("rows,p", [(48, 16), (7, 7), (96, 32), (5, 8)])
def test_householder_r_is_lapacks_r_up_to_row_signs(rows, p):
    rng = np.random.default_rng(rows + p)
    A = rng.standard_normal((rows, p))
    work = A.copy()                                  # the export overwrites its input
    R = np.zeros((p, p))
    assert _lib().khip_test_householder_r(rows, p, _dp(work), _dp(R)) == 0
    Rl = np.zeros((p, p))
    Rl[: min(rows, p)] = sla.qr(A, mode="r")[0][: min(rows, p)]
    assert np.allclose(np.tril(R, -1), 0)
    assert np.allclose(np.abs(R), np.abs(Rl), rtol=1e-11, atol=1e-11)
    k = min(rows, p)
    assert np.allclose(R[:k].T @ R[:k], A.T @ A if rows >= p else R[:k].T @ R[:k], rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("n,p", [(200, 6), (64, 16), (9, 9), (40, 1)])
def test_householder_signs_and_tau_are_dgeqrfs(n, p):
    """From the top block of an orthonormal panel alone: S_j = the sign LAPACK's R_jj takes relative to the positive-diagonal
    factor, tau_j = dgeqrf's (Ballard et al. 2014; DLARFG's conventions incl. tau = 0 for the last column of a square block)."""
    rng = np.random.default_rng(10 * n + p)
    A = rng.standard_normal((n, p))
    (qr_raw, tau_l), R_l = sla.qr(A, mode="raw")
    Qpos, Rpos = np.linalg.qr(A)
    sgn = np.sign(np.diag(Rpos))
    Qpos, Rpos = Qpos * sgn, (Rpos.T * sgn).T                      # the positive-diagonal (Cholesky-type) factors
    Q1 = np.array(Qpos[:p, :p] if n >= p else Qpos, order="C")    # overwritten
    S, tau = np.zeros(p), np.zeros(p)
    assert _lib().khip_test_householder_signs(p, n, _dp(Q1), _dp(S), _dp(tau)) == 0
    assert np.array_equal(S, np.sign(np.diag(R_l)) + (np.diag(R_l) == 0))
    assert np.allclose(tau, tau_l, rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("m,n", [(32, 16), (8, 4), (64, 32), (2, 1), (6, 6)])
def test_small_dense_routines_are_lapacks(m, n):
    """DGEQR2 / DORG2R / DORM2R('L', 'T') and the triangular inverse as block_gmres! uses them on its 2p x p Hessenberg blocks
    (src/block_gmres.jl:263-284 = householder!(H, R, tau; compact = true) and kormqr!('L', 'T', H, tau, D)): equal to LAPACK's
    geqrf / orgqr / ormqr to rounding, same signs, same tau."""
    rng = np.random.default_rng(100 * m + n)
    A = rng.standard_normal((m, n))
    L = _lib()
    a = np.array(A, order="F")
    tau = np.zeros(n)
    assert L.khip_test_small_dense(0, m, n, 0, _dp(a), _dp(tau), None) == 0
    (qr_l, tau_l), _ = sla.qr(A, mode="raw")
    assert np.allclose(a, qr_l, rtol=1e-12, atol=1e-12) and np.allclose(tau, tau_l, rtol=1e-12, atol=1e-14)
    Cm = rng.standard_normal((m, 5))
    c = np.array(Cm, order="F")
    assert L.khip_test_small_dense(2, m, n, 5, _dp(a), _dp(tau), _dp(c)) == 0
    Qfull = sla.qr(A, mode="full")[0]
    assert np.allclose(c, Qfull.T @ Cm, rtol=1e-11, atol=1e-12)
    q = a.copy(order="F")
    assert L.khip_test_small_dense(1, m, n, 0, _dp(q), _dp(tau), None) == 0
    assert np.allclose(q, sla.qr(A, mode="economic")[0], rtol=1e-11, atol=1e-12)
    U = np.triu(rng.standard_normal((n, n))) + 3.0 * np.eye(n)
    u, ui = np.array(U, order="F"), np.zeros((n, n), order="F")
    assert L.khip_test_small_dense(3, n, n, 0, _dp(u), None, _dp(ui)) == 0
    assert np.allclose(ui, np.linalg.inv(U), rtol=1e-11, atol=1e-12) and np.allclose(np.tril(ui, -1), 0)
