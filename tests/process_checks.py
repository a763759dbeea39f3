"""The assertions of the reference's test/test_processes.jl, as functions of host arrays, shared by the oracle pin
tests (tests/test_oracle.py) and the device parity tests (tests/test_gpu_processes.py)."""
import math

import numpy as np

S_LEAD = 5          # `s = 5`: orthonormality is asserted on the leading columns only (test/test_processes.jl:22)


def approx(a, b):
    """Julia's `≈` for arrays: norm(a - b) <= sqrt(eps) * max(norm(a), norm(b))."""
    a, b = np.asarray(a), np.asarray(b)
    return np.linalg.norm(a - b) <= math.sqrt(np.finfo(float).eps) * max(np.linalg.norm(a), np.linalg.norm(b))


def orthonormal_lead(V, W=None, s=S_LEAD):
    W = V if W is None else W
    return np.linalg.norm(V[:, :s].T @ W[:, :s] - np.eye(s)) <= 1e-4


def permutation_paige(k):
    """test/test_processes.jl:8-15."""
    P = np.zeros((2 * k, 2 * k))
    for i in range(k):
        P[i, 2 * i] = 1.0
        P[i + k, 2 * i + 1] = 1.0
    return P


def check_nonhermitian_lanczos(A, b, c, k, V, beta1, T, U, gamma1, Tt):
    """test/test_processes.jl:59-65."""
    assert orthonormal_lead(V, U) and orthonormal_lead(U, V)
    assert approx(beta1 * V[:, 0], b) and approx(gamma1 * U[:, 0], c)
    assert approx(T[:k, :k], Tt[:k, :k].T)
    assert approx(A @ V[:, :k], V @ T)
    assert approx(A.T @ U[:, :k], U @ Tt)


def _paige_block(A, B, k, V, U, T, Tt):
    m, n = A.shape
    Kmat = np.block([[np.zeros((m, m)), A], [B, np.zeros((n, n))]])
    Wk = np.block([[V[:, :k], np.zeros((m, k))], [np.zeros((n, k)), U[:, :k]]]) @ permutation_paige(k)
    Wk1 = np.block([[V, np.zeros((m, k + 1))], [np.zeros((n, k + 1)), U]]) @ permutation_paige(k + 1)
    G = permutation_paige(k + 1).T @ np.block([[np.zeros((k + 1, k)), T], [Tt, np.zeros((k + 1, k))]]) @ permutation_paige(k)
    return approx(Kmat @ Wk, Wk1 @ G)


def check_saunders_simon_yip(A, b, c, k, V, beta1, T, U, gamma1, Tt):
    """test/test_processes.jl:126-142."""
    assert orthonormal_lead(V) and orthonormal_lead(U)
    assert approx(beta1 * V[:, 0], b) and approx(gamma1 * U[:, 0], c)
    assert approx(T[:k, :k], Tt[:k, :k].T)
    assert approx(A @ U[:, :k], V @ T)
    assert approx(A.T @ V[:, :k], U @ Tt)
    assert approx(A.T @ (A @ U[:, :k - 1]), U @ Tt @ T[:k, :k - 1])
    assert approx(A @ (A.T @ V[:, :k - 1]), V @ T @ Tt[:k, :k - 1])
    assert _paige_block(A, A.T, k, V, U, T, Tt)


def check_montoison_orban(A, B, b, c, k, V, beta, H, U, gamma, F):
    """test/test_processes.jl:155-171."""
    assert orthonormal_lead(V) and orthonormal_lead(U)
    assert approx(beta * V[:, 0], b) and approx(gamma * U[:, 0], c)
    assert approx(A @ U[:, :k], V @ H)
    assert approx(B @ V[:, :k], U @ F)
    assert approx(B @ (A @ U[:, :k - 1]), U @ F @ H[:k, :k - 1])
    assert approx(A @ (B @ V[:, :k - 1]), V @ H @ F[:k, :k - 1])
    assert _paige_block(A, B, k, V, U, H, F)


# the reference's breakdown matrices, test/test_utils.jl:396-420
def ssy_mo_breakdown():
    return np.array([[1.0, 0, -1], [-1, 1, 0]]), np.ones(2), np.ones(3)


def ssy_mo_breakdown2():
    return np.array([[-1.0, 2, 0], [1, -1, 1], [0, 0, -1]]), np.array([1.0, 0, 0]), np.array([1.0, 0, 0])


def ssy_mo_breakdown3():
    return np.array([[-1.0, 1, 0], [3, -1, 0], [0, 1, -1]]), np.array([1.0, 0, 0]), np.array([1.0, 0, 0])
