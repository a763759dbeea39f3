"""CPU restatement of the reference's six Krylov processes (src/krylov_processes.jl), Float64.

TEST INFRASTRUCTURE ONLY -- importable from tests/ (and nothing under krylov.jl_amd/).  Each function follows the
reference loop line by line on the oracle's k* primitives (oracle/krylov_oracle.c: ko_dot, ko_nrm2, ko_axpy;
kdivcopy! is numpy's true division).  Operators are callables `y = A(x)` on host vectors.

Pinned (tests/test_oracle.py::test_processes_*) by the assertions of the reference's own test file
test/test_processes.jl:31-190,194-234 (orthonormality of the leading columns, beta v1 = b, A V_k = V_{k+1} T / H,
A V_k = U B, A' U = V L', the Paige-permuted block relations, exact-breakdown errors with the reference's
messages on its own breakdown matrices `ssy_mo_breakdown*`, test/test_utils.jl:396-420) on the reference's sizes
(n = 500, m = 250, k = 20); not pinned against Julia output (no Julia in this image).
"""
from __future__ import annotations

import numpy as np

import oracle as _o   # oracle/oracle.py (tests put oracle/ on sys.path)


class Breakdown(RuntimeError):
    """`error("Exact breakdown ...")` of src/krylov_processes.jl:64,94,267,289,363,371,383,392."""


def _first(v_out, b, allow_breakdown, what):
    beta = _o.nrm2(b)
    if beta == 0.0:
        if not allow_breakdown:
            raise Breakdown(f"Exact breakdown {what} == 0.")
        v_out[:] = 0.0
    else:
        v_out[:] = b / beta
    return beta


def _normalise(q, s, allow_breakdown, what, it):
    if s == 0.0:
        if not allow_breakdown:
            raise Breakdown(f"Exact breakdown {what} == 0 at iteration i = {it}.")
        q[:] = 0.0
    else:
        q[:] = q / s


def tridiag_pattern(k):
    """colptr / rowval (0-based) of the (k+1) x k tridiagonal T, src/krylov_processes.jl:35-48."""
    colptr = np.zeros(k + 1, dtype=np.int64)
    rowval = np.zeros(3 * k - 1, dtype=np.int64)
    for i in range(1, k + 1):
        pos = colptr[i - 1]
        colptr[i] = 3 * i - 1
        if i == 1:
            rowval[pos:pos + 2] = (0, 1)
        else:
            rowval[pos:pos + 3] = (i - 2, i - 1, i)
    return colptr, rowval


def bidiag_pattern(k):
    """colptr / rowval (0-based) of the (k+1) x (k+1) lower bidiagonal L, src/krylov_processes.jl:331-347."""
    colptr = np.zeros(k + 2, dtype=np.int64)
    rowval = np.zeros(2 * k + 1, dtype=np.int64)
    for i in range(1, k + 2):
        pos = colptr[i - 1]
        if i <= k:
            colptr[i] = pos + 2
            rowval[pos:pos + 2] = (i - 1, i)
        else:
            colptr[i] = pos + 1
            rowval[pos] = i - 1
    return colptr, rowval


def hermitian_lanczos(A, b, k, allow_breakdown=False, reorthogonalization=False):
    """src/krylov_processes.jl:28-102.  Returns V (n x (k+1), Fortran order), beta1, nzval of T (3k-1)."""
    n = b.size
    V = np.zeros((n, k + 1), order="F")
    nz = np.zeros(3 * k - 1)
    beta1 = 0.0
    pa = 0
    for i in range(k):
        vi, q = V[:, i], V[:, i + 1]
        if i == 0:
            beta1 = _first(vi, b, allow_breakdown, "β₁")                    # :60-68
        q[:] = A(np.ascontiguousarray(vi))                                  # :70
        if i >= 1:                                                          # :71-76
            beta_i = nz[pa - 2]
            nz[pa - 1] = beta_i
            _o.axpy(-beta_i, V[:, i - 1], q)
        alpha = _o.dot(vi, q)                                               # :77
        _o.axpy(-alpha, vi, q)                                              # :78
        if reorthogonalization:                                             # :79-89
            if i >= 1:
                bt = _o.dot(V[:, i - 1], q)
                nz[pa - 2] += bt
                nz[pa - 1] += bt
                _o.axpy(-bt, V[:, i - 1], q)
            at = _o.dot(vi, q)
            alpha += at
            _o.axpy(-at, vi, q)
        nz[pa] = alpha                                                      # :90
        beta_next = _o.nrm2(q)                                              # :91
        _normalise(q, beta_next, allow_breakdown, "βᵢ₊₁", i + 1)           # :92-97
        nz[pa + 1] = beta_next                                              # :98
        pa += 3
    return V, beta1, nz


def arnoldi(A, b, k, allow_breakdown=False, reorthogonalization=False):
    """src/krylov_processes.jl:250-296.  Returns V (n x (k+1)), beta, H ((k+1) x k dense)."""
    n = b.size
    V = np.zeros((n, k + 1), order="F")
    H = np.zeros((k + 1, k), order="F")
    beta = 0.0
    for j in range(k):
        vj, q = V[:, j], V[:, j + 1]
        if j == 0:
            beta = _first(vj, b, allow_breakdown, "β")                      # :265-272
        q[:] = A(np.ascontiguousarray(vj))                                  # :274
        for i in range(j + 1):                                              # :275-279
            H[i, j] = _o.dot(V[:, i], q)
            _o.axpy(-H[i, j], V[:, i], q)
        if reorthogonalization:                                             # :280-286
            for i in range(j + 1):
                ht = _o.dot(V[:, i], q)
                _o.axpy(-ht, V[:, i], q)
                H[i, j] += ht
        H[j + 1, j] = _o.nrm2(q)                                            # :287
        _normalise(q, H[j + 1, j], allow_breakdown, "Hᵢ₊₁.ᵢ", j + 1)       # :288-293
    return V, beta, H


def golub_kahan(A, At, b, n, k, allow_breakdown=False):
    """src/krylov_processes.jl:323-398.  A: R^n -> R^m, At its adjoint.  Returns V (n x (k+1)), U (m x (k+1)),
    beta1, nzval of L (2k+1)."""
    m = b.size
    V = np.zeros((n, k + 1), order="F")
    U = np.zeros((m, k + 1), order="F")
    nz = np.zeros(2 * k + 1)
    beta1 = 0.0
    pa = 0
    for i in range(k):
        ui, vi, q, p = U[:, i], V[:, i], U[:, i + 1], V[:, i + 1]
        if i == 0:                                                          # :359-377
            beta1 = _first(ui, b, allow_breakdown, "β₁")
            vi[:] = At(np.ascontiguousarray(ui))
            alpha1 = _o.nrm2(vi)
            if alpha1 == 0.0:
                if not allow_breakdown:
                    raise Breakdown("Exact breakdown α₁ == 0.")
                vi[:] = 0.0
            else:
                vi[:] = vi / alpha1
            nz[pa] = alpha1
        q[:] = A(np.ascontiguousarray(vi))                                  # :378
        alpha = nz[pa]
        _o.axpy(-alpha, ui, q)                                              # :380
        beta_next = _o.nrm2(q)                                              # :381
        _normalise(q, beta_next, allow_breakdown, "βᵢ₊₁", i + 1)           # :382-387
        p[:] = At(np.ascontiguousarray(q))                                  # :388
        _o.axpy(-beta_next, vi, p)                                          # :389
        alpha_next = _o.nrm2(p)                                             # :390
        _normalise(p, alpha_next, allow_breakdown, "αᵢ₊₁", i + 1)          # :391-396
        nz[pa + 1] = beta_next                                              # :397-398
        nz[pa + 2] = alpha_next
        pa += 2
    return V, U, beta1, nz


def nonhermitian_lanczos(A, At, b, c, k, allow_breakdown=False):
    """src/krylov_processes.jl:133-222 (Float64: conj is the identity).  Returns V, beta1, nzval_T, U, gamma1, nzval_Tt."""
    n = b.size
    V = np.zeros((n, k + 1), order="F")
    U = np.zeros((n, k + 1), order="F")
    nt, nh = np.zeros(3 * k - 1), np.zeros(3 * k - 1)
    beta1 = gamma1 = 0.0
    pa = 0
    for i in range(k):
        vi, ui, q, p = V[:, i], U[:, i], V[:, i + 1], U[:, i + 1]
        if i == 0:                                                          # :173-187
            cb = _o.dot(c, b)
            if cb == 0.0:
                if not allow_breakdown:
                    raise Breakdown("Exact breakdown β₁γ₁ == 0.")
                q[:] = 0.0                                                  # v1 / u1 stay as allocated (zeros here)
                p[:] = 0.0
            else:
                beta1 = np.sqrt(abs(cb))
                gamma1 = cb / beta1
                vi[:] = b / beta1
                ui[:] = c / gamma1
        q[:] = A(np.ascontiguousarray(vi))                                  # :188
        p[:] = At(np.ascontiguousarray(ui))                                 # :189
        if i >= 1:                                                          # :190-197
            beta_i, gamma_i = nt[pa - 2], nt[pa - 1]
            _o.axpy(-gamma_i, V[:, i - 1], q)
            _o.axpy(-beta_i, U[:, i - 1], p)
        alpha = _o.dot(ui, q)                                               # :198
        nt[pa] = alpha
        nh[pa] = alpha
        _o.axpy(-alpha, vi, q)                                              # :201
        _o.axpy(-alpha, ui, p)                                              # :202
        pq = _o.dot(p, q)                                                   # :203
        if pq == 0.0:                                                       # :204-209
            if not allow_breakdown:
                raise Breakdown(f"Exact breakdown βᵢ₊₁γᵢ₊₁ == 0 at iteration i = {i + 1}.")
            beta_next = gamma_next = 0.0
            q[:] = 0.0
            p[:] = 0.0
        else:                                                               # :210-215
            beta_next = np.sqrt(abs(pq))
            gamma_next = pq / beta_next
            q[:] = q / beta_next
            p[:] = p / gamma_next
        nt[pa + 1] = beta_next                                              # :216-221
        nh[pa + 1] = gamma_next
        if i + 1 <= k - 1:
            nt[pa + 2] = gamma_next
            nh[pa + 2] = beta_next
        pa += 3
    return V, beta1, nt, U, gamma1, nh


def saunders_simon_yip(A, At, b, c, k, allow_breakdown=False):
    """src/krylov_processes.jl:431-524.  A: R^n -> R^m.  Returns V (m x (k+1)), beta1, nzval_T, U (n x (k+1)), gamma1, nzval_Tt."""
    m, n = b.size, c.size
    V = np.zeros((m, k + 1), order="F")
    U = np.zeros((n, k + 1), order="F")
    nt, nh = np.zeros(3 * k - 1), np.zeros(3 * k - 1)
    beta1 = gamma1 = 0.0
    pa = 0
    for i in range(k):
        vi, ui, q, p = V[:, i], U[:, i], V[:, i + 1], U[:, i + 1]
        if i == 0:                                                          # :470-485
            beta1 = _first(vi, b, allow_breakdown, "β₁")
            gamma1 = _first(ui, c, allow_breakdown, "γ₁ᴴ")
        q[:] = A(np.ascontiguousarray(ui))                                  # :486
        p[:] = At(np.ascontiguousarray(vi))                                 # :487
        if i >= 1:                                                          # :488-495
            beta_i, gamma_i = nt[pa - 2], nt[pa - 1]
            _o.axpy(-gamma_i, V[:, i - 1], q)
            _o.axpy(-beta_i, U[:, i - 1], p)
        alpha = _o.dot(vi, q)                                               # :496
        nt[pa] = alpha
        nh[pa] = alpha
        _o.axpy(-alpha, vi, q)                                              # :499
        _o.axpy(-alpha, ui, p)                                              # :500
        beta_next = _o.nrm2(q)                                              # :501
        _normalise(q, beta_next, allow_breakdown, "βᵢ₊₁", i + 1)
        gamma_next = _o.nrm2(p)                                             # :508
        _normalise(p, gamma_next, allow_breakdown, "γᵢ₊₁", i + 1)
        nt[pa + 1] = beta_next                                              # :515-520
        nh[pa + 1] = gamma_next
        if i + 1 <= k - 1:
            nt[pa + 2] = gamma_next
            nh[pa + 2] = beta_next
        pa += 3
    return V, beta1, nt, U, gamma1, nh


def montoison_orban(A, B, b, c, k, allow_breakdown=False, reorthogonalization=False):
    """src/krylov_processes.jl:553-632.  A: R^n -> R^m, B: R^m -> R^n.  Returns V, beta, H, U, gamma, F."""
    m, n = b.size, c.size
    V = np.zeros((m, k + 1), order="F")
    U = np.zeros((n, k + 1), order="F")
    H = np.zeros((k + 1, k), order="F")
    F = np.zeros((k + 1, k), order="F")
    beta = gamma = 0.0
    for j in range(k):
        vj, uj, q, p = V[:, j], U[:, j], V[:, j + 1], U[:, j + 1]
        if j == 0:                                                          # :571-586
            beta = _first(vj, b, allow_breakdown, "β")
            gamma = _first(uj, c, allow_breakdown, "γ")
        q[:] = A(np.ascontiguousarray(uj))                                  # :587
        p[:] = B(np.ascontiguousarray(vj))                                  # :588
        for i in range(j + 1):                                              # :589-596
            H[i, j] = _o.dot(V[:, i], q)
            _o.axpy(-H[i, j], V[:, i], q)
            F[i, j] = _o.dot(U[:, i], p)
            _o.axpy(-F[i, j], U[:, i], p)
        if reorthogonalization:                                             # :597-607
            for i in range(j + 1):
                ht = _o.dot(V[:, i], q)
                _o.axpy(-ht, V[:, i], q)
                H[i, j] += ht
                ft = _o.dot(U[:, i], p)
                _o.axpy(-ft, U[:, i], p)
                F[i, j] += ft
        H[j + 1, j] = _o.nrm2(q)                                            # :608
        _normalise(q, H[j + 1, j], allow_breakdown, "Hᵢ₊₁.ᵢ", j + 1)
        F[j + 1, j] = _o.nrm2(p)                                            # :615
        _normalise(p, F[j + 1, j], allow_breakdown, "Fᵢ₊₁.ᵢ", j + 1)
    return V, beta, H, U, gamma, F
