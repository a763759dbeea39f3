"""Pins the CPU oracle: (1) the reference's exact known answers (test/test_aux.jl:3-117),
(2) the reference's own C clients compiled from where they lie (only when /root/reference
exists), (3) independent numpy / scipy / LAPACK restatements, (4) the committed golden
vectors under tests/golden/."""
import json
import math
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")


# ---- (1) exact known answers of test/test_aux.jl -------------------------------------

def test_sym_givens_known_answers(oracle):
    g = oracle.sym_givens
    assert g(0.0, 0.0) == (1.0, 0.0, 0.0)
    a = 3.14
    assert g(a, 0.0) == (1.0, 0.0, a)
    assert g(-a, 0.0) == (-1.0, 0.0, a)
    assert g(0.0, a) == (0.0, 1.0, a)
    assert g(0.0, -a) == (0.0, -1.0, a)
    for (x, y) in [(3.0, 4.0), (-4.0, 3.0), (1e-3, -7.0), (5.0, -1e-9)]:
        c, s, r = g(x, y)
        assert abs(c * x + s * y - r) <= 1e-15 * abs(r)
        assert abs(s * x - c * y) <= 1e-15 * abs(r)
        assert abs(c * c + s * s - 1) <= 4e-16


def test_roots_quadratic_known_answers(oracle):
    rq = oracle.roots_quadratic
    assert rq(0.0, 0.0, 0.0) == (0.0, 0.0)
    with pytest.raises(ValueError):
        rq(0.0, 0.0, 1.0)
    assert rq(0.0, 3.14, -1.0) == (1.0 / 3.14, 1.0 / 3.14)
    with pytest.raises(ValueError):
        rq(1.0, 0.0, 1.0)
    assert rq(1.0, 0.0, 0.0) == (0.0, 0.0)
    r = rq(1.0, 3.0, 2.0)
    assert math.isclose(r[0], -2.0) and math.isclose(r[1], -1.0)
    with pytest.raises(ValueError):
        rq(1.0e8, 1.0, 1.0)
    assert rq(-1.0e-8, 1.0e5, 1.0, nitref=0) == (1.0e13, 0.0)
    assert rq(-1.0e-8, 1.0e5, 1.0, nitref=1) == (1.0e13, -1.0e-05)
    for nit in (0, 1):
        r = rq(-1.0e-7, 1.0, 1.0, nitref=nit)
        assert math.isclose(r[0], 1.0e7, rel_tol=1e-6) and math.isclose(r[1], -1.0, rel_tol=1e-6)


def test_to_boundary_known_answers(oracle):
    n = 5
    x = np.ones(n)
    d = np.ones(n)
    d[0:n:2] = -1
    for bad in (-1.0, 0.5):
        with pytest.raises(ValueError):
            oracle.to_boundary(x, d, bad)
    with pytest.raises(ValueError):
        oracle.to_boundary(x, np.zeros(n), 1.0)
    r = oracle.to_boundary(x, d, 5.0)
    assert math.isclose(max(r), 2.209975124224178, rel_tol=1e-14)
    assert math.isclose(min(r), -1.8099751242241782, rel_tol=1e-14)
    r = oracle.to_boundary(x, d, 5.0, flip=True)
    assert math.isclose(max(r), 1.8099751242241782, rel_tol=1e-14)
    assert math.isclose(min(r), -2.209975124224178, rel_tol=1e-14)


# ---- (2) the reference's own C clients ------------------------------------------------

OUT_OF_SCOPE = ("MINRES", "Float32", "DQGMRES", "block_minres")


@pytest.fixture(scope="module")
def refbin():
    if not os.path.isdir(os.path.join(REF, "interfaces")):
        pytest.skip("reference tree absent (GPU box): _ref cannot be (re)built here")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all", "ref"])
    return os.path.join(ROOT, "oracle", "_ref")


def test_ref_basic_cg_example(refbin):
    # interfaces/examples/C/basic_cg.c:13-15 : niter 3, x = ones
    out = subprocess.run([os.path.join(refbin, "basic_cg")], capture_output=True, text=True)
    assert out.returncode == 0
    assert "Solved: yes" in out.stdout and "niter: 3" in out.stdout
    assert "x = [ 1.00 1.00 1.00 1.00 1.00 ]" in out.stdout


def test_ref_test_api_client(refbin):
    out = subprocess.run([os.path.join(refbin, "test_api")], capture_output=True, text=True)
    fails = [l for l in out.stdout.splitlines() if "FAIL" in l]
    bad = [l for l in fails if not any(k in l for k in OUT_OF_SCOPE)]
    assert not bad, bad
    assert len(fails) == 5          # 2 MINRES + 3 Float32 checks: out-of-scope solvers / dtypes
    assert "36 checks passed" in out.stdout


def test_ref_test_all_solvers_client(refbin):
    out = subprocess.run([os.path.join(refbin, "test_all_solvers")], capture_output=True, text=True)
    lines = {l.split()[0]: l for l in out.stdout.splitlines() if "..." in l}
    for s in ("cg", "gmres", "bicgstab"):
        assert "PASS" in lines[s], lines[s]
    others = [k for k, l in lines.items() if "PASS" not in l]
    assert all("returned -2" in lines[k] for k in others)   # unknown (solver,dtype) pair


def test_ref_test_block_client(refbin):
    out = subprocess.run([os.path.join(refbin, "test_block")], capture_output=True, text=True)
    sections, cur = {}, None
    for l in out.stdout.splitlines():
        if l.endswith("..."):
            cur = l
            sections[cur] = []
        elif "FAIL" in l:
            sections[cur].append(l)
    for name, fails in sections.items():
        if "block_minres" in name:
            continue
        assert not fails, (name, fails)
    ex = subprocess.run([os.path.join(refbin, "block_gmres_example")], capture_output=True, text=True)
    assert "Block solved: yes" in ex.stdout


# ---- (3) independent restatements ------------------------------------------------------

def _julia_get_div_grad(n1, n2, n3):
    """test/get_div_grad.jl:8-25 restated with scipy.sparse.kron (independent of the C stencil code)."""
    import scipy.sparse as sp

    def ddx(n):
        e = np.ones(n)
        return sp.csr_matrix((np.r_[-e, e], (np.r_[0:n, 0:n], np.r_[0:n, 1:n + 1])), shape=(n, n + 1))
    eye = sp.identity
    D1 = sp.kron(eye(n3), sp.kron(eye(n2), ddx(n1)))
    D2 = sp.kron(eye(n3), sp.kron(ddx(n2), eye(n1)))
    D3 = sp.kron(ddx(n3), sp.kron(eye(n2), eye(n1)))
    Div = sp.hstack([D1, D2, D3]).tocsr()
    return (Div @ Div.T).tocsr()


def _julia_kron_unsymmetric(n):
    """test/test_utils.jl:160-169."""
    import scipy.sparse as sp
    A = sp.diags([-np.ones(n - 1), 3.0 * np.ones(n), -2.0 * np.ones(n - 1)], [-1, 0, 1]).tocsr()
    Id = sp.identity(n)
    A = sp.kron(A, Id) + sp.kron(Id, A)
    A = sp.kron(A, Id) + sp.kron(Id, A)
    return A.tocsr()


@pytest.mark.parametrize("dims", [(2, 2, 2), (4, 4, 4), (3, 5, 4), (6, 6, 6)])
def test_poisson_generator_matches_julia_construction(oracle, dims):
    A = oracle.poisson3d(*dims)
    ref = _julia_get_div_grad(*dims)
    ref.eliminate_zeros()
    ref.sort_indices()
    S = A.to_scipy()
    assert A.nnz == ref.nnz
    assert (S != ref).nnz == 0
    n1 = dims[0]
    if dims[0] == dims[1] == dims[2]:
        assert A.nnz == 7 * n1 ** 3 - 6 * n1 ** 2


@pytest.mark.parametrize("n", [2, 3, 5])
def test_kron_unsymmetric_generator_matches_julia_construction(oracle, n):
    A = oracle.kron_unsymmetric(n)
    ref = _julia_kron_unsymmetric(n)
    ref.sort_indices()
    assert (A.to_scipy() != ref).nnz == 0


def test_spmv_and_blas1_against_numpy(oracle):
    rng = np.random.default_rng(7)
    A = oracle.poisson3d(9)
    x = rng.standard_normal(A.n)
    y = A.matvec(x)
    assert np.allclose(y, A.to_scipy() @ x, rtol=0, atol=1e-13)
    a, b = rng.standard_normal(1000), rng.standard_normal(1000)
    exact = math.fsum(a * b)   # products rounded, sum exact: differs from the long-double dot by O(eps)
    assert abs(oracle.dot(a, b) - exact) <= 1e-13 * np.abs(a * b).sum()
    assert math.isclose(oracle.nrm2(a), float(np.linalg.norm(a)), rel_tol=1e-14)
    y0 = b.copy()
    oracle.axpy(0.3, a, y0)
    assert np.allclose(y0, b + 0.3 * a, rtol=1e-15, atol=1e-15)
    y0 = b.copy()
    oracle.axpby(0.3, a, -1.7, y0)
    assert np.allclose(y0, 0.3 * a - 1.7 * b, rtol=1e-15, atol=1e-15)


def _numpy_cg(A, b, atol, rtol, itmax):
    """Independent numpy restatement of src/cg.jl:153-268 (M = I, radius = 0)."""
    n = b.size
    x = np.zeros(n)
    r = b.copy()
    p = r.copy()
    gamma = float(r @ r)
    hist = [math.sqrt(gamma)]
    eps_tol = atol + rtol * hist[0]
    it = 0
    solved = hist[0] <= eps_tol
    while not (solved or it >= itmax):
        Ap = A @ p
        alpha = gamma / float(p @ Ap)
        x += alpha * p
        r -= alpha * Ap
        gn = float(r @ r)
        hist.append(math.sqrt(gn))
        solved = hist[-1] <= eps_tol
        if not solved:
            beta = gn / gamma
            gamma = gn
            p = r + beta * p
        it += 1
    return x, it, np.array(hist)


@pytest.mark.parametrize("n1,expect", [(16, 38), (32, 78)])
def test_cg_against_numpy_restatement(oracle, n1, expect):
    A = oracle.poisson3d(n1)
    b = np.ones(A.n)
    res = oracle.cg(A, b, history=True)
    x, it, hist = _numpy_cg(A.to_scipy(), b, math.sqrt(np.finfo(float).eps), math.sqrt(np.finfo(float).eps), 2 * A.n)
    assert res.solved and res.status == "solution good enough given atol and rtol"
    assert res.niter == it == expect            # SURVEY section 8c session values
    assert np.allclose(res.residuals, hist, rtol=1e-9)
    assert np.linalg.norm(b - A.to_scipy() @ res.x) / np.linalg.norm(b) <= 1e-6   # test/test_cg.jl:22-28


def test_cg_benchmark_tolerances(oracle):
    # benchmark/benchmarks.jl:14-21 : atol=0, rtol=1e-8, itmax=n ; SURVEY 8c: 39 / 79 iterations
    for n1, expect in ((16, 39), (32, 79)):
        A = oracle.poisson3d(n1)
        res = oracle.cg(A, np.ones(A.n), atol=0.0, rtol=1e-8, itmax=A.n)
        assert res.niter == expect and res.solved


def test_cg_edge_cases(oracle):
    A = oracle.tridiag(10, -1.0, 4.0, -1.0)                         # symmetric_definite, test_utils.jl:18-23
    b = A.matvec(np.arange(1.0, 11.0))
    res = oracle.cg(A, b, itmax=10)
    assert res.solved and np.linalg.norm(b - A.matvec(res.x)) / np.linalg.norm(b) <= 1e-6
    res = oracle.cg(A, np.zeros(10))                                 # test/test_cg.jl zero rhs
    assert res.niter == 0 and res.status == "x is a zero-residual solution" and np.all(res.x == 0)
    res = oracle.cg(A, b, itmax=2)
    assert (not res.solved) and res.status == "maximum number of iterations exceeded" and res.niter == 2
    # warm start from the solution: zero iterations needed beyond tolerance check
    x_star = oracle.cg(A, b, atol=1e-14, rtol=1e-14).x
    res = oracle.cg(A, b, x0=x_star)
    assert res.solved and res.niter <= 1
    # negative-definite operator: linesearch flags nonpositive curvature at iteration 0
    Aneg = oracle.tridiag(10, 1.0, -4.0, 1.0)
    res = oracle.cg(Aneg, b, linesearch=True)
    assert res.niter == 0 and res.indefinite and res.npcCount == 1 and res.status == "nonpositive curvature"
    # trust region: lands on the boundary
    res = oracle.cg(A, b, radius=1.0)
    assert res.status == "on trust-region boundary" and math.isclose(np.linalg.norm(res.x), 1.0, rel_tol=1e-10)


def test_gmres_against_scipy_and_flags(oracle):
    import scipy.sparse.linalg as spla
    A = oracle.kron_unsymmetric(8)
    S = A.to_scipy()
    b = S @ np.ones(A.n)
    for kw in ({}, {"restart": True}, {"restart": True, "reorthogonalization": True}):
        res = oracle.gmres(A, b, memory=10, history=True, **kw)
        assert res.solved, kw
        assert np.linalg.norm(b - S @ res.x) / np.linalg.norm(b) <= 1e-6
        assert res.niter > 10
        assert np.all(np.diff(res.residuals) <= 1e-12)               # GMRES residual estimates are monotone
    # Jacobi preconditioners as callables (test/test_gmres.jl:105-128)
    d = S.diagonal()
    res = oracle.gmres(A, b, M=lambda v: v / d, memory=10)
    assert res.solved and np.linalg.norm(b - S @ res.x) / np.linalg.norm(b) <= 1e-6
    res = oracle.gmres(A, b, N=lambda v: v / d, memory=10, restart=True)
    assert res.solved and np.linalg.norm(b - S @ res.x) / np.linalg.norm(b) <= 1e-6
    res = oracle.gmres(A, np.zeros(A.n))
    assert res.niter == 0 and res.status == "x is a zero-residual solution"
    # full GMRES residual estimate equals the true residual norm of scipy's least-squares iterate
    res = oracle.gmres(A, b, memory=20, atol=1e-10, rtol=1e-10, history=True)
    xs, info = spla.gmres(S, b, rtol=1e-10, atol=1e-10, restart=A.n)
    assert info == 0 and np.allclose(res.x, xs, atol=1e-7)


def test_bicgstab_flags(oracle):
    A = oracle.kron_unsymmetric(8)
    S = A.to_scipy()
    b = S @ np.ones(A.n)
    res = oracle.bicgstab(A, b, history=True)
    assert res.solved and np.linalg.norm(b - S @ res.x) / np.linalg.norm(b) <= 1e-6
    assert len(res.residuals) == res.niter + 1
    res = oracle.bicgstab(A, np.zeros(A.n))
    assert res.niter == 0 and res.status == "x is a zero-residual solution"
    c = np.zeros(A.n)
    res = oracle.bicgstab(A, b, c=c)
    assert res.status == "Breakdown bᴴc = 0" and not res.solved
    d = S.diagonal()
    res = oracle.bicgstab(A, b, M=lambda v: v / d, N=lambda v: v.copy())
    assert res.solved and np.linalg.norm(b - S @ res.x) / np.linalg.norm(b) <= 1e-6


def test_householder_against_lapack(oracle):
    import ctypes as C
    from scipy.linalg import lapack
    rng = np.random.default_rng(3)
    L = oracle.lib()
    for (m, k) in [(40, 6), (32, 16), (7, 7)]:
        A = np.asfortranarray(rng.standard_normal((m, k)))
        Q = A.copy(order="F")
        R = np.zeros((k, k), order="F")
        tau = np.zeros(k)
        L.ko_householder(m, k, oracle._dp(Q), oracle._dp(R), oracle._dp(tau), 0)
        qr, tau_l, _, info = lapack.dgeqrf(A)
        assert info == 0
        assert np.allclose(tau, tau_l, rtol=1e-13, atol=1e-15)
        assert np.allclose(R, np.triu(qr[:k, :]), rtol=1e-12, atol=1e-13)
        q_l, _, info = lapack.dorgqr(qr, tau_l)
        assert np.allclose(Q, q_l, rtol=1e-12, atol=1e-13)
        assert np.allclose(Q @ R, A, atol=1e-12) and np.allclose(Q.T @ Q, np.eye(k), atol=1e-13)
        # ormqr side L trans T
        Cm = np.asfortranarray(rng.standard_normal((m, 3)))
        C1 = Cm.copy(order="F")
        qrc = np.asfortranarray(qr)
        L.ko_ormqr_LT(m, 3, k, oracle._dp(qrc), m, oracle._dp(tau_l), oracle._dp(C1), m)
        c_l, _, info = lapack.dormqr("L", "T", qr, tau_l, Cm, max(1, 3 * 64))
        assert np.allclose(C1, c_l, rtol=1e-12, atol=1e-13)


def test_block_gmres_oracle(oracle):
    # interfaces/test/C/test_block.c:62-70 style right-hand sides: B = A * X_true, X_true[i,j] = ((i+1)/n)^j
    A = oracle.kron_unsymmetric(6)
    S = A.to_scipy()
    n, p = A.n, 4
    Xt = np.stack([((np.arange(n) + 1.0) / n) ** j for j in range(p)], axis=1)
    B = S @ Xt
    for kw in ({}, {"restart": True}, {"reorthogonalization": True}):
        res = oracle.block_gmres(A, B, memory=8, history=True, **kw)
        assert res.solved, kw
        assert np.linalg.norm(B - S @ res.x) / np.linalg.norm(B) <= 1e-6
        assert math.isclose(res.residuals[0], np.linalg.norm(B), rel_tol=1e-13)


# ---- (4) committed golden vectors ------------------------------------------------------

def test_golden_vectors(oracle):
    path = os.path.join(GOLD, "oracle_histories.json")
    if not os.path.exists(path):
        pytest.skip("golden vectors not generated yet")
    gold = json.load(open(path))
    for case in gold["cases"]:
        A = getattr(oracle, case["matrix"])(case["n1"])
        if case["rhs"] == "ones":
            b = np.ones(A.n)
        else:
            b = A.matvec(np.ones(A.n))
        res = getattr(oracle, case["solver"])(A, b, history=True, **case["kwargs"])
        assert res.niter == case["niter"], case["name"]
        assert res.status == case["status"]
        assert np.allclose(res.residuals, np.array(case["residuals"]), rtol=1e-12, atol=0), case["name"]


def _fortran_sections(stdout):
    sections, cur = {}, None
    for l in stdout.splitlines():
        if l.rstrip().endswith("...") and "FAIL" not in l and "PASS" not in l:
            cur = l.strip()
            sections[cur] = []
        elif "FAIL" in l and cur:
            sections[cur].append(l)
    return sections


def test_ref_fortran_clients(refbin):
    """interfaces/test/Fortran/{test_all_solvers,test_block}.f90, compiled from where they lie with amdflang
    (krylov.f90 is the reference's own include file): cg, gmres, bicgstab PASS; every block_gmres section passes."""
    exe = os.path.join(refbin, "f_test_all_solvers")
    if not os.path.exists(exe):
        pytest.skip("no Fortran compiler in this image")
    out = subprocess.run([exe], capture_output=True, text=True)
    lines = {l.split()[0]: l for l in out.stdout.splitlines() if "..." in l}
    for s in ("cg", "gmres", "bicgstab"):
        assert "PASS" in lines[s], lines[s]
    others = [k for k, l in lines.items() if "PASS" not in l]
    assert all("returned -2" in lines[k] for k in others)   # solvers outside the hot path: unknown to the shim
    out = subprocess.run([os.path.join(refbin, "f_test_block")], capture_output=True, text=True)
    sections = _fortran_sections(out.stdout)
    assert any("block_gmres" in k for k in sections), out.stdout
    for name, fails in sections.items():
        if "block_minres" in name:
            continue
        assert not fails, (name, fails)


# ---- ILU(0) / IC(0) (the vendor ic02 / ilu02 of the reference's GPU recipes) ------------

def test_ilu0_defining_property_and_reference_known_answer(oracle):
    """(L U)_ij = a_ij on the pattern of A (the definition of ILU(0)); for SPD A: U = D L^T (IC(0)); and the
    reference's only preconditioned-GPU known answer: IC(0)-CG on sparse_laplacian(16) converges in <= 19
    iterations with ||b - A x|| <= 1e-6 (test/gpu/nvidia.jl:37-70)."""
    A = oracle.poisson3d(6)
    P = oracle.Ilu0(A)
    n = A.n
    L, U = np.eye(n), np.zeros((n, n))
    for i in range(n):
        for q in range(A.rowptr[i], A.rowptr[i + 1]):
            (L if A.col[q] < i else U)[i, A.col[q]] = P.lu[q]
    S = A.to_scipy().toarray()
    assert np.abs((L @ U - S)[S != 0]).max() <= 1e-14
    D = np.diag(np.diag(U))
    assert np.allclose(U, D @ L.T, atol=1e-14)                         # SPD: ILU(0) == IC(0) up to the diagonal scaling
    x = np.linspace(1.0, 2.0, n)
    assert np.allclose(L @ U @ P.solve(x), x, atol=1e-12)
    # unsymmetric pattern with unsymmetric values
    B = oracle.kron_unsymmetric(4)
    Q = oracle.Ilu0(B)
    n = B.n
    L, U = np.eye(n), np.zeros((n, n))
    for i in range(n):
        for q in range(B.rowptr[i], B.rowptr[i + 1]):
            (L if B.col[q] < i else U)[i, B.col[q]] = Q.lu[q]
    S = B.to_scipy().toarray()
    assert np.abs((L @ U - S)[S != 0]).max() <= 1e-13
    # known answer of the reference
    A16 = oracle.poisson3d(16)
    P16 = oracle.Ilu0(A16)
    b = np.ones(A16.n)
    r = oracle.cg(A16, b, M=lambda v: P16.solve(v))
    assert r.solved and r.niter <= 19 and np.linalg.norm(b - A16.matvec(r.x)) <= 1e-6
    assert oracle.cg(A16, b).niter == 38                                   # unpreconditioned, for scale
    # zero pivot is reported, not divided by
    Z = oracle.tridiag(4, 1.0, 0.0, 1.0)
    with pytest.raises(ZeroDivisionError):
        oracle.Ilu0(Z)


# ---- Krylov processes: the assertions of the reference's own test/test_processes.jl -------------------

def _approx(a, b):          # Julia's `≈` for arrays: norm(a - b) <= sqrt(eps) * max(norm(a), norm(b))
    return np.linalg.norm(a - b) <= math.sqrt(np.finfo(float).eps) * max(np.linalg.norm(a), np.linalg.norm(b))


@pytest.mark.parametrize("reorth", [False, True])
def test_processes_hermitian_lanczos_and_arnoldi_reference_assertions(oracle, reorth):
    """test/test_processes.jl:31-49 (Hermitian Lanczos) and :76-95 (Arnoldi), Float64, n = 500, k = 20, s = 5."""
    import oracle_processes as P
    import scipy.sparse as sp
    n, k, s = 500, 20, 5
    rng = np.random.default_rng(1)
    A = rng.random((n, n))
    A = A.T @ A
    b = rng.random(n)
    V, beta1, nz = P.hermitian_lanczos(lambda x: A @ x, b, k, reorthogonalization=reorth)
    colptr, rowval = P.tridiag_pattern(k)
    T = sp.csc_matrix((nz, rowval, colptr), shape=(k + 1, k)).toarray()
    assert np.linalg.norm(V[:, :s].T @ V[:, :s] - np.eye(s)) <= 1e-4
    assert _approx(beta1 * V[:, 0], b)
    assert _approx(A @ V[:, :k], V @ T)
    assert np.array_equal(np.diag(T, 1), np.diag(T, -1)[:k - 1])          # symmetric tridiagonal by construction

    A = rng.random((n, n))
    V, beta, H = P.arnoldi(lambda x: A @ x, b, k, reorthogonalization=reorth)
    assert np.linalg.norm(V[:, :s].T @ V[:, :s] - np.eye(s)) <= 1e-4
    assert _approx(beta * V[:, 0], b)
    assert _approx(A @ V[:, :k], V @ H)
    assert np.all(np.tril(H, -2) == 0.0)


def test_processes_golub_kahan_reference_assertions(oracle):
    """test/test_processes.jl:98-118, Float64, m = 250, n = 500, k = 20."""
    import oracle_processes as P
    import scipy.sparse as sp
    m, n, k, s = 250, 500, 20, 5
    rng = np.random.default_rng(2)
    A = rng.random((m, n))
    b = rng.random(m)
    V, U, beta1, nz = P.golub_kahan(lambda x: A @ x, lambda y: A.T @ y, b, n, k)
    colptr, rowval = P.bidiag_pattern(k)
    L = sp.csc_matrix((nz, rowval, colptr), shape=(k + 1, k + 1)).toarray()
    B = L[:k + 1, :k]
    assert np.linalg.norm(V[:, :s].T @ V[:, :s] - np.eye(s)) <= 1e-4
    assert np.linalg.norm(U[:, :s].T @ U[:, :s] - np.eye(s)) <= 1e-4
    assert _approx(beta1 * U[:, 0], b)
    assert _approx(A @ V[:, :k], U @ B)
    assert _approx(A.T @ U, V @ L.T)
    assert _approx(A.T @ A @ V[:, :k], V @ L.T @ B)
    assert _approx(A @ A.T @ U[:, :k], U @ B @ L[:k, :k].T)


def test_processes_exact_breakdown_messages(oracle):
    """test/test_processes.jl:194-218: A0 = I (2 x 2), b0 = 0 -- the error text and the allow_breakdown path."""
    import oracle_processes as P
    b0 = np.zeros(2)
    A0 = lambda x: x.copy()
    with pytest.raises(P.Breakdown, match="Exact breakdown β₁ == 0."):
        P.hermitian_lanczos(A0, b0, 2)
    P.hermitian_lanczos(A0, b0, 2, allow_breakdown=True)
    with pytest.raises(P.Breakdown, match="Exact breakdown β == 0."):
        P.arnoldi(A0, b0, 2)
    V, beta, H = P.arnoldi(A0, b0, 2, allow_breakdown=True)
    assert beta == 0.0 and not V.any() and not H.any()
    with pytest.raises(P.Breakdown, match="Exact breakdown β₁ == 0."):
        P.golub_kahan(A0, A0, b0, 2, 2)
    P.golub_kahan(A0, A0, b0, 2, 2, allow_breakdown=True)
    # beyond the reference's cases: an invariant subspace after one step (A = I, b = e1)
    with pytest.raises(P.Breakdown, match="Exact breakdown βᵢ₊₁ == 0 at iteration i = 1."):
        P.hermitian_lanczos(A0, np.array([1.0, 0.0]), 2)
    with pytest.raises(P.Breakdown, match="Exact breakdown Hᵢ₊₁.ᵢ == 0 at iteration i = 1."):
        P.arnoldi(A0, np.array([1.0, 0.0]), 2)


def _tri(P, k, nz):
    import scipy.sparse as sp
    colptr, rowval = P.tridiag_pattern(k)
    return sp.csc_matrix((nz, rowval, colptr), shape=(k + 1, k)).toarray()


def test_processes_two_sided_reference_assertions(oracle):
    """test/test_processes.jl:53-74 (non-Hermitian Lanczos), :120-152 (Saunders-Simon-Yip), :147-190 (Montoison-Orban)."""
    import oracle_processes as P
    import process_checks as pc
    m, n, k = 250, 500, 20
    rng = np.random.default_rng(4)
    A = rng.random((n, n)); b = rng.random(n); c = rng.random(n)
    V, b1, nt, U, g1, nh = P.nonhermitian_lanczos(lambda x: A @ x, lambda y: A.T @ y, b, c, k)
    pc.check_nonhermitian_lanczos(A, b, c, k, V, b1, _tri(P, k, nt), U, g1, _tri(P, k, nh))

    A = rng.random((m, n)); b = rng.random(m); c = rng.random(n)
    V, b1, nt, U, g1, nh = P.saunders_simon_yip(lambda x: A @ x, lambda y: A.T @ y, b, c, k)
    pc.check_saunders_simon_yip(A, b, c, k, V, b1, _tri(P, k, nt), U, g1, _tri(P, k, nh))

    B = rng.random((n, m))
    for reorth in (False, True):
        V, beta, H, U, gamma, F = P.montoison_orban(lambda x: A @ x, lambda y: B @ y, b, c, k, reorthogonalization=reorth)
        pc.check_montoison_orban(A, B, b, c, k, V, beta, H, U, gamma, F)


def test_processes_two_sided_breakdowns(oracle):
    """test/test_processes.jl:205-234 on A0 = I and the reference's ssy_mo_breakdown{,2,3} matrices."""
    import oracle_processes as P
    import process_checks as pc
    op = lambda M: (lambda x: M @ x)
    A0, b0, c0 = np.eye(2), np.zeros(2), np.ones(2)
    A1, b1, c1 = pc.ssy_mo_breakdown()
    A2, b2, c2 = pc.ssy_mo_breakdown2()
    A3, b3, c3 = pc.ssy_mo_breakdown3()

    def raises(msg, fn, *a, **kw):
        with pytest.raises(P.Breakdown) as e:
            fn(*a, **kw)
        assert str(e.value) == msg

    raises("Exact breakdown β₁γ₁ == 0.", P.nonhermitian_lanczos, op(A0), op(A0.T), b0, c0, 2)
    P.nonhermitian_lanczos(op(A0), op(A0.T), b0, c0, 2, allow_breakdown=True)
    raises("Exact breakdown β₁ == 0.", P.saunders_simon_yip, op(A0), op(A0.T), b0, c0, 2)
    P.saunders_simon_yip(op(A0), op(A0.T), b0, c0, 2, allow_breakdown=True)
    raises("Exact breakdown γ₁ᴴ == 0.", P.saunders_simon_yip, op(A0), op(A0.T), c0, b0, 2)
    P.saunders_simon_yip(op(A0), op(A0.T), c0, b0, 2, allow_breakdown=True)
    raises("Exact breakdown βᵢ₊₁ == 0 at iteration i = 1.", P.saunders_simon_yip, op(A1), op(A1.T), b1, c1, 1)
    raises("Exact breakdown βᵢ₊₁ == 0 at iteration i = 2.", P.saunders_simon_yip, op(A2), op(A2.T), b2, c2, 2)
    raises("Exact breakdown γᵢ₊₁ == 0 at iteration i = 2.", P.saunders_simon_yip, op(A3), op(A3.T), b3, c3, 2)
    raises("Exact breakdown β == 0.", P.montoison_orban, op(A0), op(A0.T), b0, c0, 2)
    P.montoison_orban(op(A0), op(A0.T), b0, c0, 2, allow_breakdown=True)
    raises("Exact breakdown γ == 0.", P.montoison_orban, op(A0), op(A0.T), c0, b0, 2)
    P.montoison_orban(op(A0), op(A0.T), c0, b0, 2, allow_breakdown=True)
    raises("Exact breakdown Hᵢ₊₁.ᵢ == 0 at iteration i = 1.", P.montoison_orban, op(A1), op(A1.T), b1, c1, 1)
    raises("Exact breakdown Hᵢ₊₁.ᵢ == 0 at iteration i = 2.", P.montoison_orban, op(A2), op(A2.T), b2, c2, 2)
    raises("Exact breakdown Fᵢ₊₁.ᵢ == 0 at iteration i = 2.", P.montoison_orban, op(A3), op(A3.T), b3, c3, 2)


def test_processes_single_step_shapes(oracle):
    """k = 1: the smallest T (2 entries), L (3 entries) and H (2 x 1) of the reference's storage, and the defining
    relations on them."""
    import oracle_processes as P
    rng = np.random.default_rng(6)
    n = 40
    A = rng.random((n, n)); S = A + A.T; b = rng.random(n); c = rng.random(n)
    V, beta, nz = P.hermitian_lanczos(lambda x: S @ x, b, 1)
    assert nz.shape == (2,) and V.shape == (n, 2)
    assert np.allclose(S @ V[:, 0], nz[0] * V[:, 0] + nz[1] * V[:, 1])
    V, beta, H = P.arnoldi(lambda x: A @ x, b, 1)
    assert H.shape == (2, 1) and np.allclose(A @ V[:, 0], V @ H[:, 0])
    V, U, beta, nz = P.golub_kahan(lambda x: A @ x, lambda y: A.T @ y, b, n, 1)
    assert nz.shape == (3,)                                  # alpha1, beta2, alpha2
    assert np.allclose(A @ V[:, 0], nz[0] * U[:, 0] + nz[1] * U[:, 1])
    assert np.allclose(A.T @ U[:, 1], nz[1] * V[:, 0] + nz[2] * V[:, 1])
    V, b1, nt, U, g1, nh = P.nonhermitian_lanczos(lambda x: A @ x, lambda y: A.T @ y, b, c, 1)
    assert nt.shape == (2,) and nh.shape == (2,) and nt[0] == nh[0]
    assert np.allclose(A @ V[:, 0], nt[0] * V[:, 0] + nt[1] * V[:, 1])
    assert np.allclose(A.T @ U[:, 0], nh[0] * U[:, 0] + nh[1] * U[:, 1])


def test_binary128_reference_distances_are_reproducible(oracle):
    """tests/golden/quad_histories.json stores, per case, the binary128 history of the oracle's recurrence and the distance
    of the double-precision oracle to it.  Re-derive that distance for the small cases with the oracle as built here:
    the stored numbers are properties of the algorithm + IEEE double, not of a particular run."""
    ok = oracle
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "quad_histories.json")))
    seen = 0
    for c in g["cases"]:
        if c["n1"] > 16 or c["solver"] == "block_gmres":
            continue
        A = getattr(ok, c["matrix"])(c["n1"])
        b = np.ones(A.n) if c["rhs"] == "ones" else A.matvec(np.ones(A.n))
        kw = dict(restart=bool(c.get("restart", False)), reorthogonalization=bool(c.get("reorthogonalization", False)))
        if c["solver"] == "cg":
            ref = ok.cg(A, b, history=True)
        elif c["solver"] == "bicgstab":
            ref = ok.bicgstab(A, b, history=True)
        else:
            ref = ok.gmres(A, b, memory=c["memory"], history=True, **kw)
        hq = np.array(c["residuals"])
        assert ref.niter == c["niter"] and ref.status == c["status"]
        dev = float(np.max(np.abs(ref.residuals - hq) / hq))
        assert abs(dev - c["oracle_double_max_rel_dev"]) <= 1e-3 * c["oracle_double_max_rel_dev"] + 1e-18, c["name"]
        seen += 1
    assert seen >= 6


def test_parallel_matvec_changes_no_value(oracle):
    """tests/golden/make_scale_golden.py runs the oracle with its operator products (and the independent entries of its panel
    products) spread over the host cores.  Rows are independent: histories and solutions must equal the serial run bit for bit."""
    ok = oracle
    A = ok.poisson3d(12)
    b = np.ones(A.n)
    B = ok.kron_unsymmetric(8)
    bb = B.matvec(np.ones(B.n))
    t = (np.arange(B.n) + 1.0) / B.n
    Bk = B.to_scipy() @ np.stack([t ** j for j in range(4)], axis=1)
    serial = (ok.cg(A, b, history=True), ok.gmres(B, bb, memory=10, restart=True, history=True),
              ok.block_gmres(B, Bk, memory=6, restart=True, history=True))
    ok.PARALLEL_MATVEC = True
    ok.lib().ko_set_threads(4)
    try:
        par = (ok.cg(A, b, history=True), ok.gmres(B, bb, memory=10, restart=True, history=True),
               ok.block_gmres(B, Bk, memory=6, restart=True, history=True))
    finally:
        ok.PARALLEL_MATVEC = False
        ok.lib().ko_set_threads(1)
    for s_, p_ in zip(serial, par):
        assert s_.niter == p_.niter and np.array_equal(s_.residuals, p_.residuals) and np.array_equal(s_.x, p_.x)


def test_matrix_free_stencil_oracle_equals_the_csr_oracle_bit_for_bit(oracle):
    """The cfg-4 oracle: ko_cg on the MATRIX-FREE get_div_grad (ko_stencil7_matvec, the 7-point product
    from the grid indices; test/get_div_grad.jl:8-25) -- the CSR arrays of 1024^3 would be 94 GB.  It must be the CSR oracle
    bit for bit: products on cubic and non-cubic grids (incl. degenerate 1-wide ones), then whole cg! histories, solutions and
    status at 32^3 and 64^3, serial and threaded."""
    ok = oracle
    rng = np.random.default_rng(7)
    for dims in ((1, 1, 1), (1, 5, 1), (4, 1, 3), (5, 7, 9), (16, 16, 16), (3, 33, 2)):
        A = ok.poisson3d(*dims)
        S = ok.Stencil7(*dims)
        for x in (rng.standard_normal(A.n), np.ones(A.n), -np.abs(rng.standard_normal(A.n))):
            y = A.matvec(x)
            assert np.array_equal(y, S.matvec(x)) and np.array_equal(np.signbit(y), np.signbit(S.matvec(x))), dims
            assert np.array_equal(y, S.matvec(x, parallel=True)), dims
    for threads in (1, 4):
        ok.lib().ko_set_threads(threads)
        try:
            for n1, kw in ((32, dict(atol=0.0, rtol=0.0, itmax=70)), (32, {}), (64, dict(atol=0.0, rtol=1e-8, itmax=64 ** 3))):
                A = ok.poisson3d(n1)
                ref = ok.cg(A, np.ones(A.n), history=True, **kw)
                idx = np.linspace(0, A.n - 1, 64).astype(np.int64)
                got = ok.cg_stencil7(n1, x_index=idx, history=True, **kw)
                assert got.rc == 0 and got.niter == ref.niter and got.status == ref.status and got.solved == ref.solved
                assert np.array_equal(got.residuals, ref.residuals), (n1, threads)
                assert np.array_equal(got.x, ref.x[idx]), (n1, threads)
        finally:
            ok.lib().ko_set_threads(1)


def test_dot2_mode_of_the_oracle_is_exact_and_changes_only_the_dots(oracle):
    """ko_set_dot_mode(1) (the yardstick for the documented oracle at sizes where binary128 is out of reach): Dot2 equals the
    exactly rounded dot (rational arithmetic) also where the sequential sum does not, whatever the thread count; cg! histories
    with it stay within 1e-13 of the documented ones at a small size; the default mode is restored."""
    from fractions import Fraction
    ok = oracle
    L = ok.lib()
    rng = np.random.default_rng(3)
    x = rng.standard_normal(30000) * 10.0 ** rng.integers(-8, 8, 30000)
    y = rng.standard_normal(30000) * 10.0 ** rng.integers(-8, 8, 30000)
    exact = float(sum(Fraction(a) * Fraction(b) for a, b in zip(x.tolist(), y.tolist())))
    assert L.ko_get_dot_mode() == 0
    try:
        L.ko_set_dot_mode(1)
        for th in (1, 3, 8):
            L.ko_set_threads(th)
            assert ok.dot(x, y) == exact
        L.ko_set_threads(1)
        A = ok.poisson3d(20)
        r1 = ok.cg(A, np.ones(A.n), history=True)
    finally:
        L.ko_set_dot_mode(0)
        L.ko_set_threads(1)
    r0 = ok.cg(A, np.ones(A.n), history=True)
    assert r0.niter == r1.niter and np.max(np.abs(r0.residuals - r1.residuals) / r0.residuals) <= 1e-13


def test_scale_goldens_are_well_formed():
    """The BASELINE-size oracle histories the GPU parity tests and bench.py compare with."""
    for name, n, niter in (("oracle_cfg2_cg512.json", 512 ** 3, 100), ("oracle_cfg3_gmres256.json", 256 ** 3, 45),
                           ("oracle_cfg5_block216.json", 216 ** 3, 7)):
        g = json.load(open(os.path.join(ROOT, "tests", "golden", name)))
        assert g["n"] == n and g["niter"] == niter and len(g["residuals"]) == niter + 1
        assert all(np.isfinite(g["residuals"])) and len(g["x_sample"]) == len(g["x_index"]) == 16
    g4 = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_cfg4_cg1024.json")))          # BASELINE cfg 4 (matrix-free oracle)
    assert g4["n"] == 1024 ** 3 and g4["nnz"] == 7 * 1024 ** 3 - 6 * 1024 ** 2 and g4["niter"] == 100 and len(g4["residuals"]) == 101
    assert g4["residuals"][0] == 32768.0 and all(np.isfinite(g4["residuals"])) and len(g4["x_sample"]) == len(g4["x_index"]) == 16
    h, hx = np.array(g4["residuals"]), np.array(g4["residuals_exact_dots"])          # the second history: the same recurrence, Dot2 dots
    assert len(hx) == 101 and hx[0] == 32768.0 and g4["oracle_vs_exact_dots_max_rel_dev"] == float(np.max(np.abs(h - hx) / hx)) < 1e-10
    for name, n, niter in (("oracle_cfg2_cg512_exact_dots.json", 512 ** 3, 1225), ("oracle_cfg3_gmres256_exact_dots.json", 256 ** 3, 940)):
        ge = json.load(open(os.path.join(ROOT, "tests", "golden", name)))          # the exact-dot histories (legs 22 / 23)
        assert ge["n"] == n and ge["niter"] == niter and len(ge["residuals"]) == niter + 1 and len(ge["prefix_residuals"]) in (101, 46)
        assert all(np.isfinite(ge["residuals"])) and ge["solved"]
    gbe = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_bicgstab256_exact_dots.json")))
    assert gbe["niter"] == 25 and len(gbe["residuals"]) == 26
    g2 = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_cfg2_cg512.json")))
    assert g2["residuals"][0] == math.sqrt(512 ** 3) and g2["nnz"] == 7 * 512 ** 3 - 6 * 512 ** 2      # ||ones||, SURVEY 8


def test_banded_random_operator_is_what_its_definition_says(oracle):
    """The non-stencil benchmark operator (SURVEY.md 8d "banded + random, fixed seed"; definition in the header of
    krylov.jl_amd/csrc/gen_irregular.cpp): symmetric, ascending columns, diagonal = 1/16 + absolute row sum, every long-range
    link an involution inside its block of rows, seeds differ, the nonsymmetric variant halves the upper triangle."""
    n = 20000
    A = oracle.banded_random(n, seed=3)
    S = A.to_scipy()
    assert abs(S - S.T).max() == 0.0
    lens = np.diff(A.rowptr)
    assert 20 <= A.nnz / n <= 31 and lens.max() <= 2 * 13 + 3 + 1
    off = abs(S).sum(axis=1).A1 - abs(S.diagonal())
    assert np.all(S.diagonal() == off + 0.0625)
    rows = np.repeat(np.arange(n), lens)
    far = np.abs(A.col - rows) > 13
    # links: block size 2^14 here; a far entry (i, j) has its mirror (j, i) and stays inside the block
    assert np.all((A.col[far] >> 14) == (rows[far] >> 14))
    assert far.sum() > 2.5 * (n >> 14 << 14)                       # ~3 per row inside the full blocks
    B = oracle.banded_random(n, seed=4)
    assert B.nnz != A.nnz or not np.array_equal(B.col, A.col)
    U = oracle.banded_random(n, seed=3, unsym=True)
    SU = U.to_scipy()
    assert np.array_equal(U.col, A.col)
    import scipy.sparse as sp
    assert abs(sp.triu(SU, 1) * 2 - sp.triu(S, 1)).max() == 0.0 and abs(sp.tril(SU, -1) - sp.tril(S, -1)).max() == 0.0
    # CG on it converges to the benchmark tolerance in a few hundred iterations
    xt = np.cos(np.arange(n) * 1e-3) + 0.5
    res = oracle.cg(A, A.matvec(xt), atol=0.0, rtol=1e-8)
    assert res.solved and 50 < res.niter < 600
    assert np.max(np.abs(res.x - xt)) < 1e-6
