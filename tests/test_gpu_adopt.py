"""The solvers on CALLER-OWNED work vectors (khip_*_workspace_adopt) through the Python mirror -- the executed twin of
julia/KrylovHIP/src/KrylovHIP.jl, whose cg! / gmres! / bicgstab! / block_gmres! methods forward to exactly these entry points
(VERDICT r04 item 1).  Each solver runs on an adopted and on a library-owned workspace (khip_*_workspace_create): iteration
counts, statuses, residual histories and solutions must be the same BITS; `solution(ws) is ws.x` (test/test_interface.jl:260);
vectors the reference allocates lazily are allocated on this side and handed over (allocate_if, src/krylov_utils.jl:281-288)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(sa, so):
    assert sa.niter == so.niter and sa.solved == so.solved and sa.status == so.status, (sa, so)
    assert np.array_equal(sa.residuals, so.residuals)


def test_python_workspaces_adopt_by_default(K, ctx):
    ws = K.CgWorkspace(ctx, 64, 64)
    assert ws.adopted and ws.x is ws.x and ws.x.ptr == K.lib().khip_cg_solution(ws._h)
    assert ws.vector("r") is ws._vec["r"] and ws.vector("z") is None
    assert ws.nbytes == 4 * 8 * 64                                     # CgWorkspace is 4 n (test/test_allocations.jl:41-57)
    assert not K.CgWorkspace(ctx, 64, 64, adopt=False).adopted


@pytest.mark.parametrize("n1", [16, 48])
@pytest.mark.parametrize("fused", [0, 1, 2])
def test_cg_adopted_equals_owned(K, ctx, n1, fused):
    A = K.CsrMatrix.stencil(ctx, "poisson", n1)
    n = n1 ** 3
    b = ctx.empty(n)
    K.kfill_(b, 1.0)
    out = {}
    for adopt in (True, False):
        ws = K.CgWorkspace(ctx, n, n, adopt=adopt)
        K.cg_(ws, A, b, history=True, fused=fused, atol=0.0, rtol=1e-8)
        out[adopt] = (ws.stats, ws.x.to_host(), ws)
    _same(out[True][0], out[False][0])
    assert np.array_equal(out[True][1], out[False][1])
    assert out[True][0].solved and out[True][0].niter > n1
    wa = out[True][2]
    assert wa.x is wa._vec["x"]                                        # solution(ws) === ws.x


def test_cg_adopted_lazy_vectors_and_warm_start(K, ctx):
    """M = Jacobi allocates z, linesearch allocates npc_dir, warm_start! allocates Δx -- on the caller's side, handed over."""
    n1 = 24
    A = K.CsrMatrix.stencil(ctx, "poisson", n1)
    n = n1 ** 3
    b = ctx.array(np.linspace(0.5, 1.5, n))
    x0 = ctx.array(np.full(n, 0.25))
    res = {}
    for adopt in (True, False):
        ws = K.CgWorkspace(ctx, n, n, adopt=adopt)
        t_create = ws.stats.allocation_timer
        assert t_create > 0.0
        M = K.Jacobi(A)
        K.cg_(ws, A, b, M=M, history=True)
        s1, x1 = ws.stats, ws.x.to_host()
        assert ws.stats.allocation_timer > t_create                    # z: allocate_if(!MisI, ws, :z, ...), src/cg.jl:142
        ws.warm_start_(x0)
        K.cg_(ws, A, b, history=True, linesearch=False)
        s2, x2 = ws.stats, ws.x.to_host()
        K.cg_(ws, A, b, history=True, linesearch=True)
        s3 = ws.stats
        res[adopt] = (s1, x1, s2, x2, s3)
        if adopt:
            assert set(ws._vec) == {"x", "r", "p", "Ap", "z", "dx", "npc_dir"}
            assert ws.nbytes == 7 * 8 * n
            assert ws.vector("z") is ws._vec["z"]
    for i in (0, 2, 4):
        _same(res[True][i], res[False][i])
    assert np.array_equal(res[True][1], res[False][1]) and np.array_equal(res[True][3], res[False][3])


@pytest.mark.parametrize("restart", [False, True])
@pytest.mark.parametrize("fused", [0, 2])
def test_gmres_adopted_equals_owned(K, ctx, restart, fused):
    n1 = 20
    A = K.CsrMatrix.stencil(ctx, "kron_unsymmetric", n1)
    n = n1 ** 3
    b = A.matvec(ctx.array(np.ones(n)))
    out = {}
    for adopt in (True, False):
        ws = K.GmresWorkspace(ctx, n, n, memory=8, adopt=adopt)
        K.gmres_(ws, A, b, history=True, restart=restart, fused=fused, itmax=60)
        out[adopt] = (ws.stats, ws.x.to_host(), ws)
    _same(out[True][0], out[False][0])
    assert np.array_equal(out[True][1], out[False][1])
    wa = out[True][2]
    if restart:
        assert len(wa.V) == 8 and "dx" in wa._vec                      # allocate_if(restart, ws, :Δx, ...), src/gmres.jl:144
    else:
        assert len(wa.V) > 8                                           # push!(V, similar(x)) through the grow callback, :319-324
        assert out[True][0].niter > 8
    c, s, z, R, inner = wa.host_state()
    assert inner >= 1 and len(R) == len(c) * (len(c) + 1) // 2
    assert np.allclose(c[:inner] ** 2 + s[:inner] ** 2, 1.0, atol=1e-14)
    # the in-place API again on the same (grown) workspace
    K.gmres_(wa, A, b, history=True, restart=restart, fused=fused, itmax=60)
    _same(wa.stats, out[False][0])


def test_gmres_adopted_preconditioned(K, ctx):
    n1 = 16
    A = K.CsrMatrix.stencil(ctx, "kron_unsymmetric", n1)
    n = n1 ** 3
    b = A.matvec(ctx.array(np.ones(n)))
    out = {}
    for adopt in (True, False):
        ws = K.GmresWorkspace(ctx, n, n, memory=10, adopt=adopt)
        K.gmres_(ws, A, b, M=K.Jacobi(A), N=K.Jacobi(A), history=True, restart=True)
        out[adopt] = (ws.stats, ws.x.to_host())
        if adopt:
            assert {"q", "p", "dx"} <= set(ws._vec)
    _same(out[True][0], out[False][0])
    assert np.array_equal(out[True][1], out[False][1])


@pytest.mark.parametrize("fused", [0, 2])
def test_bicgstab_adopted_equals_owned(K, ctx, fused):
    n1 = 24
    A = K.CsrMatrix.stencil(ctx, "kron_unsymmetric", n1)
    n = n1 ** 3
    b = A.matvec(ctx.array(np.ones(n)))
    out = {}
    for adopt in (True, False):
        ws = K.BicgstabWorkspace(ctx, n, n, adopt=adopt)
        K.bicgstab_(ws, A, b, history=True, fused=fused)
        s1, x1 = ws.stats, ws.x.to_host()
        K.bicgstab_(ws, A, b, M=K.Jacobi(A), N=K.Jacobi(A), history=True, fused=fused)
        out[adopt] = (s1, x1, ws.stats, ws.x.to_host())
        if adopt:
            assert ws.nbytes == 8 * 8 * n                              # 6 n + t + yz
    _same(out[True][0], out[False][0])
    _same(out[True][2], out[False][2])
    assert np.array_equal(out[True][1], out[False][1]) and np.array_equal(out[True][3], out[False][3])


@pytest.mark.parametrize("restart", [False, True])
def test_block_gmres_adopted_equals_owned(K, ctx, restart):
    n1, p = 18, 8
    A = K.CsrMatrix.stencil(ctx, "stencil27", n1)
    n = n1 ** 3
    rng = np.random.default_rng(7)
    B = rng.standard_normal((n, p))
    Bd = ctx.array(np.asfortranarray(B).ravel(order="F"))
    out = {}
    for adopt in (True, False):
        ws = K.BlockGmresWorkspace(ctx, n, n, p, memory=3, adopt=adopt)
        K.block_gmres_(ws, A, Bd, history=True, restart=restart, itmax=12)
        out[adopt] = (ws.stats, ws.X, ws)
    _same(out[True][0], out[False][0])
    assert np.array_equal(out[True][1], out[False][1])
    wa = out[True][2]
    assert wa.nbytes_extra < out[False][2].nbytes_extra                # no panel copy of B: it is read in place
    if not restart:
        assert len(wa.V) > 3                                           # push!(V, SM(undef, n, p)), src/block_gmres.jl:300-305
    # warm start through the caller's ΔX panel
    X0 = rng.standard_normal((n, p)) * 0.01
    res = {}
    for adopt in (True, False):
        ws = K.BlockGmresWorkspace(ctx, n, n, p, memory=3, adopt=adopt)
        ws.warm_start_(X0)
        K.block_gmres_(ws, A, Bd, history=True, restart=restart, itmax=9)
        res[adopt] = (ws.stats, ws.X)
    _same(res[True][0], res[False][0])
    assert np.array_equal(res[True][1], res[False][1])


def test_bench_line_names_the_adopt_entry_and_carries_the_int32_leg():
    """bench.py's one JSON line (here at 96^3, 6 steps): config.entry says the timed solve ran through khip_cg_workspace_adopt, and
    the nested, labelled general-CSR leg (int32 columns, spmv_codes = 0) is present beside the headline's roofline."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, KHIP_SPMV_CODES="2")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--n1", "96", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["steps"] == 6 and d["n_gpus"] == 1 and d["value"] > 0
    assert "khip_cg_workspace_adopt" in d["config"]["entry"]
    r, r32 = d["roofline"], d["roofline_int32_csr"]
    assert r["bound"] == "hbm" and 0 < r["frac"] and r["bytes_moved_per_launch"] < r["bytes_per_launch"]        # 8-bit codes: fewer bytes moved
    assert r32 is not None and "NOT_THE_HEADLINE" in r32 and r32["bytes_per_launch"] == r["bytes_per_launch"]
    assert r32["steps"] == 6 and r32["avg_ms"] > 0 and 0 < r32["frac"]


def test_adopt_argument_errors(K, ctx):
    """The adopt entries validate what a binding hands over (error codes, never a crash; khip_last_error says why)."""
    import ctypes as C
    L = K.lib()
    n = 1000
    v = [ctx.empty(n) for _ in range(6)]
    h = C.c_void_p()
    assert L.khip_cg_workspace_adopt(ctx._h, n, n, v[0].ptr, v[0].ptr, v[2].ptr, v[3].ptr, C.byref(h)) == -1      # x and r alias
    assert b"distinct" in L.khip_last_error()
    assert L.khip_cg_workspace_adopt(ctx._h, n, n, v[0].ptr, None, v[2].ptr, v[3].ptr, C.byref(h)) == -1           # a null vector
    assert L.khip_bicgstab_workspace_adopt(ctx._h, n, n, v[0].ptr, v[1].ptr, v[2].ptr, v[3].ptr, v[4].ptr, v[4].ptr, C.byref(h)) == -1
    ptrs = (C.c_void_p * 3)(v[2].ptr, None, v[4].ptr)
    assert L.khip_gmres_workspace_adopt(ctx._h, n, n, 3, v[0].ptr, v[1].ptr, ptrs, C.byref(h)) == -1                 # a null basis vector
    ptrs = (C.c_void_p * 3)(v[2].ptr, v[3].ptr, v[4].ptr)
    assert L.khip_gmres_workspace_adopt(ctx._h, n, n, 3, v[0].ptr, v[1].ptr, ptrs, C.byref(h)) == 0
    assert L.khip_gmres_workspace_adopt_vector(h, b"x", None) == -1                                                 # x cannot be emptied
    assert L.khip_gmres_workspace_adopt_vector(h, b"nope", v[5].ptr) == -1
    two = (C.c_void_p * 2)(v[2].ptr, v[3].ptr)
    assert L.khip_gmres_workspace_adopt_basis(h, 2, two) == -1                                                      # fewer vectors than the memory
    # one pointer, one slot -- a pointer that already is another vector of the workspace (or a basis vector) is refused,
    # so emptying one slot can never hand the other slot's storage to khip_free
    assert L.khip_gmres_workspace_adopt_vector(h, b"p", v[5].ptr) == 0
    assert L.khip_gmres_workspace_adopt_vector(h, b"q", v[5].ptr) == -1 and b"already is the workspace's 'p'" in L.khip_last_error()
    assert L.khip_gmres_workspace_adopt_vector(h, b"dx", v[3].ptr) == -1 and b"'V'" in L.khip_last_error()
    assert L.khip_gmres_workspace_adopt_vector(h, b"p", v[5].ptr) == 0                                              # the same pointer again: a no-op
    dup = (C.c_void_p * 3)(v[2].ptr, v[5].ptr, v[4].ptr)
    assert L.khip_gmres_workspace_adopt_basis(h, 3, dup) == -1                                                      # v[5] is p
    dup = (C.c_void_p * 3)(v[2].ptr, v[2].ptr, v[4].ptr)
    assert L.khip_gmres_workspace_adopt_basis(h, 3, dup) == -1
    assert L.khip_gmres_workspace_adopt_vector(h, b"p", None) == 0
    assert L.khip_gmres_workspace_destroy(h) == 0
    # a library-owned workspace keeps its basis to itself
    ho = C.c_void_p()
    assert L.khip_gmres_workspace_create(ctx._h, n, n, 3, C.byref(ho)) == 0
    assert L.khip_gmres_workspace_adopt_basis(ho, 3, ptrs) == -1 and b"owns its basis" in L.khip_last_error()
    assert L.khip_gmres_workspace_destroy(ho) == 0
    # re-adopting the pointer a library-owned slot already holds must not turn it into a borrowed one (it would leak):
    # the lazily allocated z of an owned workspace stays the library's and is freed by destroy (no double free, no error)
    hc = C.c_void_p()
    assert L.khip_cg_workspace_create(ctx._h, n, n, C.byref(hc)) == 0
    A = K.CsrMatrix.stencil(ctx, "poisson", 10)
    assert L.khip_cg_workspace_adopt_vector(hc, b"z", v[5].ptr) == 0
    assert L.khip_cg_workspace_adopt_vector(hc, b"dx", v[5].ptr) == -1                                               # z already holds it
    assert L.khip_cg_workspace_adopt_vector(hc, b"z", None) == 0
    assert L.khip_cg_workspace_destroy(hc) == 0
    for x in v:                                           # nothing of the caller's was freed along the way
        K.kfill_(x, 1.0)
    assert K.knorm(n, v[0]) == pytest.approx(np.sqrt(n))
