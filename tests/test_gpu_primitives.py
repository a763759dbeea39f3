"""GPU parity of the k* primitives and the CSR SpMV against the CPU oracle, through the C ABI.
Bit-exact where the arithmetic is order-independent (elementwise ops, stream SpMV, generators);
reductions: the compensated device result must agree with the oracle's extended-precision
result to 2 ulp of the result + 1e-16 * sum|x_i y_i| (tolerance stated per test)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES = [1, 2, 3, 63, 64, 65, 255, 257, 1000, 4099, 100003, (1 << 20) + 1]
EPS = np.finfo(float).eps


def _vec(rng, n):
    return rng.standard_normal(n) * np.exp(rng.uniform(-3, 3, n))


def _dev(K, ctx, a, misalign=False):
    """misalign=True puts the vector at an odd element offset (8-byte aligned only)."""
    if not misalign:
        return ctx.array(a)
    base = ctx.zeros(a.size + 1)
    v = base.slice(1, a.size + 1)
    v.copy_from_host(a)
    v._base = base
    return v


@pytest.mark.parametrize("n", SIZES)
@pytest.mark.parametrize("mis", [False, True])
def test_elementwise_ops_bit_exact(K, ctx, oracle, n, mis):
    rng = np.random.default_rng(n)
    x, y = _vec(rng, n), _vec(rng, n)
    L = oracle.lib()

    def run(gpu_fn, cpu_fn, nout=1):
        dx, dy = _dev(K, ctx, x, mis), _dev(K, ctx, y, mis)
        gpu_fn(dx, dy)
        hx, hy = x.copy(), y.copy()
        cpu_fn(hx, hy)
        assert np.array_equal(dy.to_host(), hy)
        assert np.array_equal(dx.to_host(), hx)

    dp = oracle._dp
    run(lambda a, b: K.kaxpy_(n, 0.37, a, b), lambda a, b: L.ko_axpy(n, 0.37, dp(a), dp(b)))
    run(lambda a, b: K.kaxpby_(n, -1.25, a, 0.7, b), lambda a, b: L.ko_axpby(n, -1.25, dp(a), 0.7, dp(b)))
    run(lambda a, b: K.kcopy_(n, b, a), lambda a, b: L.ko_copy(n, dp(b), dp(a)))
    run(lambda a, b: K.kscal_(n, 3.3, b), lambda a, b: L.ko_scal(n, 3.3, dp(b)))
    run(lambda a, b: K.kdiv_(n, b, 3.3), lambda a, b: L.ko_div(n, dp(b), 3.3))
    run(lambda a, b: K.kscalcopy_(n, b, -0.1, a), lambda a, b: L.ko_scalcopy(n, dp(b), -0.1, dp(a)))
    run(lambda a, b: K.kdivcopy_(n, b, a, 7.7), lambda a, b: L.ko_divcopy(n, dp(b), dp(a), 7.7))
    run(lambda a, b: K.kfill_(b, 2.5), lambda a, b: L.ko_fill(n, dp(b), 2.5))
    run(lambda a, b: K.kref_(n, a, b, 0.6, 0.8), lambda a, b: L.ko_ref(n, dp(a), dp(b), 0.6, 0.8))
    # fused copy+axpy == kcopy! then kaxpy!
    dw, dx, dy = ctx.zeros(n), _dev(K, ctx, x, mis), _dev(K, ctx, y, mis)
    K.waxpy_(n, dw, dx, -0.45, dy)
    ref = x.copy()
    L.ko_axpy(n, -0.45, dp(y), dp(ref))
    assert np.array_equal(dw.to_host(), ref)
    # exact aliasing (BLAS semantics): y += s*y, w aliasing x
    dy = _dev(K, ctx, y, mis)
    K.kaxpy_(n, 0.5, dy, dy)
    ref = y.copy()
    L.ko_axpy(n, 0.5, dp(y.copy()), dp(ref))
    assert np.array_equal(dy.to_host(), ref)


@pytest.mark.parametrize("n", SIZES + [5_000_003])
@pytest.mark.parametrize("mis", [False, True])
def test_reductions_match_oracle(K, ctx, oracle, parity_log, n, mis):
    rng = np.random.default_rng(n + 17)
    x, y = _vec(rng, n), _vec(rng, n)
    dx, dy = _dev(K, ctx, x, mis), _dev(K, ctx, y, mis)
    absum = float(np.abs(x * y).sum())
    d_gpu, d_cpu = K.kdot(n, dx, dy), oracle.dot(x, y)
    tol = 2 * EPS * abs(d_cpu) + 1e-16 * absum
    assert abs(d_gpu - d_cpu) <= tol, (d_gpu, d_cpu)
    n_gpu, n_cpu = K.knorm(n, dx), oracle.nrm2(x)
    assert abs(n_gpu - n_cpu) <= 2 * EPS * n_cpu
    # aliased dot (src/cg.jl:242 with z === r) equals norm^2
    assert abs(K.kdot(n, dx, dx) - oracle.dot(x, x)) <= 2 * EPS * oracle.dot(x, x)
    a, b = K.dot2(n, dx, dy)
    assert a == d_gpu or abs(a - d_gpu) <= tol
    assert abs(b - oracle.dot(x, x)) <= 2 * EPS * b
    parity_log(test="dot", n=n, misaligned=mis, rel=abs(d_gpu - d_cpu) / max(abs(d_cpu), 1e-300),
               rel_to_abs=abs(d_gpu - d_cpu) / absum)
    # run-to-run determinism
    assert K.kdot(n, dx, dy) == d_gpu


def test_reduction_uncompensated_mode(K, ctx, oracle):
    n = 1_000_003
    rng = np.random.default_rng(1)
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    dx, dy = ctx.array(x), ctx.array(y)
    ctx.set_option("compensated", 0)
    try:
        d = K.kdot(n, dx, dy)
    finally:
        ctx.set_option("compensated", 1)
    assert abs(d - oracle.dot(x, y)) <= 1e-12 * float(np.abs(x * y).sum())


def test_empty_vectors(K, ctx):
    v = ctx.empty(4)
    assert K.kdot(0, v, v) == 0.0 and K.knorm(0, v) == 0.0
    K.kaxpy_(0, 1.0, v, v)


@pytest.mark.parametrize("n", [1, 2, 3, 1001, 300007, 5_000_003])
@pytest.mark.parametrize("mis", [False, True])
def test_one_pass_cg_setup_equals_the_four_primitives(K, ctx, oracle, n, mis):
    """khip_cg_setup (round 4): kfill!(x, 0); kcopy!(r, b); kcopy!(p, r); kdotr(r, r) of src/cg.jl:153-162 in one pass -- the same
    vectors bit for bit, gamma equal to the device's own kdot (and to the oracle's within the dot's bound), and cg! histories
    identical with the pass on and off."""
    rng = np.random.default_rng(7 * n + mis)
    b = _vec(rng, n)
    db = _dev(K, ctx, b, mis)
    dx, dr, dp_ = (_dev(K, ctx, rng.standard_normal(n), mis) for _ in range(3))     # garbage to overwrite
    g = K.cg_setup_(n, db, dx, dr, dp_)
    assert np.array_equal(dr.to_host(), b) and np.array_equal(dp_.to_host(), b)
    xh = dx.to_host()
    assert np.array_equal(xh, np.zeros(n)) and not np.signbit(xh).any()
    assert g == K.kdot(n, db, db)                          # the same bits as the separate reduction
    gc = oracle.dot(b, b)
    assert abs(g - gc) <= 2 * EPS * gc
    assert np.array_equal(db.to_host(), b)                 # b untouched
    if n == 300007 and not mis:
        A_cpu = oracle.poisson3d(20)
        A = K.CsrMatrix.stencil(ctx, "poisson", 20)
        bb = ctx.array(np.linspace(0.5, 2.0, A_cpu.n))
        prev = ctx.get_option("cg_setup_fused")
        try:
            hist = {}
            for on in (1, 0):
                ctx.set_option("cg_setup_fused", on)
                for fused in (2, 1):
                    xs, st, _ = K.cg(A, bb, history=True, fused=fused)
                    hist[(on, fused)] = (st.niter, st.residuals.copy(), xs.to_host())
        finally:
            ctx.set_option("cg_setup_fused", prev)
        ref = hist[(0, 1)]
        for k, v in hist.items():
            assert v[0] == ref[0] and np.array_equal(v[1], ref[1]) and np.array_equal(v[2], ref[2]), k
        with pytest.raises(K.KhipError):
            K.cg_setup_(8, bb, bb, dr, dp_)                # the four vectors must be distinct


@pytest.mark.parametrize("n", [1, 2, 1001, 300007])
def test_fused_axpy_sqnorm_and_cg_update(K, ctx, oracle, n):
    """The two kernels of the fused CG iteration (src/cg.jl:239-242,259) against the unfused device
    sequence (bit-identical vectors) and the oracle."""
    rng = np.random.default_rng(100 + n)
    p, q, x, r = (_vec(rng, n) for _ in range(4))
    alpha, beta = 0.3, 0.7
    dp_, dq, dx, dr = (ctx.array(v) for v in (p, q, x, r))
    g = K.axpy_sqnorm(n, -alpha, dq, dr)            # r -= alpha q ; r.r
    K.cg_update_(n, alpha, beta, dr, dp_, dx)       # x += alpha p ; p = r + beta p
    ux, ur, up = ctx.array(x), ctx.array(r), ctx.array(p)
    K.kaxpy_(n, alpha, up, ux)
    K.kaxpy_(n, -alpha, dq, ur)
    g2 = K.kdot(n, ur, ur)
    K.kaxpby_(n, 1.0, ur, beta, up)
    assert np.array_equal(dx.to_host(), ux.to_host())
    assert np.array_equal(dr.to_host(), ur.to_host())
    assert np.array_equal(dp_.to_host(), up.to_host())
    assert abs(g - g2) <= 2 * EPS * abs(g2)
    hx, hr, hp = x.copy(), r.copy(), p.copy()
    oracle.axpy(alpha, hp, hx)
    oracle.axpy(-alpha, q, hr)
    oracle.axpby(1.0, hr, beta, hp)
    assert np.array_equal(dx.to_host(), hx) and np.array_equal(dr.to_host(), hr) and np.array_equal(dp_.to_host(), hp)
    assert abs(g - oracle.dot(hr, hr)) <= 2 * EPS * g


@pytest.mark.parametrize("n", [2, 1001, 300007])
def test_fused_axpy2_dot(K, ctx, oracle, n):
    rng = np.random.default_rng(n)
    p, q, x, r = (_vec(rng, n) for _ in range(4))
    dp_, dq, dx, dr = (ctx.array(v) for v in (p, q, x, r))
    g = K.axpy2_dot(n, 0.3, dp_, dq, dx, dr)
    # unfused sequence on the device
    ux, ur = ctx.array(x), ctx.array(r)
    K.kaxpy_(n, 0.3, dp_, ux)
    K.kaxpy_(n, -0.3, dq, ur)
    g2 = K.kdot(n, ur, ur)
    assert np.array_equal(dx.to_host(), ux.to_host()) and np.array_equal(dr.to_host(), ur.to_host())
    assert abs(g - g2) <= 2 * EPS * abs(g2)
    # and against the oracle
    hx, hr = x.copy(), r.copy()
    oracle.axpy(0.3, p, hx)
    oracle.axpy(-0.3, q, hr)
    assert np.array_equal(dx.to_host(), hx) and np.array_equal(dr.to_host(), hr)
    assert abs(g - oracle.dot(hr, hr)) <= 2 * EPS * g


@pytest.mark.parametrize("n,k", [(1000, 1), (4097, 5), (200001, 30)])
def test_mgs_and_multi_axpy(K, ctx, oracle, n, k):
    rng = np.random.default_rng(k)
    Q, _ = np.linalg.qr(rng.standard_normal((n, k)))
    V = [np.ascontiguousarray(Q[:, i]) for i in range(k)]
    q = rng.standard_normal(n)
    dV = [ctx.array(v) for v in V]
    dq = ctx.array(q)
    h, nrm = K.mgs_(n, dV, dq)
    # oracle: the reference's loop, src/gmres.jl:259-262,274
    hq = q.copy()
    href = []
    for i in range(k):
        hi = oracle.dot(V[i], hq)
        href.append(hi)
        oracle.axpy(-hi, V[i], hq)
    scale = np.linalg.norm(q)
    assert np.allclose(h, href, rtol=0, atol=4 * EPS * scale)
    assert np.allclose(dq.to_host(), hq, rtol=0, atol=8 * EPS * scale)
    assert abs(nrm - oracle.nrm2(hq)) <= 1e-14 * scale
    # unfused device sequence: same coefficients to 1 ulp-ish, same vector
    dq2 = ctx.array(q)
    for i in range(k):
        hi = K.kdot(n, dV[i], dq2)
        K.kaxpy_(n, -hi, dV[i], dq2)
    assert np.allclose(dq.to_host(), dq2.to_host(), rtol=0, atol=8 * EPS * scale)
    # accumulate pass (reorthogonalisation, src/gmres.jl:265-271)
    h2, _ = K.mgs_(n, dV, dq, accumulate_into=h)
    assert np.allclose(h2, h, rtol=0, atol=1e-12 * scale)
    # multi-axpy == k kaxpy! calls, bit for bit
    y = rng.standard_normal(k)
    x0 = rng.standard_normal(n)
    dx1, dx2 = ctx.array(x0), ctx.array(x0)
    K.multi_axpy_(n, y, dV, dx1)
    for i in range(k):
        K.kaxpy_(n, float(y[i]), dV[i], dx2)
    assert np.array_equal(dx1.to_host(), dx2.to_host())
    hx = x0.copy()
    for i in range(k):
        oracle.axpy(float(y[i]), V[i], hx)
    assert np.array_equal(dx1.to_host(), hx)


# ------------------------------------------------------------------------------ CSR

def _upload(K, ctx, A):
    return K.CsrMatrix.from_host(ctx, A.rowptr, A.col, A.val, (A.n, A.n))


@pytest.mark.parametrize("kind,dims", [("poisson", (16, 16, 16)), ("poisson", (33, 17, 9)), ("poisson", (1, 1, 1)),
                                       ("poisson", (2, 3, 1)), ("kron_unsymmetric", (12,)), ("stencil27", (11,))])
def test_device_generators_bit_exact(K, ctx, oracle, kind, dims):
    gen = {"poisson": oracle.poisson3d, "kron_unsymmetric": oracle.kron_unsymmetric, "stencil27": oracle.stencil27_unsym}
    A = gen[kind](*dims)
    rowptr, col, val = K.gen_stencil_arrays(ctx, kind, *dims)
    assert np.array_equal(rowptr, A.rowptr) and np.array_equal(col, A.col) and np.array_equal(val, A.val)
    # a row slab with global columns (distributed layout)
    r0, r1 = A.n // 3, (2 * A.n) // 3 + 1
    rp2, col2, val2 = K.gen_stencil_arrays(ctx, kind, *dims, rows=(r0, r1))
    sl = A.row_slice(r0, r1)
    assert np.array_equal(rp2, sl.rowptr) and np.array_equal(col2, sl.col) and np.array_equal(val2, sl.val)


@pytest.mark.parametrize("n,kw", [(5000, {}), (70001, dict(seed=7)), (1 << 16, dict(half_band=5, links=6, seed=3)),
                                  (40000, dict(unsym=True, dense_rows=3)), (300, dict(links=2))])
def test_banded_random_generator_equals_the_oracles(K, ctx, oracle, n, kw):
    """The non-stencil benchmark operator ("banded + random, fixed seed"): the product's generator
    (csrc/gen_irregular.cpp) and the oracle's independent restatement (ko_csr_banded_random) give the same arrays; the
    operator is what its definition says -- symmetric pattern (and values unless unsym), ascending columns, strictly
    diagonally dominant, far more than 2048 distinct diagonals at size."""
    A = oracle.banded_random(n, **kw)
    rowptr, col, val = K.gen_banded_random_arrays(ctx, n, **kw)
    assert np.array_equal(rowptr, A.rowptr) and np.array_equal(col, A.col) and np.array_equal(val, A.val)
    r0, r1 = n // 3, (2 * n) // 3 + 1
    rp2, col2, val2 = K.gen_banded_random_arrays(ctx, n, rows=(r0, r1), **kw)
    sl = A.row_slice(r0, r1)
    assert np.array_equal(rp2, sl.rowptr) and np.array_equal(col2, sl.col) and np.array_equal(val2, sl.val)
    S = A.to_scipy()
    if not kw.get("unsym"):
        assert abs(S - S.T).max() == 0.0
    else:
        P = S.copy(); P.data[:] = 1.0
        lens = np.diff(A.rowptr)
        assert np.sort(lens)[-3] > 3000                                   # the dense rows
        dense = set(np.argsort(lens)[-3:].tolist())
        keep = np.array([i not in dense for i in range(n)])
        Pk = P[keep][:, keep]
        assert abs(Pk - Pk.T).max() == 0.0                                # pattern symmetric apart from the dense rows
    off = abs(S).sum(axis=1).A1 - abs(S.diagonal())
    assert np.all(S.diagonal() == off + 0.0625)
    lens = np.diff(A.rowptr)
    for i in range(0, n, 211):
        assert np.all(np.diff(A.col[A.rowptr[i]:A.rowptr[i + 1]]) > 0)
    if n >= 40000:
        diags = np.unique(A.col.astype(np.int64) - np.repeat(np.arange(n, dtype=np.int64), lens))
        assert diags.size > 2048


@pytest.mark.parametrize("unsym,dense_rows", [(False, 0), (True, 2)])
def test_irregular_operator_through_the_whole_path(K, ctx, oracle, parity_log, unsym, dense_rows):
    """The banded + random operator at a moderate size through SpMV (int32 column stream: too many diagonals to code),
    fused SpMV + dot, SpMM with 16 and 8 columns (tile kernel on consecutive-row groups / window / direct), cg! (symmetric)
    or gmres! + bicgstab! (nonsymmetric, with rows of 3000 entries) and block_gmres! -- bit-exact products, oracle histories."""
    n = 70001
    A = oracle.banded_random(n, seed=5, unsym=unsym, dense_rows=dense_rows)
    dA = K.CsrMatrix.banded_random(ctx, n, seed=5, unsym=unsym, dense_rows=dense_rows)
    assert dA.nnz == A.nnz
    rng = np.random.default_rng(17)
    x = rng.standard_normal(n)
    y_ref = A.matvec(x)
    dx = ctx.array(x)
    assert np.array_equal(dA.matvec(dx).to_host(), y_ref)
    assert dA.code_info[0] == 32                                       # nothing stencil-specific: plain int32 columns
    for kern in (1, 3, 4):                                             # stream, ordered sub-wave and staged-rows kernels: serial order
        ctx.set_option("spmv_kernel", kern)
        assert np.array_equal(dA.matvec(dx).to_host(), y_ref), kern
    ctx.set_option("spmv_kernel", 2)                                   # strided vector kernel (rows of hundreds of entries): FMA + tree, not bit-identical
    assert np.max(np.abs(dA.matvec(dx).to_host() - y_ref)) <= 1e-13 * np.max(np.abs(y_ref))
    ctx.set_option("spmv_kernel", 0)
    dy = ctx.zeros(n)
    d = K.spmv_dot(dA, dx, dy)                                         # fused x . (A x)
    assert np.array_equal(dy.to_host(), y_ref)
    ref_dot = oracle.dot(x, y_ref)
    assert abs(d - ref_dot) <= 4e-16 * np.abs(x * y_ref).sum()
    # SpMM
    for p in (16, 8):
        X = rng.standard_normal((n, p))
        ref = np.stack([A.matvec(np.ascontiguousarray(X[:, j])) for j in range(p)], axis=1)
        outs = []
        for tile, window in ((1, 1), (0, 1), (0, 0)):
            ctx.set_option("spmm_tile", tile); ctx.set_option("spmm_window", window)
            dY = K.Panel(ctx, n, p)
            K.spmm_(dA, K.Panel.from_host(ctx, X), dY)
            outs.append(dY.to_host())
        ctx.set_option("spmm_tile", 1); ctx.set_option("spmm_window", 1)
        assert all(np.array_equal(o, ref) for o in outs), p
    info = dA.tile_info
    assert info["state"] == 1 and info["grid_tiles"] == 0, info
    if dense_rows:
        assert info["direct_groups"] >= dense_rows, info
    # solvers: b = A x_true
    xt = np.cos(np.arange(n) * 1e-3) + 0.5
    b = A.matvec(xt)
    db = ctx.array(b)
    if not unsym:
        # CG's residual recurrence on this operator (condition ~1e3, eigenvalue clusters from the random links) amplifies
        # the different roundings of the dots: the histories agree to 1e-10 over the first 60 iterations and drift apart
        # after that, and the oracle itself sits 0.24 % above the threshold one iteration before it stops -- so the
        # iteration counts may differ by one or two; what must hold is the tolerance on the TRUE residual.
        ref = oracle.cg(A, b, history=True, atol=0.0, rtol=1e-8)
        for fused in (0, 1, 2):
            xs, st, _ = K.cg(dA, db, history=True, atol=0.0, rtol=1e-8, fused=fused)
            m = min(st.niter, ref.niter) + 1
            dev = np.abs(st.residuals[:m] - ref.residuals[:m]) / ref.residuals[:m]
            true_res = np.linalg.norm(b - A.matvec(xs.to_host())) / np.linalg.norm(b)
            parity_log(test="irregular_cg", n=n, fused=fused, niter=st.niter, ref_niter=ref.niter, hist_max_rel_first60=float(dev[:61].max()),
                       hist_max_rel_all=float(dev.max()), true_rel_residual=float(true_res))
            assert st.solved and abs(st.niter - ref.niter) <= 2 and dev[:61].max() <= 1e-10 and true_res <= 2e-8, (fused, st.niter, ref.niter, dev[:61].max(), true_res)
    else:
        # (to rtol 1e-8 gmres!(20) takes 1333 and bicgstab! 787 iterations here: prefixes are compared)
        ref = oracle.gmres(A, b, memory=20, restart=True, history=True, atol=0.0, rtol=0.0, itmax=50)
        xs, st, _ = K.gmres(dA, db, memory=20, restart=True, history=True, atol=0.0, rtol=0.0, itmax=50)
        dev = float(np.max(np.abs(st.residuals - ref.residuals) / ref.residuals))
        parity_log(test="irregular_gmres", n=n, niter=st.niter, ref_niter=ref.niter, hist_max_rel=dev)
        assert st.niter == ref.niter == 50 and dev <= 1e-10, (st.niter, ref.niter, dev)
        ref = oracle.bicgstab(A, b, history=True, atol=0.0, rtol=0.0, itmax=30)
        xs, st, _ = K.bicgstab(dA, db, history=True, atol=0.0, rtol=0.0, itmax=30)
        # bicgstab!'s recurrences amplify the rounding differences of the dots quickly on this operator (DESIGN 3.2b shows the
        # same for the oracle against its own binary128 build): the histories agree to 1e-10 over the first iterations only
        devs = np.abs(st.residuals - ref.residuals) / ref.residuals
        parity_log(test="irregular_bicgstab", n=n, niter=st.niter, ref_niter=ref.niter, hist_max_rel=float(devs.max()),
                   per_iteration=[float(v) for v in devs])
        assert st.niter == ref.niter == 30 and devs[:9].max() <= 1e-10, (st.niter, ref.niter, devs)
    p = 16
    B = np.random.default_rng(23).standard_normal((n, p))          # a well-conditioned block (the panel QRs differ in how they round: Householder in the oracle, CholeskyQR2 here)
    # block_gmres!(memory = 5) needs > 1000 iterations on this operator: 24 iterations (four restarts) are compared
    ref = oracle.block_gmres(A, B, memory=5, history=True, restart=True, atol=0.0, rtol=0.0, itmax=24)
    Xs, st, _ = K.block_gmres(dA, B, memory=5, ctx=ctx, history=True, restart=True, atol=0.0, rtol=0.0, itmax=24)
    devs = np.abs(np.array(st.residuals) - ref.residuals) / ref.residuals
    dev = float(devs.max())
    parity_log(test="irregular_block_gmres", n=n, niter=st.niter, ref_niter=ref.niter, hist_max_rel=dev, per_iteration=[float(v) for v in devs])
    assert st.niter == ref.niter == 24 and dev <= 1e-9, (st.niter, ref.niter, dev)


@pytest.mark.parametrize("dims", [(16, 16, 16), (33, 17, 9), (64, 64, 64), (5, 1, 1)])
def test_spmv_stream_bit_exact_vs_oracle(K, ctx, oracle, parity_log, dims):
    A = oracle.poisson3d(*dims)
    rng = np.random.default_rng(11)
    x = _vec(rng, A.n)
    dA = _upload(K, ctx, A)
    y_ref = A.matvec(x)
    dx = ctx.array(x)
    defaults = {k: ctx.get_option(k) for k in ("spmv_kernel", "spmv_rows", "spmv_vec", "spmv_nt", "spmv_xcd",
                                               "spmv_persist", "spmv_lanes")}
    try:
        # LDS-staged stream kernel: every tiling / load width / launch shape gives the same bits
        for rows in (256, 128, 64, 32):
            for vec in (1, 2):
                for nt in (0, 1):
                    for persist, xcd in ((0, 0), (1, 0), (1, 1)):
                        for k, v in dict(spmv_kernel=1, spmv_rows=rows, spmv_vec=vec, spmv_nt=nt,
                                         spmv_persist=persist, spmv_xcd=xcd).items():
                            ctx.set_option(k, v)
                        dy = ctx.zeros(A.n)
                        dA.matvec(dx, dy)
                        assert np.array_equal(dy.to_host(), y_ref), ("stream", rows, vec, nt, persist, xcd)
        # staged-rows kernel (one lane per row out of LDS)
        for nt in (0, 1):
            for persist, xcd in ((0, 0), (0, 16), (1, 0)):
                for k, v in dict(spmv_kernel=4, spmv_nt=nt, spmv_persist=persist, spmv_xcd=xcd).items():
                    ctx.set_option(k, v)
                dy = ctx.zeros(A.n)
                dA.matvec(dx, dy)
                assert np.array_equal(dy.to_host(), y_ref), ("stage", nt, persist, xcd)
        # ordered sub-wave kernel: any lane count (rows longer than L take the chunk loop)
        for lanes in (4, 8, 16, 32, 64):
            for nt in (0, 1):
                for persist in (0, 1):
                    for k, v in dict(spmv_kernel=3, spmv_lanes=lanes, spmv_nt=nt, spmv_persist=persist,
                                     spmv_xcd=0).items():
                        ctx.set_option(k, v)
                    dy = ctx.zeros(A.n)
                    dA.matvec(dx, dy)
                    assert np.array_equal(dy.to_host(), y_ref), ("ordered", lanes, nt, persist)
    finally:
        for k, v in defaults.items():
            ctx.set_option(k, v)
    # device-generated operator gives the same product
    dB = K.CsrMatrix.stencil(ctx, "poisson", *dims)
    assert dB.nnz == A.nnz
    assert np.array_equal(dB.matvec(dx).to_host(), y_ref)
    assert dB.spmv_bytes == 12 * A.nnz + 4 * (A.n + 1) + 16 * A.n
    parity_log(test="spmv_stream", dims=dims, bit_exact=True)


@pytest.mark.parametrize("lanes", [4, 8, 16, 32, 64])
def test_spmv_vector_kernel(K, ctx, oracle, lanes):
    A = oracle.stencil27_unsym(13)
    rng = np.random.default_rng(2)
    x = _vec(rng, A.n)
    dA = _upload(K, ctx, A)
    ctx.set_option("spmv_kernel", 2); ctx.set_option("spmv_lanes", lanes)
    try:
        y = dA.matvec(ctx.array(x)).to_host()
    finally:
        ctx.set_option("spmv_kernel", 0); ctx.set_option("spmv_lanes", 0)
    y_ref = A.matvec(x)
    S = A.to_scipy()
    bound = 40 * EPS * (abs(S) @ np.abs(x))          # FMA + tree order: a few ulps of sum |a_ij x_j|
    assert np.all(np.abs(y - y_ref) <= bound)


def test_spmv_general_matrix_and_long_rows(K, ctx):
    import scipy.sparse as sp
    rng = np.random.default_rng(9)
    n = 3001
    S = sp.random(n, n, density=0.01, random_state=4, format="lil")
    S[7, :] = rng.standard_normal(n)           # one dense row (3001 entries > one LDS pass)
    S[n - 1, : n // 2] = 1.0
    S = S.tocsr()
    S.sort_indices()
    x = rng.standard_normal(n)
    dA = K.CsrMatrix.from_scipy(ctx, S)
    ref = S @ x
    bound = 64 * EPS * (abs(S) @ np.abs(x)) + 1e-300
    ys = {}
    for kernel in (0, 1, 2, 3, 4):
        ctx.set_option("spmv_kernel", kernel)
        try:
            y = dA.matvec(ctx.array(x)).to_host()
        finally:
            ctx.set_option("spmv_kernel", 0)
        assert np.all(np.abs(y - ref) <= bound), kernel
        ys[kernel] = y
    # the two in-order kernels (LDS-staged and ordered sub-wave) agree bit for bit on ANY matrix,
    # including the 3001-entry row that spans several LDS passes / 47 lane-group chunks
    assert np.array_equal(ys[1], ys[3]) and np.array_equal(ys[4], ys[3])
    # sequential reference in stored order: rounded product, rounded add
    seq = np.zeros(n)
    for i in range(n):
        acc = 0.0
        for j in range(S.indptr[i], S.indptr[i + 1]):
            acc = acc + S.data[j] * x[S.indices[j]]
        seq[i] = acc
    assert np.array_equal(ys[3], seq)
    # 1-based input arrays (Julia convention)
    dB = K.CsrMatrix.from_host(ctx, S.indptr + 1, S.indices + 1, S.data, S.shape, index_base=1)
    assert np.all(np.abs(dB.matvec(ctx.array(x)).to_host() - ref) <= bound)
    # 64-bit row pointers
    dC = K.CsrMatrix.from_host(ctx, S.indptr.astype(np.int64), S.indices, S.data, S.shape)
    assert np.all(np.abs(dC.matvec(ctx.array(x)).to_host() - ref) <= bound)
    # empty operator
    E = sp.csr_matrix((5, 5))
    dE = K.CsrMatrix.from_scipy(ctx, E)
    assert np.array_equal(dE.matvec(ctx.array(np.ones(5))).to_host(), np.zeros(5))


@pytest.mark.parametrize("dims", [(16, 16, 16), (40, 40, 40)])
def test_spmv_dot_fused_equals_unfused(K, ctx, oracle, dims):
    A = oracle.poisson3d(*dims)
    rng = np.random.default_rng(21)
    x = _vec(rng, A.n)
    dA = _upload(K, ctx, A)
    dx = ctx.array(x)
    dy = ctx.zeros(A.n)
    d = K.spmv_dot(dA, dx, dy)
    y_ref = A.matvec(x)
    assert np.array_equal(dy.to_host(), y_ref)
    d_unfused = K.kdot(A.n, dx, dy)
    d_cpu = oracle.dot(x, y_ref)
    tol = 2 * EPS * abs(d_cpu) + 1e-16 * float(np.abs(x * y_ref).sum())
    assert abs(d - d_unfused) <= tol and abs(d - d_cpu) <= tol


def test_spmm_rowmajor_panel(K, ctx, oracle):
    A = oracle.stencil27_unsym(9)
    rng = np.random.default_rng(31)
    for p in (1, 3, 16):
        X = rng.standard_normal((A.n, p))
        dA = _upload(K, ctx, A)
        dX = ctx.array(X.ravel())          # row-major n x p
        dY = ctx.zeros(A.n * p)
        K._ck(K.lib().khip_spmm(ctx._h, dA._h, dX.ptr, dY.ptr, p))
        Y = dY.to_host().reshape(A.n, p)
        ref = np.stack([A.matvec(np.ascontiguousarray(X[:, j])) for j in range(p)], axis=1)
        assert np.array_equal(Y, ref)


# ------------------------------------------------------------------ full-size properties (cfg 2)

def test_full_size_spmv_properties_512(K, ctx, parity_log):
    """BASELINE cfg 2 (512^3, n = 134,217,728, nnz = 937,951,232): properties that need no oracle run.
    (1) A*ones = number of missing neighbours per row (exact small integers), (2) linearity
    A(ax + by) = a Ax + b Ay to rounding, (3) symmetry <Ax, y> = <x, Ay> to reduction accuracy."""
    n1 = 512
    n = n1 ** 3
    A = K.CsrMatrix.stencil(ctx, "poisson", n1)
    assert A.nnz == 7 * n - 6 * n1 * n1 == 937_951_232
    ones = ctx.empty(n)
    K.kfill_(ones, 1.0)
    y = ctx.empty(n)
    A.matvec(ones, y)
    s1 = K.kdot(n, ones, y)            # sum of row sums = 6 n - (nnz - n) = 6 * n1^2 * ... exact integer
    assert s1 == float(6 * n - (A.nnz - n))
    sq = K.kdot(n, y, y)               # sum of (missing neighbours)^2, exact integer
    faces = 6 * (n1 - 2) ** 2
    edges = 12 * (n1 - 2)
    corners = 8
    assert sq == float(faces * 1 + edges * 4 + corners * 9)
    # non-trivial vectors built on the device: x = 0.5 + 0.25 * (A ones), z = 2 * ones - 3 * x
    x = ctx.empty(n)
    K.kfill_(x, 0.5)
    K.kaxpy_(n, 0.25, y, x)
    Ax = ctx.empty(n)
    A.matvec(x, Ax)
    z = ctx.empty(n)
    K.kcopy_(n, z, x)
    K.kaxpby_(n, 2.0, ones, -3.0, z)
    Az = ctx.empty(n)
    A.matvec(z, Az)
    # symmetry: <A x, z> == <x, A z>
    lhs, rhs = K.kdot(n, Ax, z), K.kdot(n, x, Az)
    sym = abs(lhs - rhs) / abs(lhs)
    assert sym <= 1e-13
    # linearity: A z == 2 * (A ones) - 3 * (A x)
    K.kaxpby_(n, 2.0, y, -3.0, Ax)
    K.kaxpy_(n, -1.0, Az, Ax)
    lin = K.knorm(n, Ax) / K.knorm(n, Az)
    assert lin <= 1e-14
    parity_log(test="spmv_512_properties", linearity_rel=lin, symmetry_rel=sym)


@pytest.mark.parametrize("n1", [5, 12])
def test_fused_bicgstab_passes(K, ctx, oracle, n1):
    """The five passes of the fused bicgstab! iteration (src/bicgstab.jl:221-240) against the unfused device
    sequence (vectors bit-identical, reductions within an ulp) and the oracle."""
    A = oracle.kron_unsymmetric(n1)
    n = A.n
    dA = K.CsrMatrix.from_host(ctx, A.rowptr, A.col, A.val, (n, n))
    rng = np.random.default_rng(n1)
    p, c, r, x0 = (_vec(rng, n) for _ in range(4))
    dp_, dc, dr = ctx.array(p), ctx.array(c), ctx.array(r)
    # v = A p ; c . v
    dv = ctx.empty(n)
    cv = K.spmv_dotw(dA, dp_, dv, dc)
    v = A.matvec(p)
    assert np.array_equal(dv.to_host(), v)
    assert abs(cv - oracle.dot(c, v)) <= 4 * EPS * np.abs(c * v).sum()
    # s = r - alpha v ; x += alpha p
    alpha = 0.37
    ds, dx = ctx.empty(n), ctx.array(x0)
    K.bicgstab_sx_(n, alpha, dr, dv, dp_, ds, dx)
    us, ux = ctx.empty(n), ctx.array(x0)
    K.waxpy_(n, us, dr, -alpha, dv)
    K.kaxpy_(n, alpha, dp_, ux)
    assert np.array_equal(ds.to_host(), us.to_host()) and np.array_equal(dx.to_host(), ux.to_host())
    s = r.copy(); oracle.axpy(-alpha, v, s)
    assert np.array_equal(ds.to_host(), s)
    # t = A s ; (t . s, t . t)
    dt = ctx.empty(n)
    ts, tt = K.spmv_dot2(dA, ds, dt)
    t = A.matvec(s)
    assert np.array_equal(dt.to_host(), t)
    assert abs(ts - oracle.dot(t, s)) <= 4 * EPS * np.abs(t * s).sum() and abs(tt - oracle.dot(t, t)) <= 4 * EPS * tt
    # x += omega s ; r = s - omega t ; (c . r, r . r)
    omega = ts / tt
    rho, rr = K.bicgstab_xr_(n, omega, ds, dt, ds, dc, dx, dr)
    K.kaxpy_(n, omega, ds, ux)
    ur = ctx.empty(n)
    K.waxpy_(n, ur, ds, -omega, dt)
    assert np.array_equal(dx.to_host(), ux.to_host()) and np.array_equal(dr.to_host(), ur.to_host())
    rn = ur.to_host()
    assert abs(rho - oracle.dot(c, rn)) <= 4 * EPS * np.abs(c * rn).sum() and abs(rr - oracle.dot(rn, rn)) <= 4 * EPS * rr
    assert abs(rho - K.kdot(n, dc, ur)) <= 4 * EPS * np.abs(c * rn).sum()
    # p = r + beta (p - omega v)
    beta = -0.81
    up = ctx.array(p)
    K.bicgstab_p_(n, omega, beta, dv, dr, dp_)
    K.kaxpy_(n, -omega, dv, up)
    K.kaxpby_(n, 1.0, ur, beta, up)
    assert np.array_equal(dp_.to_host(), up.to_host())
    hp = p.copy(); oracle.axpy(-omega, v, hp); oracle.axpby(1.0, rn, beta, hp)
    assert np.array_equal(dp_.to_host(), hp)


def test_spmv_dot2_on_long_rows_falls_back(K, ctx, oracle):
    """Only the staged-rows kernel carries the second reduction; other row shapes get y.y from a separate pass."""
    rng = np.random.default_rng(3)
    import scipy.sparse as sp
    M = sp.random(300, 300, density=0.4, random_state=3, format="csr") + sp.eye(300, format="csr")
    M.sort_indices()
    dA = K.CsrMatrix.from_host(ctx, M.indptr.astype(np.int64), M.indices.astype(np.int32), M.data, (300, 300))
    x = rng.standard_normal(300)
    dy = ctx.empty(300)
    xy, yy = K.spmv_dot2(dA, ctx.array(x), dy)
    y = dy.to_host()
    assert np.allclose(y, M @ x, rtol=1e-13, atol=1e-13)
    assert abs(xy - x @ y) <= 1e-12 * np.abs(x * y).sum() and abs(yy - y @ y) <= 1e-12 * yy


@pytest.mark.parametrize("case", ["kron", "rect", "random", "empty_cols"])
def test_adjoint_operator_bit_exact(K, ctx, oracle, case):
    """khip_csr_transpose: y = A' x (`mul!(y, A', x)`, docs/src/matrix_free.md:36-42) accumulates each entry of a
    column of A in increasing row order -- the serial loop over the CSC column, reproduced bit for bit."""
    import scipy.sparse as sp
    rng = np.random.default_rng(17)
    if case == "kron":
        A = oracle.kron_unsymmetric(7)
        S = A.to_scipy().tocsr()
    elif case == "rect":
        S = sp.random(230, 517, density=0.03, random_state=5, format="csr")
    elif case == "random":
        S = (sp.random(400, 400, density=0.05, random_state=9, format="csr") + sp.eye(400, format="csr")).tocsr()
    else:
        S = sp.random(64, 300, density=0.01, random_state=2, format="csr")     # many empty columns -> empty rows of A'
    S.sort_indices()
    m, n = S.shape
    dA = K.CsrMatrix.from_host(ctx, S.indptr.astype(np.int64), S.indices.astype(np.int32), S.data, (m, n))
    dT = dA.transpose()
    assert dT.shape == (n, m) and dT.nnz == dA.nnz
    x = rng.standard_normal(m)
    y = dT.matvec(ctx.array(x), ctx.empty(n)).to_host()
    # reference: serial loop over the columns of A in row order = rows of the (sorted) CSR of A'
    St = S.T.tocsr()
    St.sort_indices()
    ref = oracle.CsrMatrix.from_arrays(St.indptr.astype(np.int64), St.indices.astype(np.int32), St.data.copy()) if m == n else None
    y_ref = np.zeros(n)
    for j in range(n):
        acc = 0.0
        for q in range(St.indptr[j], St.indptr[j + 1]):
            acc = acc + St.data[q] * x[St.indices[q]]
        y_ref[j] = acc
    assert np.array_equal(y, y_ref)
    if ref is not None:
        assert np.array_equal(y, ref.matvec(x))
    # (A')' == A
    dTT = dT.transpose()
    z = rng.standard_normal(n)
    assert np.array_equal(dTT.matvec(ctx.array(z), ctx.empty(m)).to_host(), dA.matvec(ctx.array(z), ctx.empty(m)).to_host())


def test_spmv_fuzz_all_kernels_bit_exact(K, ctx, oracle):
    """Random shapes through every CSR kernel: empty rows, ragged rows, rows longer than the staged kernel's LDS
    window, rectangular operators, a single row.  y must equal the serial loop bit for bit (kernels 1, 3, 4, 5),
    and the fused dot must agree to an ulp of its condition."""
    import scipy.sparse as sp
    rng = np.random.default_rng(2024)
    cases = []
    for (m, n, mean, heavy) in [(1, 1, 1, 0), (3, 700, 5, 0), (257, 257, 1, 0), (1000, 999, 3, 0), (513, 2000, 7, 0),
                                (900, 900, 8, 3), (300, 300, 20, 2), (2049, 64, 6, 0), (64, 5000, 2, 1)]:
        rows, cols, vals = [], [], []
        for i in range(m):
            k = int(rng.poisson(mean)) if rng.random() > 0.1 else 0                 # 10 % empty rows
            if heavy and i % 97 == heavy:
                k = min(n, 2500 if heavy == 1 else 300)                             # longer than one 2048-entry window
            k = min(k, n)
            cs = np.sort(rng.choice(n, size=k, replace=False))
            rows += [i] * k; cols += list(cs); vals += list(rng.standard_normal(k))
        S = sp.csr_matrix((vals, (rows, cols)), shape=(m, n))
        S.sort_indices()
        cases.append(S)
    defaults = {k: ctx.get_option(k) for k in ("spmv_kernel", "spmv_lanes", "spmv_rows", "spmv_cap")}
    try:
        for S in cases:
            m, n = S.shape
            dA = K.CsrMatrix.from_host(ctx, S.indptr.astype(np.int64), S.indices.astype(np.int32), S.data, (m, n))
            x = rng.standard_normal(n)
            y_ref = np.zeros(m)
            for i in range(m):
                acc = 0.0
                for q in range(S.indptr[i], S.indptr[i + 1]):
                    acc = acc + S.data[q] * x[S.indices[q]]
                y_ref[i] = acc
            dx = ctx.array(x)
            for kern in (0, 1, 3, 4, 6):
                ctx.set_option("spmv_kernel", kern)
                dy = ctx.zeros(m)
                dA.matvec(dx, dy)
                assert np.array_equal(dy.to_host(), y_ref), (S.shape, S.nnz, kern)
            for cap in (256, 512):                              # wave-private windows smaller than the row groups: several windows per group
                ctx.set_option("spmv_kernel", 6)
                ctx.set_option("spmv_cap", cap)
                dy = ctx.zeros(m)
                dA.matvec(dx, dy)
                assert np.array_equal(dy.to_host(), y_ref), (S.shape, S.nnz, "wave kernel", cap)
                if m == n:
                    dyw = ctx.zeros(m)
                    dw = K.spmv_dot(dA, dx, dyw)
                    assert np.array_equal(dyw.to_host(), y_ref)
                    assert abs(dw - float(np.dot(x, y_ref))) <= 1e-13 * float(np.abs(x * y_ref).sum()) + 1e-300
            ctx.set_option("spmv_cap", 0)
            ctx.set_option("spmv_kernel", 0)
            if m == n:
                d = K.spmv_dot(dA, dx, ctx.empty(m))
                assert abs(d - float(np.dot(x, y_ref))) <= 1e-13 * float(np.abs(x * y_ref).sum()) + 1e-300
            T = dA.compress()                              # random values: not compressible, must stay correct
            dy = ctx.zeros(m)
            dA.matvec(dx, dy)
            assert np.array_equal(dy.to_host(), y_ref), (S.shape, "after compress", T)
    finally:
        for k, v in defaults.items():
            ctx.set_option(k, v)


def test_spmv_block_delta_columns_bit_exact(K, ctx, oracle):
    """The stream kernel's block-delta column stream (csrc/coldelta.hip: col = base[block] + 1- or 2-byte code, the entries that
    do not fit as (position, int32 column) escapes) and its 16-byte-load form on plain int32 columns: y equals the oracle's
    serial loop bit for bit on the non-stencil operators -- the banded + random operator (3 long-range links per row: escapes
    with 8 and with 16 bits), its nonsymmetric variant with rows of 3000 entries (blocks of several windows, more than 256
    escapes per block), random matrices (nearly everything escapes), empty rows, a rectangular operator -- with and without
    the fused dots; the bytes the kernel streams are what khip_spmv_bytes_stored reports."""
    import scipy.sparse as sp
    rng = np.random.default_rng(4242)
    cases = []
    for kw in (dict(n=20000, seed=3), dict(n=30000, seed=5, unsym=True, dense_rows=3), dict(n=5000, half_band=40, links=1, seed=9)):
        A = oracle.banded_random(**kw)
        cases.append((A.rowptr.copy(), A.col.copy(), A.val.copy(), (A.n, A.n), "banded_random %r" % kw))
    for (m, n, dens) in ((3000, 3000, 0.006), (2500, 4100, 0.008), (700, 90000, 0.0004)):
        S = sp.random(m, n, density=dens, random_state=int(rng.integers(1 << 30)), format="csr")
        S.sort_indices()
        cases.append((S.indptr.astype(np.int64), S.indices.astype(np.int32), S.data.copy(), (m, n), "random %dx%d" % (m, n)))
    saved = {k: ctx.get_option(k) for k in ("spmv_kernel", "spmv_delta", "spmv_wide", "spmv_codes")}
    try:
        ctx.set_option("spmv_kernel", 1)
        ctx.set_option("spmv_codes", 0)
        for rp, ci, va, (m, n), name in cases:
            nnz = int(rp[-1])
            x = _vec(rng, n)
            y_ref = oracle.CsrMatrix.from_arrays(rp, ci, va).matvec(x) if m == n else None
            if y_ref is None:
                y_ref = np.zeros(m)
                for i in range(m):
                    acc = 0.0
                    for q in range(rp[i], rp[i + 1]):
                        acc = acc + va[q] * x[ci[q]]
                    y_ref[i] = acc
            dx = ctx.array(x)
            seen = set()
            for delta, wide in ((0, 0), (0, 1), (8, 1), (16, 1), (2, 1), (1, 1)):
                ctx.set_option("spmv_delta", delta)
                ctx.set_option("spmv_wide", wide)
                dA = K.CsrMatrix.from_host(ctx, rp, ci, va, (m, n))
                assert dA.delta_info == (32, 0, 0)                                     # nothing is built before the first product
                dy = ctx.zeros(m)
                dA.matvec(dx, dy)
                assert np.array_equal(dy.to_host(), y_ref), (name, delta, wide)
                bits, rows, esc = dA.delta_info
                seen.add(bits)
                if delta in (8, 16):
                    assert bits == delta and rows in (32, 64, 128, 256) and 0 <= esc <= nnz, (name, dA.delta_info)
                    blocks = (m + rows - 1) // rows
                    assert dA.spmv_bytes_stored == (8 + bits // 8) * nnz + 6 * esc + 8 * blocks + 4 * (m + 1) + 8 * n + 8 * m
                    # the escapes are exactly the entries outside [base, base + 2^bits - 2] of their block
                    r_of = np.repeat(np.arange(m), np.diff(rp))
                    base = np.maximum(0, (r_of // rows) * rows - ((1 << bits) - 1 - rows) // 2)
                    assert esc == int(np.count_nonzero((ci < base) | (ci - base >= (1 << bits) - 1))), name
                elif delta in (0, 1):
                    assert bits == 32                                                  # 1: only operators of >= 4 M entries
                    assert dA.spmv_bytes_stored == 12 * nnz + 4 * (m + 1) + 8 * n + 8 * m
                if m == n:
                    dy2 = ctx.zeros(m)
                    d = K.spmv_dot(dA, dx, dy2)
                    assert np.array_equal(dy2.to_host(), y_ref), (name, delta, "fused dot")
                    d_cpu = oracle.dot(x, y_ref)
                    assert abs(d - d_cpu) <= 2 * EPS * abs(d_cpu) + 1e-16 * float(np.abs(x * y_ref).sum())
                    dy3 = ctx.zeros(m)
                    d2 = K.spmv_dot2(dA, dx, dy3)                                      # x.y and y.y from one product
                    assert np.array_equal(dy3.to_host(), y_ref)
                    yy = oracle.dot(y_ref, y_ref)
                    assert abs(d2[0] - d_cpu) <= 2 * EPS * abs(d_cpu) + 1e-16 * float(np.abs(x * y_ref).sum())
                    assert abs(d2[1] - yy) <= 4 * EPS * yy
            assert 8 in seen and 16 in seen and 32 in seen
    finally:
        for k, v in saved.items():
            ctx.set_option(k, v)


@pytest.mark.parametrize("kind,n1,bits,diags", [("poisson", 20, 8, 7), ("kron_unsymmetric", 12, 8, 7), ("stencil27", 9, 8, 27)])
def test_spmv_coded_columns_bit_exact(K, ctx, oracle, kind, n1, bits, diags):
    """The staged kernel's dictionary-coded column stream (csrc/colcode.hip: one byte per entry = the rank of the
    entry's diagonal) gives the y of the int32 stream and of the oracle's serial loop bit for bit, with and without the
    fused dots; forced two-byte codes and the switch `spmv_codes = 0` too."""
    gen = {"poisson": oracle.poisson3d, "kron_unsymmetric": oracle.kron_unsymmetric, "stencil27": oracle.stencil27_unsym}[kind]
    A = gen(n1)
    rng = np.random.default_rng(77)
    x = _vec(rng, A.n)
    y_ref = A.matvec(x)
    dx = ctx.array(x)
    saved = {k: ctx.get_option(k) for k in ("spmv_codes", "spmv_kernel", "spmv_sell")}
    try:
        ctx.set_option("spmv_kernel", 4)            # the 27-point operator would take the ordered kernel by default
        ctx.set_option("spmv_sell", 0)              # the coded CSR stream itself (the sliced form: test_spmv_sliced_form_bit_exact)
        ctx.set_option("spmv_codes", 1)             # the default: only operators of >= 4 M entries get the coded stream
        dS = K.CsrMatrix.stencil(ctx, kind, n1)
        dS.matvec(dx, ctx.zeros(A.n))
        assert dS.code_info == (32, 0)
        for mode, want_bits in ((2, bits), (16, 16), (0, 32)):
            ctx.set_option("spmv_codes", mode)
            dA = K.CsrMatrix.stencil(ctx, kind, n1)
            assert dA.code_info == (32, 0)                              # nothing is built before the first product
            dy = ctx.zeros(A.n)
            dA.matvec(dx, dy)
            assert dA.code_info == ((want_bits, diags) if mode else (32, 0))
            assert np.array_equal(dy.to_host(), y_ref), (kind, mode)
            dy2 = ctx.zeros(A.n)
            d = K.spmv_dot(dA, dx, dy2)
            assert np.array_equal(dy2.to_host(), y_ref)
            d_cpu = oracle.dot(x, y_ref)
            assert abs(d - d_cpu) <= 2 * EPS * abs(d_cpu) + 1e-16 * float(np.abs(x * y_ref).sum())
            stored = dA.spmv_bytes_stored
            want = (8 + (want_bits // 8)) * A.nnz + 4 * (A.n + 1) + 16 * A.n
            assert stored == want and dA.spmv_bytes == 12 * A.nnz + 4 * (A.n + 1) + 16 * A.n
    finally:
        for k, v in saved.items():
            ctx.set_option(k, v)


@pytest.mark.parametrize("kind,n1", [("poisson", 20), ("poisson", 37), ("kron_unsymmetric", 12), ("stencil27", 9), ("stencil27", 21)])
def test_spmv_sliced_form_bit_exact(K, ctx, oracle, kind, n1):
    """The sliced form of a coded operator (csrc/colcode.hip csr_build_sell + spmv_sell_kernel: every 64 rows transposed, a lane
    loads its own row's codes and values with coalesced 8-byte loads; ctx option spmv_sell = 1 / 2, 2 the default): y equals the
    oracle's serial loop bit for bit, the fused dots equal the coded CSR kernel's bit for bit (same rows per lane, same partials),
    for 1 / 2 / 3 row blocks per workgroup, uniform and per-slice-offset layouts, row counts that are no multiple of 64.  (Rows too long
    for 256 of them in the coded kernel's LDS window -- the 27-point operator -- keep 256-row blocks in the sliced form, which has no
    window: other partials there, the dots agree to rounding.)"""
    gen = {"poisson": oracle.poisson3d, "kron_unsymmetric": oracle.kron_unsymmetric, "stencil27": oracle.stencil27_unsym}[kind]
    A = gen(n1)
    rng = np.random.default_rng(78)
    x = _vec(rng, A.n)
    y_ref = A.matvec(x)
    dx = ctx.array(x)
    saved = {k: ctx.get_option(k) for k in ("spmv_codes", "spmv_kernel", "spmv_sell", "spmv_tiles", "spmv_blk_pub", "spmv_sell_narrow", "spmv_sell_pair")}
    try:
        ctx.set_option("spmv_kernel", 4); ctx.set_option("spmv_codes", 2)
        ctx.set_option("spmv_sell_narrow", 1)          # the 4-bit code words where the operator allows them (off by default: slower)
        ref = {}
        layouts = set()
        same_blocks = 256.0 * A.nnz / A.n <= 2048.0          # both forms walk 256-row blocks (spmv.hip launch_spmv)
        for sell in (0, 1, 2):
            for tiles in (1, 2, 3):
                for pub in (1, 0):
                    ctx.set_option("spmv_sell", sell); ctx.set_option("spmv_tiles", tiles); ctx.set_option("spmv_blk_pub", pub)
                    dA = K.CsrMatrix.stencil(ctx, kind, n1)
                    assert dA.sell_info == (0, 0, 0)                        # nothing is built before the first product
                    dy = ctx.zeros(A.n)
                    dA.matvec(dx, dy)
                    assert np.array_equal(dy.to_host(), y_ref), (kind, sell, tiles, pub)
                    st, upl, total = dA.sell_info
                    diags = dA.code_info[1]
                    assert st == (1 if sell else 0) and dA.code_info[0] == 8
                    dy2 = ctx.zeros(A.n)
                    d = K.spmv_dot(dA, dx, dy2)
                    d2 = K.spmv_dot2(dA, dx, dy2)
                    assert np.array_equal(dy2.to_host(), y_ref)
                    key = (tiles, pub)
                    if sell == 0: ref[key] = (d, d2)
                    elif same_blocks: assert (d, d2) == ref[key], (kind, sell, tiles, pub)     # the SAME partials: bit-identical reductions
                    else:             # longer rows: the coded kernel's LDS window takes 64-row blocks, the sliced form 256 -- other partials
                        for got, want in zip((d, d2[0], d2[1]), (ref[key][0],) + tuple(ref[key][1])):
                            assert abs(got - want) <= 4 * EPS * abs(want)
                    slices = (A.n + 63) // 64
                    if sell:
                        layouts.add(upl > 0)
                        assert total == (upl * slices if upl else total) and total >= slices
                        narrow = diags <= 15 and int(np.diff(A.rowptr).max()) <= 8          # eight 4-bit codes per row in one word
                        assert dA.sell_narrow == narrow
                        want = 512 * total + (0 if upl else 4 * (slices + 1)) + (256 * slices if narrow else 0) + 16 * A.n
                    else:
                        want = 9 * A.nnz + 4 * (A.n + 1) + 16 * A.n
                    assert dA.spmv_bytes_stored == want and dA.spmv_bytes == 12 * A.nnz + 4 * (A.n + 1) + 16 * A.n
        # narrow codes off: the byte-coded words, same results
        ctx.set_option("spmv_codes", 2); ctx.set_option("spmv_sell", 2); ctx.set_option("spmv_tiles", 0); ctx.set_option("spmv_blk_pub", 1)
        got = []
        for nar, pair in ((1, 1), (0, 1), (0, 0)):         # 4-bit code words | the row's words in 16-byte pairs (the default) | plain 8-byte words
            ctx.set_option("spmv_sell_narrow", nar); ctx.set_option("spmv_sell_pair", pair)
            dN = K.CsrMatrix.stencil(ctx, kind, n1)
            dy = ctx.zeros(A.n); dN.matvec(dx, dy)
            assert np.array_equal(dy.to_host(), y_ref) and (not dN.sell_narrow or nar == 1) and dN.sell_info[0] == 1
            got.append((K.spmv_dot(dN, dx, dy), K.spmv_dot2(dN, dx, dy)))
        assert got[0] == got[1] == got[2]
        ctx.set_option("spmv_sell_narrow", 0); ctx.set_option("spmv_sell_pair", 1)
        # the int32 column stream (spmv_codes = 0): its sliced form (two columns per word) against the staged CSR kernel
        ctx.set_option("spmv_codes", 0); ctx.set_option("spmv_tiles", 0); ctx.set_option("spmv_blk_pub", 1)
        ref32 = None
        for sell, pair in ((0, 1), (3, 1), (3, 2), (1, 1)):     # sell 3: whatever the size (the default takes operators of >= 4 M entries); pair 2: the int32 form in 16-byte pairs too
            ctx.set_option("spmv_sell", sell); ctx.set_option("spmv_sell_pair", pair)
            dC = K.CsrMatrix.stencil(ctx, kind, n1)
            dy = ctx.zeros(A.n); dC.matvec(dx, dy)
            assert np.array_equal(dy.to_host(), y_ref), (kind, "int32", sell)
            dd_ = (K.spmv_dot(dC, dx, dy), K.spmv_dot2(dC, dx, dy))
            ref32 = dd_ if ref32 is None else ref32
            assert dd_ == ref32 and dC.code_info == (32, 0) and dC.sell_info[0] == 0
            taken = sell == 3 and int(np.diff(A.rowptr).max()) <= 8   # the staged kernel with 256-row blocks (256 x longest row <= 2048)
            assert dC.sell32_info[0] == (1 if taken else 0), (kind, sell, dC.sell32_info)
            if taken:
                st, upl, total = dC.sell32_info
                assert dC.spmv_bytes_stored == 512 * total + (0 if upl else 4 * ((A.n + 63) // 64 + 1)) + 16 * A.n
            else:
                assert dC.spmv_bytes_stored == 12 * A.nnz + 4 * (A.n + 1) + 16 * A.n
        ctx.set_option("spmv_sell_pair", 1)
        # an operator the sliced form does not take (two-byte codes): the coded CSR stream, silently
        ctx.set_option("spmv_sell", 2); ctx.set_option("spmv_codes", 16); ctx.set_option("spmv_tiles", 0); ctx.set_option("spmv_blk_pub", 1)
        dB = K.CsrMatrix.stencil(ctx, kind, n1)
        dy = ctx.zeros(A.n); dB.matvec(dx, dy)
        assert np.array_equal(dy.to_host(), y_ref) and dB.sell_info[0] == 0 and dB.code_info[0] == 16
    finally:
        for k, v in saved.items():
            ctx.set_option(k, v)


@pytest.mark.parametrize("kernel,codes", [(4, 2), (4, 0), (1, 0)])
def test_spmv_fused_dots_with_workgroup_level_publish(K, ctx, oracle, kernel, codes):
    """spmv_blk_pub = 1 (block_publish, csrc/device_reduce.hpp: the four waves of a tile fold their lanes' double-double
    partials through LDS and one wave runs the tree) and spmv_tiles = 1 / 2 / 3 on the staged, coded and 16-byte-load stream
    kernels: y bit-identical, the fused dots within the usual bound and equal to the per-wave form's to an ulp."""
    A = oracle.stencil27_unsym(11)
    rng = np.random.default_rng(5)
    x = _vec(rng, A.n)
    y_ref = A.matvec(x)
    d_cpu, yy = oracle.dot(x, y_ref), oracle.dot(y_ref, y_ref)
    saved = {k: ctx.get_option(k) for k in ("spmv_kernel", "spmv_codes", "spmv_blk_pub", "spmv_tiles", "spmv_wide", "spmv_delta")}
    try:
        ctx.set_option("spmv_kernel", kernel); ctx.set_option("spmv_codes", codes)
        ctx.set_option("spmv_wide", 1); ctx.set_option("spmv_delta", 0)
        dx = ctx.array(x)
        got = {}
        for pub in (0, 1):
            for tiles in (1, 2, 3):
                ctx.set_option("spmv_blk_pub", pub); ctx.set_option("spmv_tiles", tiles)
                dA = K.CsrMatrix.stencil(ctx, "stencil27", 11)
                dy = ctx.zeros(A.n)
                d = K.spmv_dot(dA, dx, dy)
                assert np.array_equal(dy.to_host(), y_ref), (pub, tiles)
                assert abs(d - d_cpu) <= 2 * EPS * abs(d_cpu) + 1e-16 * float(np.abs(x * y_ref).sum())
                d2 = K.spmv_dot2(dA, dx, dy)
                assert np.array_equal(dy.to_host(), y_ref)
                assert abs(d2[0] - d_cpu) <= 2 * EPS * abs(d_cpu) + 1e-16 * float(np.abs(x * y_ref).sum()) and abs(d2[1] - yy) <= 4 * EPS * yy
                got[(pub, tiles)] = (d, d2[0], d2[1])
        ref = got[(0, 1)]
        for k, v in got.items():
            assert all(abs(a - b) <= EPS * abs(b) for a, b in zip(v, ref)), (k, v, ref)
    finally:
        for k, v in saved.items():
            ctx.set_option(k, v)


def test_spmv_coded_columns_general_matrices(K, ctx):
    """Matrices whose entries lie on few / many / too many diagonals: 8-bit codes, 16-bit codes, and the int32 stream
    (more than 2048 distinct column - row offsets); empty rows, a rectangular operator, rows wider than one LDS window.
    y equals the serial loop bit for bit in every case."""
    import scipy.sparse as sp
    rng = np.random.default_rng(99)

    def banded(m, n, offs, drop=0.2):
        rows, cols = [], []
        for i in range(m):
            for d in offs:
                j = i + d
                if 0 <= j < n and rng.random() > drop:
                    rows.append(i); cols.append(j)
        vals = rng.standard_normal(len(rows))
        S = sp.csr_matrix((vals, (rows, cols)), shape=(m, n)); S.sort_indices()
        return S
    cases = [
        (banded(3000, 3000, [-700, -3, -1, 0, 1, 2, 900]), 8),
        (banded(2000, 2600, list(range(-150, 151, 1))[::50] + [400, 599]), 8),           # rectangular
        (banded(1500, 1500, sorted(set(int(v) for v in rng.integers(-1400, 1400, size=300))), drop=0.98), 16),
        (sp.random(4000, 4000, density=0.0015, random_state=5, format="csr"), 32),        # ~4000+ distinct offsets
    ]
    saved = ctx.get_option("spmv_kernel")
    try:
        ctx.set_option("spmv_kernel", 4)
        for S, want_bits in cases:
            S.sort_indices()
            m, n = S.shape
            dA = K.CsrMatrix.from_host(ctx, S.indptr.astype(np.int64), S.indices.astype(np.int32), S.data, (m, n))
            x = rng.standard_normal(n)
            y_ref = np.zeros(m)
            for i in range(m):
                acc = 0.0
                for q in range(S.indptr[i], S.indptr[i + 1]):
                    acc = acc + S.data[q] * x[S.indices[q]]
                y_ref[i] = acc
            dy = ctx.zeros(m)
            dA.matvec(ctx.array(x), dy)
            assert dA.code_info[0] == want_bits, (S.shape, dA.code_info)
            assert np.array_equal(dy.to_host(), y_ref), (S.shape, dA.code_info)
    finally:
        ctx.set_option("spmv_kernel", saved)


def test_csr_create_rejects_malformed_arrays(K, ctx):
    """khip_csr_create validates what every kernel afterwards trusts: monotone row pointers from 0 to nnz,
    columns inside [0, n), 1-based arrays passed as 0-based, 64-bit row pointers that do not fit a shard."""
    import ctypes as C
    rp = np.array([0, 2, 3, 5], dtype=np.int64)
    cl = np.array([0, 1, 1, 0, 2], dtype=np.int32)
    vl = np.ones(5)
    K.CsrMatrix.from_host(ctx, rp, cl, vl, (3, 3))                                   # well-formed
    for bad_rp, bad_cl, why in [
            (np.array([0, 3, 2, 5], dtype=np.int64), cl, "row pointers"),            # decreasing
            (np.array([1, 2, 3, 5], dtype=np.int64), cl, "row pointers"),            # does not start at 0
            (np.array([0, 2, 3, 4], dtype=np.int64), cl, "row pointers"),            # does not end at nnz
            (rp, np.array([0, 1, 1, 0, 3], dtype=np.int32), "column index"),         # column == n
            (rp, np.array([0, -1, 1, 0, 2], dtype=np.int32), "column index"),        # negative column
            (rp + 1, cl + 1, "row pointers"),                                        # 1-based arrays declared 0-based
            (np.array([0, 2, 3, 2 ** 40], dtype=np.int64), cl, "int32")]:
        with pytest.raises(K.KhipError) as ei:
            K.CsrMatrix.from_host(ctx, bad_rp, bad_cl, vl, (3, 3))
        assert why in str(ei.value), (why, str(ei.value))
