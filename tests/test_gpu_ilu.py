"""ILU(0) / IC(0) preconditioner on the device (csrc/ilu.hip) against the CPU oracle.

The reference builds this operator with the vendor library (ic02 / ilu02 + triangular ldiv!, docs/src/gpu.md:74-163)
and passes it as M to cg!/bicgstab!; its known answer: IC(0)-CG on sparse_laplacian(16) needs <= 19 iterations
(test/gpu/nvidia.jl:37-70).  Factors and solves walk each row in stored order => bit-identical to the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _upload(K, ctx, A):
    return K.CsrMatrix.from_host(ctx, A.rowptr, A.col, A.val, (A.n, A.n))


@pytest.mark.parametrize("gen,arg", [("poisson3d", 7), ("kron_unsymmetric", 5), ("stencil27_unsym", 5), ("tridiag", 300)])
@pytest.mark.parametrize("graph", [True, False])
def test_ilu0_factor_and_solve_bit_identical(K, ctx, oracle, gen, arg, graph):
    A = oracle.tridiag(arg, -1.0, 2.5, -1.5) if gen == "tridiag" else getattr(oracle, gen)(arg)
    ref = oracle.Ilu0(A)
    dA = _upload(K, ctx, A)
    P = K.Ilu0(dA, graph=graph)
    assert np.array_equal(P.values(), ref.lu)
    rng = np.random.default_rng(arg)
    for _ in range(3):                      # the cached graph is replayed / re-captured for new pointers
        x = rng.standard_normal(A.n)
        dx, dy = ctx.array(x), ctx.empty(A.n)
        P(dx, dy)
        assert np.array_equal(dy.to_host(), ref.solve(x))
        P(dx, dy)
        assert np.array_equal(dy.to_host(), ref.solve(x))
    lo, up = P.levels
    if gen == "poisson3d":
        assert lo == up == 3 * arg - 2      # hyperplanes i + j + k = const
    if gen == "tridiag":
        assert lo == up == arg              # a chain: no parallelism at all, still correct


def test_ic0_cg_reference_known_answer(K, ctx, oracle, parity_log):
    """test/gpu/nvidia.jl:37-70: x, stats = cg(A_gpu, b_gpu, M=opM) with opM = IC(0) of sparse_laplacian(16):
    norm(b - A x) <= 1e-6 and stats.niter <= 19."""
    A = oracle.poisson3d(16)
    b = np.ones(A.n)
    dA = _upload(K, ctx, A)
    P = K.Ilu0(dA)
    refP = oracle.Ilu0(A)
    ref = oracle.cg(A, b, M=lambda v: refP.solve(v), history=True)
    for fused in (0, 1, 2):
        x, st, _ = K.cg(dA, ctx.array(b), M=P, history=True, fused=fused)
        xh = x.to_host()
        assert st.solved and st.niter <= 19 and st.niter == ref.niter
        assert np.linalg.norm(b - A.matvec(xh)) <= 1e-6
        dev = float(np.max(np.abs(st.residuals - ref.residuals) / ref.residuals))
        assert dev <= 1e-10
        assert np.allclose(xh, ref.x, rtol=0, atol=1e-10 * np.abs(ref.x).max())
    parity_log(test="ic0_cg_laplacian16", niter=st.niter, hist_max_rel=dev)
    plain = K.cg(dA, ctx.array(b))[1]
    assert plain.niter == 38


def test_ilu0_bicgstab_and_gmres(K, ctx, oracle, parity_log):
    """docs/src/gpu.md:118-163: bicgstab(A_gpu, b_gpu, M=opM) with opM = ILU(0); same operator as right
    preconditioner N of gmres."""
    A = oracle.kron_unsymmetric(8)
    rng = np.random.default_rng(5)
    bh = A.matvec(rng.standard_normal(A.n))
    dA = _upload(K, ctx, A)
    P = K.Ilu0(dA)
    refP = oracle.Ilu0(A)
    ref = oracle.bicgstab(A, bh, M=lambda v: refP.solve(v), history=True)
    x, st, _ = K.bicgstab(dA, ctx.array(bh), M=P, history=True)
    assert st.solved and st.niter == ref.niter
    assert np.linalg.norm(bh - A.matvec(x.to_host())) <= 1e-6 * np.linalg.norm(bh)
    plain = K.bicgstab(dA, ctx.array(bh))[1]
    assert st.niter < plain.niter
    refg = oracle.gmres(A, bh, N=lambda v: refP.solve(v), history=True)
    x, stg, _ = K.gmres(dA, ctx.array(bh), N=P, history=True)
    assert stg.solved and stg.niter == refg.niter
    assert np.max(np.abs(stg.residuals - refg.residuals) / refg.residuals[0]) <= 1e-10
    parity_log(test="ilu0_bicgstab_gmres", bicgstab_niter=st.niter, bicgstab_plain=plain.niter, gmres_niter=stg.niter)


def test_ilu0_errors(K, ctx, oracle):
    Z = oracle.tridiag(4, 1.0, 0.0, 1.0)                        # explicit zero diagonal -> zero pivot
    with pytest.raises(K.KhipError) as e:
        K.Ilu0(_upload(K, ctx, Z))
    assert "pivot" in str(e.value)
    # structurally missing diagonal
    rowptr = np.array([0, 1, 2], dtype=np.int64)
    col = np.array([1, 0], dtype=np.int32)
    val = np.array([1.0, 1.0])
    with pytest.raises(K.KhipError) as e:
        K.Ilu0(K.CsrMatrix.from_host(ctx, rowptr, col, val, (2, 2)))
    assert "diagonal" in str(e.value)
    # unsorted columns are refused (the merge in the factorisation needs them sorted)
    rowptr = np.array([0, 2, 4], dtype=np.int64)
    col = np.array([1, 0, 0, 1], dtype=np.int32)
    val = np.array([1.0, 4.0, 1.0, 4.0])
    with pytest.raises(K.KhipError) as e:
        K.Ilu0(K.CsrMatrix.from_host(ctx, rowptr, col, val, (2, 2)))
    assert "sorted" in str(e.value)


@pytest.mark.parametrize("n1,world", [(12, 3), (24, 2)])
def test_block_jacobi_ilu0_on_distributed_handle(K, oracle, n1, world):
    """A distributed handle factors its owned diagonal block only (ghost columns ignored): the result equals
    ILU(0) of that block computed by the oracle, and CG with this block-Jacobi IC(0) converges on every rank
    to the same solution as the unpreconditioned solve.  24^3 on two ranks: slabs of 12 planes, >= 4096 rows each, so the
    solves take the block schedule on the renumbered [owned | ghost] slab."""
    import threading
    A_cpu = oracle.poisson3d(n1)
    n = A_cpu.n
    starts = K.row_partition(n, world)
    S = A_cpu.to_scipy().tocsr()
    out = [None] * world
    errs = []

    def run(rank):
        try:
            c = K.Context(0)
            c.comm_init_local(rank, world, 4242 + world)
            r0, r1 = starts[rank], starts[rank + 1]
            A = K.CsrMatrix.stencil(c, "poisson", n1, rows=(r0, r1), distributed=True)
            P = K.Ilu0(A)
            if n1 == 24:
                dims, nb, failed = P.block_info()
                assert dims == (24, 24, 12) and nb > 0 and failed == 0, (dims, nb, failed)
            x = np.linspace(1, 2, r1 - r0)
            y = P(c.array(x), c.empty(r1 - r0)).to_host()
            b = c.empty(r1 - r0)
            K.kfill_(b, 1.0)
            xs, st, _ = K.cg(A, b, M=P, rtol=1e-10)
            out[rank] = (y, xs.to_host(), st.niter, st.solved)
            c.close()
        except Exception as e:      # noqa: BLE001
            errs.append((rank, repr(e)))
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(timeout=120) for t in ts]
    assert not errs, errs
    x_ref = oracle.cg(A_cpu, np.ones(n), rtol=1e-12).x
    for rank in range(world):
        r0, r1 = starts[rank], starts[rank + 1]
        blk = S[r0:r1, r0:r1].tocsr()
        blk.sort_indices()
        B = oracle.CsrMatrix.from_arrays(blk.indptr.astype(np.int64), blk.indices.astype(np.int32), blk.data.copy()) \
            if hasattr(oracle.CsrMatrix, "from_arrays") else None
        y, xs, niter, solved = out[rank]
        if B is not None:
            assert np.array_equal(y, oracle.Ilu0(B).solve(np.linspace(1, 2, r1 - r0)))
        assert solved and np.allclose(xs, x_ref[r0:r1], atol=1e-7)
    assert len({o[2] for o in out}) == 1          # every rank ran the same number of iterations


def test_ilu0_chain_operator_uses_batched_small_levels(K, ctx, oracle):
    """A tridiagonal matrix is one dependency chain: n levels of one row.  Runs of small levels are executed by one
    workgroup with barriers in between (one launch per triangle instead of n): still bit-identical, and not n launches slow."""
    import time
    n = 20000
    A = oracle.tridiag(n, -1.0, 2.5, -1.5)
    ref = oracle.Ilu0(A)
    P = K.Ilu0(_upload(K, ctx, A))
    assert P.levels == (n, n)
    assert np.array_equal(P.values(), ref.lu)
    x = np.linspace(-1.0, 1.0, n)
    dx, dy = ctx.array(x), ctx.empty(n)
    P(dx, dy)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        P(dx, dy)
    ctx.sync()
    dt = (time.perf_counter() - t0) / 3
    assert np.array_equal(dy.to_host(), ref.solve(x))
    assert dt < 0.1, dt          # 2 n per-level launches would take ~0.2 s


@pytest.mark.parametrize("gen,args", [("poisson3d", (16,)), ("poisson3d", (19,)), ("poisson3d", (23, 17, 12)), ("kron_unsymmetric", (17,)),
                                       ("poisson3d", (70, 64, 1)), ("stencil27_unsym", (18,)), ("stencil27_unsym", (21,))])
def test_ilu0_block_schedule_bit_identical(K, ctx, oracle, gen, args):
    """Structured grids (>= 4096 rows) take the block schedule (csrc/ilu.hip: ilu_block_solve_kernel): same y, bit for bit,
    as the level-scheduled kernels and as the oracle's serial loops -- whole and partial 8 x 8 x 8 blocks, a 2-D grid, an
    unsymmetric pattern, and the 27-point stencil, whose lower triangle reaches (x + 1, y - 1, z) and (x + 1, y + 1, z - 1): its
    blocks are cubes in the skewed basis (x + y + 2 z, y + z, z) and run the general path (13 entries per row)."""
    A = getattr(oracle, gen)(*args)
    ref = oracle.Ilu0(A)
    dA = _upload(K, ctx, A)
    P = K.Ilu0(dA)
    dims, nb, failed = P.block_info()
    assert dims[0] * dims[1] * dims[2] == A.n and nb > 0 and failed == 0, (dims, nb, failed)
    if gen == "poisson3d":
        want = tuple(args) if len(args) == 3 else (args[0],) * 3
        assert dims == want
    ctx.set_option("ilu_blocks", 0)
    try:
        Pl = K.Ilu0(dA)                      # level scheduling on the same handle
    finally:
        ctx.set_option("ilu_blocks", 1)
    assert Pl.block_info()[1] == 0
    assert np.array_equal(P.values(), ref.lu)
    rng = np.random.default_rng(len(args) + args[0])
    for _ in range(3):
        x = rng.standard_normal(A.n)
        dx, dy, dz = ctx.array(x), ctx.empty(A.n), ctx.empty(A.n)
        P(dx, dy); Pl(dx, dz)
        want_y = ref.solve(x)
        assert np.array_equal(dy.to_host(), want_y)
        assert np.array_equal(dz.to_host(), want_y)
        P(dx, dy)                            # again: the epochs and the ticket base move on
        assert np.array_equal(dy.to_host(), want_y)
    assert P.block_info()[2] == 0
    if gen in ("stencil27_unsym", "kron_unsymmetric"):          # the same on the packed entry lists (no row records)
        ctx.set_option("ilu_blocks", 2)
        try:
            Pp = K.Ilu0(dA)
        finally:
            ctx.set_option("ilu_blocks", 1)
        dx, dy = ctx.array(x), ctx.empty(A.n)
        Pp(dx, dy)
        assert np.array_equal(dy.to_host(), ref.solve(x))
        assert Pp.block_info()[1] > 0 and Pp.block_info()[2] == 0


@pytest.mark.parametrize("blocks", [1, 2])
def test_ilu0_block_schedule_without_a_grid(K, ctx, oracle, blocks):
    """A pattern that is no grid in natural ordering (a random symmetric permutation of one): the blocks are pieces of the
    level-sorted row sequence (dims 0, 0, 0 but blocks > 0) -- on row records and on the packed lists -- and y is still the
    oracle's, bit for bit.  A chain (levels of one row) keeps level scheduling: see the next test."""
    import scipy.sparse as sp
    A = oracle.poisson3d(16)
    S = A.to_scipy().tocsr()
    perm = np.random.default_rng(0).permutation(A.n)
    Sp = S[perm][:, perm].tocsr(); Sp.sort_indices()
    Ap = oracle.CsrMatrix.from_arrays(Sp.indptr.astype(np.int64), Sp.indices.astype(np.int32), Sp.data)
    dA = _upload(K, ctx, Ap)
    ctx.set_option("ilu_blocks", blocks)
    try:
        P = K.Ilu0(dA)
    finally:
        ctx.set_option("ilu_blocks", 1)
    dims, nb, failed = P.block_info()
    assert dims == (0, 0, 0) and nb > 0 and failed == 0, (dims, nb, failed)
    ref = oracle.Ilu0(Ap)
    assert np.array_equal(P.values(), ref.lu)
    for seed in range(3):
        x = np.random.default_rng(seed).standard_normal(A.n)
        dy = ctx.empty(A.n)
        P(ctx.array(x), dy)
        assert np.array_equal(dy.to_host(), ref.solve(x))
    assert P.block_info()[2] == 0
    T = oracle.tridiag(5000, -1.0, 2.5, -1.5)                  # a chain: levels of one row, no blocks
    Pt = K.Ilu0(_upload(K, ctx, T))
    assert Pt.block_info()[:2] == ((0, 0, 0), 0)


@pytest.mark.parametrize("records", [1, 2])
@pytest.mark.parametrize("dims", [(18, 17, 16), (70, 64, 1)])
def test_ilu0_block_schedule_general_rows(K, ctx, oracle, dims, records):
    """More than three entries per row and triangle (second neighbours along every axis: offsets 1, 2, n1, 2 n1, ...): still a
    grid with dependencies towards smaller coordinates, so the block schedule applies -- on the wide row records (rows
    with 4..16 entries; option ilu_blocks = 1) or on the packed entry lists every pattern can fall back to (= 2).
    Unsymmetric values."""
    import scipy.sparse as sp
    n1, n2, n3 = dims
    def band(n, lo2, lo1, di, up1, up2):
        return sp.diags([np.full(n - 2, lo2), np.full(n - 1, lo1), np.full(n, di), np.full(n - 1, up1), np.full(n - 2, up2)],
                        [-2, -1, 0, 1, 2], format="csr")
    I = lambda n: sp.identity(n, format="csr")
    S = sp.kron(sp.kron(I(n3), I(n2)), band(n1, -0.25, -1.0, 2.5, -1.5, -0.3)) + \
        sp.kron(sp.kron(I(n3), band(n2, -0.2, -1.1, 2.5, -0.9, -0.35)), I(n1))
    if n3 > 1:
        S = S + sp.kron(sp.kron(band(n3, -0.15, -1.2, 2.5, -0.8, -0.1), I(n2)), I(n1))
    S = S.tocsr(); S.sort_indices()
    A = oracle.CsrMatrix.from_arrays(S.indptr.astype(np.int64), S.indices.astype(np.int32), S.data)
    ref = oracle.Ilu0(A)
    ctx.set_option("ilu_blocks", records)
    try:
        P = K.Ilu0(_upload(K, ctx, A))
    finally:
        ctx.set_option("ilu_blocks", 1)
    got_dims, nb, failed = P.block_info()
    assert got_dims == dims and nb > 0 and failed == 0, (got_dims, nb, failed)
    assert np.array_equal(P.values(), ref.lu)
    rng = np.random.default_rng(n1)
    for _ in range(2):
        x = rng.standard_normal(A.n)
        dy = ctx.empty(A.n)
        P(ctx.array(x), dy)
        assert np.array_equal(dy.to_host(), ref.solve(x))
    assert P.block_info()[2] == 0
