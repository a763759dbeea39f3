"""Row-template compressed handles (csrc/template.hip): same operator, 2 bytes of matrix data per row.
Everything must be BIT-IDENTICAL to the CSR kernels and to the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _upload(K, ctx, A):
    return K.CsrMatrix.from_host(ctx, A.rowptr, A.col, A.val, (A.n, A.n))


@pytest.mark.parametrize("gen,arg,templates", [("poisson3d", (9, 7, 5), 27), ("poisson3d", (16, 16, 16), 27),
                                               ("kron_unsymmetric", (6,), None), ("stencil27_unsym", (6,), None),
                                               ("poisson3d", (5, 1, 1), 3)])
def test_compressed_spmv_bit_identical(K, ctx, oracle, gen, arg, templates):
    A = getattr(oracle, gen)(*arg)
    dA = _upload(K, ctx, A)
    T = dA.compress()
    assert T > 0 and (templates is None or T == templates)
    assert dA.spmv_bytes_stored == 2 * A.n + 16 * A.n and dA.spmv_bytes == 12 * A.nnz + 4 * (A.n + 1) + 16 * A.n
    rng = np.random.default_rng(1)
    x = rng.standard_normal(A.n)
    y_ref = A.matvec(x)
    dx, dy = ctx.array(x), ctx.empty(A.n)
    dA.matvec(dx, dy)
    assert np.array_equal(dy.to_host(), y_ref)
    # fused reductions ride along
    d = K.spmv_dot(dA, dx, dy)
    assert np.array_equal(dy.to_host(), y_ref) and abs(d - oracle.dot(x, y_ref)) <= 4e-16 * np.abs(x * y_ref).sum()
    xy, yy = K.spmv_dot2(dA, dx, dy)
    assert abs(xy - d) <= 4e-16 * np.abs(x * y_ref).sum() and abs(yy - oracle.dot(y_ref, y_ref)) <= 4e-16 * yy
    w = rng.standard_normal(A.n)
    dw = K.spmv_dotw(dA, dx, dy, ctx.array(w))
    assert abs(dw - oracle.dot(w, y_ref)) <= 4e-16 * np.abs(w * y_ref).sum()
    # the option switches back to the CSR kernels on the same handle
    ctx.set_option("spmv_template", 0)
    try:
        dA.matvec(dx, dy)
        assert np.array_equal(dy.to_host(), y_ref)
    finally:
        ctx.set_option("spmv_template", 1)


def test_incompressible_operator_stays_csr(K, ctx, oracle):
    import scipy.sparse as sp
    M = (sp.random(400, 400, density=0.02, random_state=7, format="csr") + sp.eye(400, format="csr")).tocsr()
    M.sort_indices()
    dA = K.CsrMatrix.from_host(ctx, M.indptr.astype(np.int64), M.indices.astype(np.int32), M.data, (400, 400))
    assert dA.compress() == 0 and dA.spmv_bytes_stored == dA.spmv_bytes
    x = np.linspace(-1, 1, 400)
    y = dA.matvec(ctx.array(x), ctx.empty(400)).to_host()
    assert np.allclose(y, M @ x, rtol=1e-13, atol=1e-13)


def test_solvers_on_compressed_handle_identical_histories(K, ctx, oracle):
    A = oracle.poisson3d(20)
    b = ctx.array(np.ones(A.n))
    dA, dC = _upload(K, ctx, A), _upload(K, ctx, A)
    assert dC.compress() == 27
    for fused in (0, 1, 2):
        x1, s1, _ = K.cg(dA, b, history=True, fused=fused)
        x2, s2, _ = K.cg(dC, b, history=True, fused=fused)
        assert s1.niter == s2.niter and np.array_equal(x1.to_host(), x2.to_host())
        assert np.allclose(s1.residuals, s2.residuals, rtol=1e-13)
    B = oracle.kron_unsymmetric(8)
    bh = B.matvec(np.ones(B.n))
    dB, dD = _upload(K, ctx, B), _upload(K, ctx, B)
    assert dD.compress() > 0
    for solver in (K.bicgstab, K.gmres):
        x1, s1, _ = solver(dB, ctx.array(bh), history=True)
        x2, s2, _ = solver(dD, ctx.array(bh), history=True)
        assert s1.niter == s2.niter and np.allclose(x1.to_host(), x2.to_host(), rtol=0, atol=1e-12)


def test_compressed_distributed_slabs(K, oracle):
    """The renumbered [owned | ghost] local CSR of a slab partition compresses too (ghost offsets are constant
    within a plane); distributed SpMV and CG stay bit-identical to the uncompressed distributed run."""
    import threading
    n1, world = 12, 3
    A_cpu = oracle.poisson3d(n1)
    n = A_cpu.n
    x = np.cos(np.arange(n))
    y_ref = A_cpu.matvec(x)
    starts = K.row_partition(n, world)
    out, errs = [None] * world, []

    def run(rank):
        try:
            c = K.Context(0)
            c.comm_init_local(rank, world, 5151)
            r0, r1 = starts[rank], starts[rank + 1]
            A = K.CsrMatrix.stencil(c, "poisson", n1, rows=(r0, r1), distributed=True)
            b = c.empty(r1 - r0)
            K.kfill_(b, 1.0)
            x0, s0, _ = K.cg(A, b, history=True, fused=2)
            T = A.compress()
            y = A.matvec(c.array(x[r0:r1]), c.empty(r1 - r0)).to_host()
            x1, s1, _ = K.cg(A, b, history=True, fused=2)
            out[rank] = (T, y, s0.niter, s1.niter, np.array_equal(x0.to_host(), x1.to_host()),
                         np.array_equal(s0.residuals, s1.residuals))
            c.barrier()
            c.close()
        except Exception as e:      # noqa: BLE001
            import traceback
            errs.append((rank, repr(e), traceback.format_exc()))
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join(timeout=120) for t in ts]
    assert not errs, errs
    for rank in range(world):
        T, y, n0, n1_, same_x, same_h = out[rank]
        assert T > 0 and np.array_equal(y, y_ref[starts[rank]:starts[rank + 1]])
        assert n0 == n1_ and same_x and same_h
