"""The distributed code path on ONE GPU: a 1-rank RCCL communicator exercises comm_init, the halo-plan
construction on the device (collect / sort / remap kernels), the [owned | ghost] SpMV variant, the
all-gathered (hi, lo) dot and the interior / boundary split -- everything except the actual peer
traffic, which needs more than one GPU (covered on CPU by tests/test_dist_gloo.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dctx(K):
    c = K.Context(0)
    c.comm_init(0, 1, K.Context.comm_unique_id())
    yield c
    c.close()


def test_single_rank_communicator_matches_plain_path(K, ctx, dctx, oracle):
    n1 = 24
    A_cpu = oracle.poisson3d(n1)
    n = A_cpu.n
    Ad = K.CsrMatrix.stencil(dctx, "poisson", n1, rows=(0, n), distributed=True)
    Ap = K.CsrMatrix.stencil(ctx, "poisson", n1)
    x = np.linspace(-1, 1, n) ** 3
    yd = Ad.matvec(dctx.array(x)).to_host()
    assert np.array_equal(yd, Ap.matvec(ctx.array(x)).to_host())
    assert np.array_equal(yd, A_cpu.matvec(x))
    # dot through the all-gather path == plain path
    a, b = dctx.array(x), dctx.array(x[::-1].copy())
    assert K.kdot(n, a, b) == K.kdot(n, ctx.array(x), ctx.array(x[::-1].copy()))
    # CG: identical history
    bd = dctx.empty(n); K.kfill_(bd, 1.0)
    bp = ctx.empty(n); K.kfill_(bp, 1.0)
    _, st_d, _ = K.cg(Ad, bd, history=True)
    _, st_p, _ = K.cg(Ap, bp, history=True)
    ref = oracle.cg(A_cpu, np.ones(n), history=True)
    assert st_d.niter == st_p.niter == ref.niter
    assert np.array_equal(st_d.residuals, st_p.residuals)
    dctx.barrier()


def test_row_slab_with_ghost_columns_single_process(K, dctx, oracle):
    """A middle slab [r0, r1) of the grid as a 'distributed' operator needs ghost planes on both sides; with one
    rank nobody can own them, so creation must fail loudly (partition does not cover the operator)."""
    n1 = 12
    n = n1 ** 3
    with pytest.raises(K.KhipError):
        K.CsrMatrix.stencil(dctx, "poisson", n1, rows=(n // 4, n // 2), distributed=True)
