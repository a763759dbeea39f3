"""Host-only checks of the analysis behind the ILU(0) block schedule (csrc/ilu.hip: detect_grid, make_grid_partition,
make_level_partition, analyse_blocks) through the test export khip_test_ilu_blocks_host: no device needed.  The export
itself verifies that every row lands in exactly one block and that no block depends on a later one; here the patterns
and what must be recognised."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp


def _lib():
    import krylov_jl_amd as K
    return K.lib()


def _analyse(S, mode=1):
    S = S.tocsr(); S.sort_indices()
    rp = np.ascontiguousarray(S.indptr, dtype=np.int64)
    ci = np.ascontiguousarray(S.indices, dtype=np.int32)
    out = (C.c_int64 * 10)()
    rc = _lib().khip_test_ilu_blocks_host(S.shape[0], rp.ctypes.data_as(C.POINTER(C.c_int64)), ci.ctypes.data_as(C.POINTER(C.c_int32)), mode, out)
    assert rc == 0, (rc, _lib().khip_last_error().decode())
    return dict(dims=tuple(out[0:3]), skew=out[3], nb=(out[4], out[5]), max_ext=out[6], rec_ok=out[7], max_row=out[8], rows_cap=out[9])


def _tri(n, k=1):
    return sp.diags([np.ones(n - abs(o)) for o in range(-k, k + 1)], list(range(-k, k + 1)), format="csr")


def _kron3(a, b, c):
    return sp.kron(sp.kron(a, b), c, format="csr")


def _star(n1, n2, n3, k=1):          # 5- / 7-point (k = 1) or second-neighbour star (k = 2)
    I = lambda n: sp.identity(n, format="csr")
    S = _kron3(I(n3), I(n2), _tri(n1, k)) + _kron3(I(n3), _tri(n2, k), I(n1))
    return S + _kron3(_tri(n3, k), I(n2), I(n1)) if n3 > 1 else S


@pytest.mark.parametrize("dims", [(16, 16, 16), (19, 23, 17), (70, 64, 1), (31, 8, 20)])
def test_star_stencils_are_grids_with_the_natural_basis(dims):
    r = _analyse(_star(*dims))
    assert r["dims"] == dims and r["skew"] == 0 and r["rec_ok"] == 1 and r["max_row"] <= 3
    cube = 512 if dims[2] > 1 else 256
    full = np.prod([-(-d // (8 if dims[2] > 1 else 16)) for d in dims[:2]]) * (-(-dims[2] // 8) if dims[2] > 1 else 1)
    assert r["nb"] == (full, full) and r["rows_cap"] == cube


def test_second_neighbour_star_keeps_the_natural_basis_and_wide_rows():
    r = _analyse(_star(18, 17, 16, k=2))
    assert r["dims"] == (18, 17, 16) and r["skew"] == 0 and r["rec_ok"] == 0 and r["max_row"] == 6


@pytest.mark.parametrize("dims", [(16, 18, 17), (72, 64, 1)])
def test_box_stencils_need_the_skewed_basis(dims):
    n1, n2, n3 = dims
    S = _kron3(_tri(n3), _tri(n2), _tri(n1)) if n3 > 1 else sp.kron(_tri(n2), _tri(n1), format="csr")      # 27- / 9-point
    r = _analyse(S)
    assert r["dims"] == dims and r["skew"] == 1 and r["rec_ok"] == 0
    assert r["max_row"] == (13 if n3 > 1 else 4)
    assert r["nb"][0] == r["nb"][1] > 0


def test_periodic_coupling_is_not_a_grid():
    n1 = 16
    S = _star(n1, n1, n1).tolil()
    S[0, n1 - 1] = 1.0; S[n1 - 1, 0] = 1.0          # wraps around the x face
    r = _analyse(S.tocsr())
    assert r["dims"] == (0, 0, 0) and r["nb"][0] > 0          # level-sequence blocks instead


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_level_sequence_blocks_on_patterns_without_a_grid(seed):
    rng = np.random.default_rng(seed)
    S = _star(16, 16, 16)
    perm = rng.permutation(S.shape[0])
    Sp = S[perm][:, perm]
    r = _analyse(Sp)
    assert r["dims"] == (0, 0, 0) and r["nb"][0] > 0 and r["nb"][1] > 0 and r["rows_cap"] % 64 == 0
    R = (sp.random(6000, 6000, density=0.002, random_state=seed, format="csr") + sp.identity(6000, format="csr")).tocsr()
    R = (R + R.T).tocsr()
    r = _analyse(R)
    assert r["dims"] == (0, 0, 0) and r["nb"][0] > 0
    # the same grid through the level sequence: more, smaller blocks than its cubes
    g = _analyse(S, mode=1); lv = _analyse(S, mode=3)
    assert lv["dims"] == (0, 0, 0) and lv["nb"][0] > g["nb"][0]
