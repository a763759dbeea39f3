"""tools/bench_mtx.py -- the reference's benchmark loop over MatrixMarket files (benchmark/cg_bmark.jl:29-54,
benchmark/gpu.jl:15-47) through the HIP path.  CPU: the loader on the two committed fixtures
(tests/golden/tiny_*.mtx, made by tests/golden/make_mtx_fixtures.py) and the --dry --oracle mode; GPU: the tool end to end,
iteration counts against the oracle's."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "bench_mtx.py")
GOLD = os.path.join(ROOT, "tests", "golden")


def _rows(args):
    p = subprocess.run([sys.executable, TOOL] + args, capture_output=True, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return [json.loads(l) for l in p.stdout.decode().splitlines() if l.startswith("{")]


def test_loader_mirrors_symmetric_files_and_sorts_general_ones():
    import scipy.sparse as sp
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_mtx
    S, sym = bench_mtx.load_mtx(os.path.join(GOLD, "tiny_spd_sym.mtx"))
    n1, n2 = 6, 7
    T = lambda n: sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(n, n))
    want = (sp.kron(sp.identity(n2), T(n1)) + sp.kron(T(n2), sp.identity(n1)) + 0.5 * sp.identity(n1 * n2)).tocsr()
    want.sort_indices()
    assert sym and S.shape == (42, 42) and S.nnz == want.nnz == 184            # 113 stored entries, mirrored
    assert np.array_equal(S.indptr, want.indptr) and np.array_equal(S.indices, want.indices) and np.array_equal(S.data, want.data)
    U, symu = bench_mtx.load_mtx(os.path.join(GOLD, "tiny_unsym.mtx"))
    assert not symu and U.shape == (30, 30) and U.nnz == 162 and U.has_sorted_indices
    assert all(np.all(np.diff(U.indices[U.indptr[i]:U.indptr[i + 1]]) > 0) for i in range(30))
    d = bench_mtx.describe("x.mtx", S, sym)
    assert d["bandwidth"] == 6 and d["distinct_diagonals"] == 5 and d["max_row"] == 5


def test_dry_mode_with_the_oracle_needs_no_gpu():
    rows = _rows(["--dry", "--oracle", "--rtol", "1e-8", GOLD])
    assert [r["matrix"] for r in rows] == ["tiny_spd_sym.mtx", "tiny_unsym.mtx"]
    assert rows[0]["solver"] == "cg" and rows[1]["solver"] == "bicgstab"
    assert all(r["oracle"]["solved"] for r in rows) and "niter" not in rows[0]


@pytest.mark.gpu
def test_mtx_files_through_the_hip_path_match_the_oracle():
    rows = _rows(["--oracle", "--rtol", "1e-8", GOLD])
    assert len(rows) == 2
    for r in rows:
        assert r["solved"] and r["niter"] == r["oracle"]["niter"] and r["status"] == r["oracle"]["status"], r
        assert r["true_rel_residual"] <= 1e-7 and r["spmv_ms"] > 0 and r["alg_bytes"] == 12 * r["nnz"] + 4 * (r["rows"] + 1) + 16 * r["rows"]
