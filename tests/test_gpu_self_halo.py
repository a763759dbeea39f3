"""The measurement hook behind tools/slab_iteration.py: one RCCL rank owning a slab of planes exchanges its two
boundary planes with ITSELF through the real neighbour-exchange code (pack kernel, grouped ncclSend / ncclRecv on the split-off
halo communicator and stream, interior / boundary split, 16-byte all-gather + combine of the dots).  What it computes is the
PERIODIC slab; this test pins that against the same operator as a plain single-GPU CSR."""
import json
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n1,k0,k1", [(24, 8, 16), (32, 4, 28)])
def test_self_halo_slab_equals_the_periodic_operator(n1, k0, k1):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "r.json")
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        try:
            p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "self_halo_worker.py"), str(n1), str(k0), str(k1), out],
                               env=env, capture_output=True, text=True, timeout=240)
        except subprocess.TimeoutExpired:
            pytest.fail("the self Send/Recv hangs")
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
        r = json.load(open(out))
    assert r["rccl_ranks"] == 1 and r["gather_mode"] == 0
    assert r["n_ghost"] == 2 * n1 * n1 == r["n_send"]                  # the two neighbouring planes, SURVEY 8e
    assert r["spmv_uses_halo"]
    assert r["spmv_bit_identical"] and r["spmv_overlap1_bit_identical"] and r["spmv_overlap0_bit_identical"]
    assert r["spmv_priority0_bit_identical"]                            # communication stream at default priority (ctx option comm_priority)
    # the phases of three products, each bracketed once per product on the stream it runs on (khip_profile_kernels)
    pl = r["phase_launches"]
    assert (pl["halo_pack"], pl["halo_transfer"], pl["spmv"], pl["dot_allgather_combine"]) == (3, 3, 3, 0)
    assert pl["spmv_boundary"] in (3, 6)            # the two boundary ranges are ONE launch where the staged / coded kernel takes them, two otherwise
    assert r["phase_ms_positive"]
    for fused in (0, 2):
        assert r[f"cg_fused{fused}_niter"][0] == r[f"cg_fused{fused}_niter"][1]
        # same kernels on the same values; the dots go through one more (1-rank) combine step: <= 1 ulp per dot
        assert r[f"cg_fused{fused}_max_rel_dev"] <= 1e-12
