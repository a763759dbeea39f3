"""The distributed path over REAL RCCL: one process per GPU, ncclCommInitRank through libkrylov_hip, both halo modes
(neighbour Send/Recv on the split-off halo communicator, all-gather of x), the all-gathered (hi, lo) dots, the
device-resident loops running several iterations ahead of the host -- against the CPU oracle's solve of the global
system.  Needs >= 2 GPUs in one box: skipped (cleanly, at collection of the parametrisation) on the 1-GPU boxes the
round's own runs get; the in-process backend (tests/test_gpu_dist.py) covers everything above the three transport calls
there.  World sizes 2, 4 and 8, each where that many GPUs are visible.  ref: docs/src/custom_workspaces.md:477-586 (the MPI recipe)."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EPS = np.finfo(float).eps


def _ngpu():
    sys.path.insert(0, ROOT)
    import krylov_jl_amd as K
    return K.device_count() if K.gpu_available() else 0


@pytest.mark.parametrize("world", [2, 4, 8])
def test_rccl_ranks_match_oracle(oracle, world):
    ngpu = _ngpu()
    if ngpu < world:
        pytest.skip(f"{world} ranks over RCCL need {world} GPUs in one box ({ngpu} visible)")
    with tempfile.TemporaryDirectory() as d:
        uid = os.path.join(d, "uid.bin")
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py"), str(r), str(world), uid, d],
                                  env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
        logs = []
        for p in procs:
            try:
                o, _ = p.communicate(timeout=600)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                pytest.fail("a rank is stuck (collective mismatch?)")
            logs.append(o.decode(errors="replace"))
        assert all(p.returncode == 0 for p in procs), "\n".join(logs)
        res = [dict(np.load(os.path.join(d, f"rank{r}.npz"))) for r in range(world)]

    # ---- references: the oracle on the global systems
    n1 = 24
    A = oracle.poisson3d(n1)
    n = A.n
    x = np.linspace(-1, 1, n) ** 3 + 0.25
    y_ref = A.matvec(x)
    cg_ref = oracle.cg(A, np.ones(n), history=True)
    import krylov_jl_amd as K
    starts = K.row_partition(n, world)
    n1b, p = 14, 4
    Ab = oracle.kron_unsymmetric(n1b)
    nb = Ab.n
    bh = Ab.matvec(np.ones(nb))
    g_ref = oracle.gmres(Ab, bh, memory=10, restart=True, history=True)
    b_ref = oracle.bicgstab(Ab, bh, history=True)
    tt = (np.arange(nb) + 1.0) / nb
    Xt = np.stack([tt ** j for j in range(p)], axis=1)
    B = np.stack([Ab.matvec(np.ascontiguousarray(Xt[:, j])) for j in range(p)], axis=1)
    k_ref = oracle.block_gmres(Ab, B, memory=8, history=True)
    startsb = K.row_partition(nb, world)
    xt = np.cos(np.arange(nb) * 0.01)
    Sb = Ab.to_scipy().T.tocsr()
    Sb.sort_indices()
    yt_ref = oracle.CsrMatrix.from_arrays(Sb.indptr.astype(np.int64), Sb.indices.astype(np.int32), Sb.data.copy()).matvec(xt)

    for rank, out in enumerate(res):
        assert int(out["rccl_ranks"]) == world
        r0, r1 = starts[rank], starts[rank + 1]
        q0, q1 = startsb[rank], startsb[rank + 1]
        for mode in (1, 2):
            assert int(out[f"gather{mode}"]) == mode - 1
            for overlap in (1, 0):
                assert np.array_equal(out[f"y{mode}{overlap}"], y_ref[r0:r1]), (rank, mode, overlap)      # bit-identical
            for fused in (2, 1, 0):
                h = out[f"cg{mode}{fused}_hist"]
                assert len(h) == len(cg_ref.residuals)
                assert np.max(np.abs(h - cg_ref.residuals) / cg_ref.residuals) <= 1e-10
                assert np.allclose(out[f"cg{mode}{fused}_x"], cg_ref.x[r0:r1], atol=1e-10)
                assert np.array_equal(h, res[0][f"cg{mode}{fused}_hist"])                                # same scalars on every rank
            assert np.array_equal(out[f"cg{mode}2_hist"], out[f"cg{mode}1_hist"])
            assert abs(len(out[f"cgv{mode}_hist"]) - len(cg_ref.residuals)) <= 2
            assert np.array_equal(out[f"timed{mode}"], res[0][f"timed{mode}"]) and int(out[f"timed{mode}"][1]) == 1
            assert np.array_equal(out[f"b{mode}"], bh[q0:q1]) and np.array_equal(out[f"B{mode}"], B[q0:q1])
            for key, ref, tol in ((f"gmres{mode}", g_ref, 1e-8), (f"bicgstab{mode}", b_ref, 1e-7), (f"block{mode}", k_ref, 1e-7)):
                h = out[key]
                assert len(h) == len(ref.residuals), key
                assert np.max(np.abs(h - ref.residuals) / (tol * ref.residuals + 100 * EPS * ref.residuals[0])) <= 1.0, key
                assert np.array_equal(h, res[0][key])
            assert np.allclose(out[f"blockX{mode}"], k_ref.x[q0:q1], atol=1e-8 * np.abs(k_ref.x).max())
            assert np.array_equal(out[f"At{mode}"], yt_ref[q0:q1]), (rank, mode, "A' x over RCCL")
        assert np.array_equal(out["cg12_hist"], out["cg22_hist"])                                           # the two halo modes agree bit for bit



def test_bench_with_one_rank_communicator_equals_the_plain_path():
    """bench.py's distributed code path with ONE rank over real RCCL (KHIP_FORCE_COMM=1: ncclCommInitRank, the row-partitioned
    handle, the all-gathered (hi, lo) dots inside the device-resident loop) gives the same residuals as the plain single-GPU
    path and reports the rank RCCL saw.  What an 8-GPU run adds on top is only peer traffic (rows above, when there are GPUs)."""
    import json
    outs = {}
    for force in ("0", "1"):
        env = dict(os.environ, KHIP_FORCE_COMM=force, HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.pop("RANK", None); env.pop("WORLD_SIZE", None)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n1", "96", "--steps", "30", "--warmup", "2", "--no-cpu-baseline",
                            "--also-variant1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r.returncode == 0, r.stderr.decode(errors="replace")[-2000:]
        lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
        assert len(lines) == 1, lines                                   # exactly one JSON line on stdout
        outs[force] = json.loads(lines[0])
    plain, comm = outs["0"], outs["1"]
    assert plain["rccl_ranks_seen"] == 0 and comm["rccl_ranks_seen"] == 1
    assert plain["phases"] is None and comm["phases"]["per_rank"][0]["dot_allgather_combine"]["launches_per_iteration"] == 2.0
    chosen = comm["halo_probe"]["chosen_overlap_halo"]                    # the warm-up probe of the halo overlap ran (one rank: no halo, a tie)
    assert chosen in (0, 1) and set(comm["halo_probe"]["ms_per_iteration"]) == {"overlap_halo_1", "overlap_halo_0"}
    assert set(comm["ab"]) == {f"overlap_halo_{1 - chosen}", "comm_priority_0"} and plain["halo_probe"] is None
    assert comm["final_residual_norm"] == plain["final_residual_norm"]                  # same bits through the communicator
    assert comm["steps"] == plain["steps"] == 30 and comm["n_gpus"] == 1
    for o in (plain, comm):
        sr = o["single_reduction_cg"]
        assert sr is not None and "NOT_THE_HEADLINE" in sr and sr["steps"] == 30 and sr["value"] > 0
        assert o["roofline"]["frac"] > 0 and o["metric"] == "cg_iters_per_sec_poisson3d_csr_512cubed"


@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_gpus_n_runs_n_rccl_ranks(world):
    """`python bench.py --gpus N` as the driver types it at N = 1 (no launcher around it): N ranks over real RCCL, one per
    GPU, n_gpus = N in the line, RCCL saw N ranks, and the partitioned 512^3 history equals the 1-GPU golden bit for bit
    (self_consistency) and the CPU oracle's within 1e-12 (parity).  Skips below N GPUs (tests/test_bench_launch.py covers
    the launch itself on CPU and the refusal on a 1-GPU box)."""
    import json
    ngpu = _ngpu()
    if ngpu < world:
        pytest.skip(f"bench.py --gpus {world} needs {world} GPUs in one box ({ngpu} visible)")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "20", "--warmup", "5",
                        "--no-full-parity"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    o = json.loads(lines[0])
    assert o["n_gpus"] == world and o["rccl_ranks_seen"] == world and o["steps"] == 20 and o["scaling"] == "strong"
    assert o["self_consistency"]["max_rel_dev"] == 0.0 and o["self_consistency"]["iterations_compared"] == 100
    assert o["parity"]["ok"] and o["parity"]["max_rel_dev"] <= 1e-12
    assert o["value"] > 0 and o["config"]["partition"] == f"1-D rows over {world} GPU(s)"
    # the run explains itself: per-rank phase times from HIP events, the link / communicator facts, the A/B legs
    assert [row["rank"] for row in o["phases"]["per_rank"]] == list(range(world))
    for row in o["phases"]["per_rank"]:
        assert row["halo_transfer"]["launches_per_iteration"] == 1.0 and row["halo_pack"]["launches_per_iteration"] == 1.0
        assert row["spmv"]["launches_per_iteration"] == 1.0 and row["spmv_boundary"]["launches_per_iteration"] == 1.0
        assert row["dot_allgather_combine"]["launches_per_iteration"] == 2.0 and row["dot_allgather_combine"]["avg_us"] > 0
    assert o["comm"]["rccl_ranks"] == world and o["comm"]["halo"]["gather_mode"] == 0 and o["comm"]["halo"]["n_ghost"] > 0
    chosen = o["halo_probe"]["chosen_overlap_halo"]
    assert set(o["ab"]) == {f"overlap_halo_{1 - chosen}", "comm_priority_0"} and all(v["value"] > 0 for v in o["ab"].values())
    assert all(v > 0 for v in o["halo_probe"]["ms_per_iteration"].values())
