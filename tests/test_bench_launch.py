"""bench.py's N-rank launch path: `python bench.py --gpus N` must itself start N ranks (one per GPU),
keep working under `python -m torch.distributed.run`, and REFUSE when fewer than N devices are visible -- never shrink to
one rank or let ranks share a GPU silently.  --dry-launch runs only the launch + gloo rendezvous, so the path is covered
here without a GPU.  ref: docs/src/custom_workspaces.md:583-637 (the reference's `mpiexecjl -n 4` recipe)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    env.update(extra)
    return env


def _json_line(stdout: bytes):
    lines = [l for l in stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines          # stdout carries exactly one line
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [2, 3])
def test_bench_gpus_n_launches_n_ranks_itself(n):
    p = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--dry-launch"], env=_clean_env(), capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()
    d = _json_line(p.stdout)
    assert d["dry_launch"] and d["n_gpus"] == n and d["ranks_rendezvoused"] == n and d["launcher"] == "bench.py"
    assert sorted(r[0] for r in d["ranks"]) == list(range(n))              # ranks 0..n-1, each once
    assert [r[1] for r in sorted(d["ranks"])] == list(range(n))            # LOCAL_RANK = rank: one device each
    assert len({r[2] for r in d["ranks"]}) == n                            # n distinct processes
    # the per-rank phase report of a real N > 1 run, gathered and summarised by the same code on synthetic
    # brackets: one row per rank, every phase with launches / ms per iteration / average, and the falsifier checks of DESIGN.md 5
    ph = d["phases"]
    assert [row["rank"] for row in ph["per_rank"]] == list(range(n))
    for row in ph["per_rank"]:
        for k in ("halo_pack", "halo_transfer", "spmv", "spmv_boundary", "dot_allgather_combine"):
            assert set(row[k]) == {"launches_per_iteration", "ms_per_iteration", "avg_us"}, row
        assert row["dot_allgather_combine"]["launches_per_iteration"] == 2.0
    c = ph["checks"]
    assert c["dot_allgather_combine_avg_us_max"] == pytest.approx(20.0 * n)          # the slowest rank's average
    assert c["falsifier_allgather_above_50us"] == (20.0 * n > 50.0)
    assert c["falsifier_halo_longer_than_interior_product"] is False
    assert set(d["ab"]) == {"overlap_halo_0", "comm_priority_0"}


def test_phase_summary_flags_the_falsifiers():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    steps = 20
    profs = [{"spmv": (steps, 0.26 * steps), "spmv_boundary": (steps, 0.01 * steps), "halo_pack": (steps, 0.004 * steps),
              "halo_transfer": (steps, 0.30 * steps), "dot_allgather_combine": (2 * steps, 0.06 * 2 * steps)} for _ in range(8)]
    out = bench.summarize_phases(profs, steps)
    assert len(out["per_rank"]) == 8
    assert out["checks"]["falsifier_allgather_above_50us"] and out["checks"]["falsifier_halo_longer_than_interior_product"]
    assert out["per_rank"][3]["halo_transfer"]["avg_us"] == pytest.approx(300.0)
    # a single-GPU profile has no communication phases
    one = bench.summarize_phases([{"spmv": (steps, 2.2 * steps)}], steps)
    assert one["per_rank"][0]["halo_transfer"]["avg_us"] is None and one["checks"]["falsifier_allgather_above_50us"] is False


def test_bench_under_torch_distributed_run_still_works():
    """The driver's launch: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ..."""
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "2", "--dry-launch"], env=_clean_env(), capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()
    d = _json_line(p.stdout)
    assert d["n_gpus"] == 2 and d["ranks_rendezvoused"] == 2 and d["launcher"] == "external"


def test_single_rank_default_does_not_spawn():
    p = subprocess.run([sys.executable, BENCH, "--dry-launch"], env=_clean_env(), capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()
    d = _json_line(p.stdout)
    assert d["n_gpus"] == 1 and d["ranks_rendezvoused"] == 1 and d["launcher"] == "external"


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="CPU-box form of the refusal (the GPU form is test_refuses_more_ranks_than_devices)")
def test_refuses_without_devices():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1"], env=_clean_env(), capture_output=True, timeout=300)
    assert p.returncode == 2 and p.stdout.strip() == b""
    assert b"refusing" in p.stderr


@pytest.mark.gpu
def test_refuses_more_ranks_than_devices():
    sys.path.insert(0, ROOT)
    import krylov_jl_amd as K
    ndev = K.device_count()
    p = subprocess.run([sys.executable, BENCH, "--gpus", str(ndev + 1), "--steps", "2", "--warmup", "1"], env=_clean_env(),
                       capture_output=True, timeout=300)
    assert p.returncode == 2 and p.stdout.strip() == b"", (p.returncode, p.stdout)
    assert f"--gpus {ndev + 1} asked for, {ndev} HIP device(s) visible".encode() in p.stderr
    # ... and a rank started by an external launcher without a device of its own refuses as well (no silent sharing)
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = _clean_env(RANK="0", LOCAL_RANK=str(ndev), WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "KHIP_ALLOW_SHARED_GPU"):
        env.pop(v, None)
    if ndev > 1 or True:
        p = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "2", "--warmup", "1", "--n1", "32", "--no-cpu-baseline"],
                           env=env, capture_output=True, timeout=300)
        assert p.returncode == 2 and b"refusing to share a GPU" in p.stderr, (p.returncode, p.stderr[-400:])


def test_a_dying_rank_ends_the_launch_with_its_return_code():
    """One rank dies before the rendezvous: the launcher must not hang on the others (they wait in the gloo rendezvous) -- it
    stops exactly the processes it started and returns the worst return code; nothing is printed on stdout."""
    import time
    t0 = time.time()
    p = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--dry-launch"], env=_clean_env(KHIP_BENCH_TEST_FAIL_RANK="1"),
                       capture_output=True, timeout=300)
    assert p.returncode == 9 or p.returncode == 7, (p.returncode, p.stderr.decode()[-500:])      # 7 from the rank, 9 = a survivor that had to be killed
    assert p.stdout.strip() == b"" and b"rank return codes" in p.stderr
    assert time.time() - t0 < 120
