#!/usr/bin/env python3
"""bench.py -- CG iterations/second (+ achieved HBM GB/s) on the 3-D Poisson 7-point CSR operator.

Workload (BASELINE.json configs[1]): cg! on get_div_grad(512,512,512) (n = 134,217,728, nnz = 937,951,232),
Float64, b = ones, x0 = 0.  A "step" is one CG iteration (1 SpMV + 2 dots + 2 axpy + 1 axpby, fused into
3 kernels: SpMV+p.Ap | r update + r.r | x and p update).  With --gpus N the SAME 512^3 problem is row-partitioned over N ranks (strong scaling, one
process per GPU, RCCL over xGMI inside libkrylov_hip: all-gather of the (hi, lo) dot partials, neighbour
halo exchange before each SpMV).  Inputs are generated on the device, so the timed region starts with
everything resident in HBM.

Prints ONE JSON line on rank 0 (stdout); diagnostics go to stderr.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# RCCL between processes needs dmabuf IPC on this driver (hipIpcGetMemHandle fails in legacy mode): set before any HIP runtime loads,
# whatever launcher started this rank (the self-launcher below sets it too; torch.distributed.run inherits the caller's environment)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

PARITY_ITERS = 100          # iterations of the oracle history (tests/golden/oracle_cfg2_cg512.json)
PMC_PROFILE = "r06end_spmv_pmc.json"   # rocprofv3 counter passes of this round's kernels (tools/gpu_prof.sh)
HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md); 6.29 TB/s is the measured copy ceiling


def log(*a):
    print(*a, file=sys.stderr, flush=True)


KERNEL_SOURCES = ("spmv.hip", "spmv_common.hpp", "colcode.hip", "device_reduce.hpp")


def kernel_source_sha():
    """Fingerprint of the SpMV kernel sources: a PMC profile is only quoted when it was taken from this very build."""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "krylov.jl_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(n1, kernel_substr):
    """HBM-side bytes per launch of the fused SpMV kernel from the committed rocprofv3 PMC passes of THIS build
    (tools/gpu_prof.sh: separate --pmc FETCH_SIZE / WRITE_SIZE runs of tools/spmv_only.py at 512^3; the json carries
    the sha of the kernel sources it was taken from).  Correction per MI355X_MICROARCH.md: counters are in KiB and
    FETCH_SIZE tallies 128-B line fetches as 64 B, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
    Returns (bytes or None, note)."""
    path = os.path.join(ROOT, "profiles", PMC_PROFILE)
    if n1 != 512 or not os.path.exists(path):
        return None, "no PMC profile for this size"
    try:
        d = json.load(open(path))
        if d.get("_kernel_source_sha") != kernel_source_sha():
            return None, "profiles/" + PMC_PROFILE + f" was taken from other kernel sources ({d.get('_kernel_source_sha')}): not quoted"
        for k, v in d.items():
            if isinstance(v, dict) and kernel_substr in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                return (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0, (
                    "L2-miss-side bytes per launch from separate rocprofv3 --pmc passes of this build "
                    "(profiles/" + PMC_PROFILE + ", (2 FETCH_SIZE + WRITE_SIZE) KiB); includes Infinity-Cache hits")
    except Exception as e:
        return None, f"unreadable PMC profile: {e}"
    return None, "kernel not in the PMC profile"


def _hist_dev(residuals, golden):
    k = min(len(golden), len(residuals))
    if k == 0:
        return None, 0
    return max(abs(float(residuals[i]) - golden[i]) / golden[i] for i in range(k)), k - 1


def parity_vs_oracle(n1, residuals, full=None):
    """This build's residual history against the CPU ORACLE's (oracle/krylov_oracle.c ko_cg = src/cg.jl:120-291 at 512^3,
    goldens made by tests/golden/make_scale_golden.py).  Two legs: the FULL SOLVE of the benchmark definition
    cg(A, b, atol = 0, rtol = 1e-8, itmax = n) (benchmark/benchmarks.jl:14-21; tests/golden/oracle_cfg2_cg512_full.json, 1225
    iterations) -- equal iteration count and status, whole history within 1e-8 (DESIGN.md 3.2c derives that bound from
    binary128 runs) -- and the first 100 iterations with atol = rtol = 0 within the north star's 1e-12
    (tests/golden/oracle_cfg2_cg512.json)."""
    path = os.path.join(ROOT, "tests", "golden", "oracle_cfg2_cg512.json")
    if n1 != 512 or not os.path.exists(path):
        return None
    dev, k = _hist_dev(residuals, json.load(open(path))["residuals"])
    if dev is None:
        return None
    out = {"against": "CPU oracle history, oracle/krylov_oracle.c ko_cg at 512^3 (tests/golden/oracle_cfg2_cg512_full.json: full solve; "
                      "oracle_cfg2_cg512.json: 100-iteration prefix)",
           "iterations_compared": k, "max_rel_dev": dev, "tolerance": 1e-12, "ok": bool(dev <= 1e-12)}
    fpath = os.path.join(ROOT, "tests", "golden", "oracle_cfg2_cg512_full.json")
    if full is not None and os.path.exists(fpath):
        g = json.load(open(fpath))
        fdev, fk = _hist_dev(full["residuals"], g["residuals"])
        same = bool(full["niter"] == g["niter"] and full["status"] == g["status"])
        out = {"against": out["against"], "iterations_compared": fk, "niter": int(full["niter"]), "oracle_niter": int(g["niter"]),
               "equal_iteration_count_and_status": same, "status": full["status"], "max_rel_dev": fdev, "tolerance": 1e-8,
               "ok": bool(same and fdev is not None and fdev <= 1e-8 and out["ok"]),
               "prefix_100_iterations": {"iterations_compared": k, "max_rel_dev": dev, "tolerance": 1e-12, "ok": bool(dev <= 1e-12)},
               "setting": "full solve: atol = 0, rtol = 1e-8, itmax = n (benchmark/benchmarks.jl:14-21); prefix: atol = rtol = 0"}
    # the same recurrence with EXACT dots on the oracle's side (ko_set_dot_mode(1): Dot2; tests/golden/oracle_cfg2_cg512_exact_dots.json,
    # make_scale_golden.py leg 22): what is left when the documented oracle's own sequential extended-precision summation error is
    # taken out of the comparison -- DESIGN.md 3.2d
    epath = os.path.join(ROOT, "tests", "golden", "oracle_cfg2_cg512_exact_dots.json")
    if os.path.exists(epath):
        e = json.load(open(epath))
        pdev, pk = _hist_dev(residuals, e["prefix_residuals"])
        ex = {"against": "oracle/krylov_oracle.c ko_cg with Dot2 dots (tests/golden/oracle_cfg2_cg512_exact_dots.json)",
              "prefix_iterations_compared": pk, "prefix_max_rel_dev": pdev}
        if full is not None:
            fdev, fk = _hist_dev(full["residuals"], e["residuals"])
            ex.update(full_iterations_compared=fk, full_max_rel_dev=fdev,
                      full_history_bit_identical=bool(len(full["residuals"]) == len(e["residuals"]) and
                                                      all(float(a) == b for a, b in zip(full["residuals"], e["residuals"]))))
        out["vs_exact_dot_oracle"] = ex
    return out


def _prefix_dev(hist, golden_file, x=None):
    """max relative deviation of a residual history from a golden oracle history (tests/golden/<file>), and of the solution sample."""
    path = os.path.join(ROOT, "tests", "golden", golden_file)
    if not os.path.exists(path):
        return None
    g = json.load(open(path))
    href = g["residuals"]
    k = min(len(hist), len(href))
    dev = max(abs(float(hist[i]) - href[i]) / href[i] for i in range(k) if href[i] != 0.0)
    out = {"against": f"tests/golden/{golden_file} ({g['oracle']})", "iterations_compared": k - 1, "max_rel_dev": dev,
           "tolerance": 1e-12, "ok": bool(dev <= 1e-12 and len(hist) == len(href))}
    if x is not None:
        xg = g["x_sample"]
        import numpy as np
        xs = np.asarray(x)[g["x_index"]]
        out["x_sample_max_rel_dev"] = float(np.max(np.abs(xs - np.asarray(xg))) / np.max(np.abs(np.asarray(xg))))
    return out


def other_configs(K, ctx, log):
    """BASELINE cfg 3 and cfg 5 on the driver-run line: nested, labelled, NOT part of `value`.  Each leg: an
    untimed parity solve against the oracle's golden history, then a timed run with HIP-event brackets around every kernel of the
    families that carry the bytes (khip_profile_kernels), fractions of the 8 TB/s peak on ALGORITHMIC bytes."""
    import numpy as np
    out = {}
    # ---------------- cfg 3: gmres!(memory = 30, restart = true) on kron_unsymmetric(256), b = A * ones ----------------
    t_leg = time.time()
    n1 = 256
    n = n1 ** 3
    A = K.CsrMatrix.stencil(ctx, "kron_unsymmetric", n1)
    ones = ctx.empty(n); K.kfill_(ones, 1.0)
    b = ctx.empty(n); A.matvec(ones, b)
    ws = K.GmresWorkspace(ctx, n, n, memory=30)
    K.gmres_(ws, A, b, restart=True, atol=0.0, rtol=0.0, itmax=45, history=True)                   # the golden's 45 iterations
    parity = _prefix_dev(ws.stats.residuals, "oracle_cfg3_gmres256.json", ws.x.to_host())
    K.gmres_(ws, A, b, restart=True, itmax=30, atol=0.0, rtol=0.0)                                 # warm-up cycle
    ctx.set_option("profile_spmv", 1); ctx.profile_kernels()
    ctx.sync(); t0 = time.perf_counter()
    K.gmres_(ws, A, b, restart=True, itmax=90, atol=0.0, rtol=0.0)                                 # three whole cycles
    ctx.sync(); dt = time.perf_counter() - t0
    prof = ctx.profile_kernels(); ctx.set_option("profile_spmv", 0)
    it = ws.stats.niter
    sb = A.spmv_bytes
    # one 30-step cycle as the reference issues it (src/gmres.jl:236-330): per inner iteration k the product, k x (kdot 16n +
    # kaxpy 24n), knorm 8n, kdivcopy 16n; per cycle the residual (product + kaxpby 24n), its norm and scaling (8n + 16n), and the
    # solution update (30 kaxpy of 24n)
    cyc_ref = sum(sb + k * 40 * n + 24 * n for k in range(1, 31)) + 30 * 24 * n + sb + 24 * n + 24 * n
    # ... and with the legal fusions this library applies: product; per basis vector ONE pass (read V_i and q, write q: 24n, the
    # coefficient of the next vector folded in) -- the last one also yields ||q||^2; scaling 16n; per cycle the residual product
    # with its update fused (sb + 16n), and the solution update as one multi-axpy ((30 + 2) 8n)
    cyc_fused = sum(sb + k * 24 * n + 16 * n for k in range(1, 31)) + sb + 16 * n + 32 * 8 * n
    cycles = it / 30.0
    spmv_l, spmv_ms = prof["spmv"]
    out["cfg3_gmres"] = {
        "NOT_THE_HEADLINE": "BASELINE cfg 3, untimed for `value`: gmres!(memory = 30, restart = true) on kron_unsymmetric(256) CSR (117,047,296 entries), "
                            "b = A * ones, atol = rtol = 0, khip_gmres_solve on an adopted workspace, fused = 2",
        "inner_iterations": int(it), "seconds": dt, "ms_per_inner_iteration": 1e3 * dt / it, "inner_iterations_per_sec": it / dt,
        "bytes_per_cycle_reference_sequence": cyc_ref, "bytes_per_cycle_fused": cyc_fused,
        "achieved_reference_sequence": cyc_ref * cycles / dt / 1e9, "frac_reference_sequence": cyc_ref * cycles / dt / 1e9 / HBM_PEAK_GBPS,
        "achieved_fused": cyc_fused * cycles / dt / 1e9, "frac_fused": cyc_fused * cycles / dt / 1e9 / HBM_PEAK_GBPS, "unit": "GB/s", "peak": HBM_PEAK_GBPS,
        "frac_note": "algorithmic bytes / time / 8 TB/s; at 256^3 a vector is 134 MB: the basis vector and q of a Gram-Schmidt pass come partly "
                     "out of the 256 MB Infinity Cache, so the fractions are not HBM fractions (DESIGN.md 3.2)",
        "spmv": {"launches": spmv_l, "avg_ms": spmv_ms / max(spmv_l, 1), "bytes_per_launch": sb,
                 "frac": (sb / (spmv_ms / max(spmv_l, 1) * 1e-3) / 1e9 / HBM_PEAK_GBPS) if spmv_ms > 0 else None,
                 "kernel": f"SpMV kernel {A.spmv_kernel_choice}, code_info {list(A.code_info)}"},
        "parity": parity, "leg_seconds": None}
    del ws, A, b, ones
    out["cfg3_gmres"]["leg_seconds"] = time.time() - t_leg
    log(f"[bench] cfg 3 leg {out['cfg3_gmres']['leg_seconds']:.1f} s: {out['cfg3_gmres']['ms_per_inner_iteration']:.3f} ms per inner iteration")

    # ---------------- cfg 5: block_gmres!(memory = 5, restart = true), p = 16, 27-point 216^3 ----------------
    t_leg = time.time()
    n1, p = 216, 16
    n = n1 ** 3
    A = K.CsrMatrix.stencil(ctx, "stencil27", n1)
    t = (np.arange(n) + 1.0) / n
    Xt = np.stack([np.cos(j * np.pi * t) + 0.1 * j for j in range(p)], axis=1)                     # tests/golden/make_scale_golden.py cfg5_xtrue
    dXt = K.Panel.from_host(ctx, Xt)
    dB = K.Panel(ctx, n, p)
    K.spmm_(A, dXt, dB)                                                                            # B = A * X_true (also builds the tile records)
    del Xt, dXt
    ws = K.BlockGmresWorkspace(ctx, n, n, p, memory=5)
    K.block_gmres_(ws, A, dB, restart=True, atol=0.0, rtol=0.0, itmax=7, history=True)             # the golden's 7 iterations
    parity = _prefix_dev(ws.stats.residuals, "oracle_cfg5_block216.json", ws.X)
    ctx.set_option("profile_spmv", 1); ctx.profile_kernels()
    ctx.sync(); t0 = time.perf_counter()
    K.block_gmres_(ws, A, dB, restart=True, itmax=20, atol=0.0, rtol=0.0)                          # four whole cycles of 5
    ctx.sync(); dt = time.perf_counter() - t0
    prof = ctx.profile_kernels(); ctx.set_option("profile_spmv", 0)
    it = ws.stats.niter
    pb = 8 * n * p                                                                                 # bytes of one n x p panel
    spmm_b = 12 * A.nnz + 4 * n + 2 * pb
    # one iteration k (1..5) of a cycle as the reference issues it (src/block_gmres.jl:236-310): mul!(W, A, P) = SpMM; k x
    # (mul!(R, V_i', Q): 2 panels read; mul!(Q, V_i, R, -1, 1): 2 read + 1 written); householder!(Q): geqrf + orgqr, two panel
    # passes of read + write.  Mean over k = 1..5 (k = 3), plus per cycle the residual SpMM (+ 3 panels) and X += sum V_i Y_i (5 x 3 panels)
    it_ref = spmm_b + 3 * 5 * pb + 4 * pb
    cyc_extra_ref = spmm_b + 3 * pb + 5 * 3 * pb
    # with this library's fusions: SpMM; V_1' Q (2 panels); k fused steps Q -= V_i Psi_i ; Psi_{i+1} = V_{i+1}' Q (3 read + 1 written);
    # CholeskyQR2 (5 panel passes); per cycle the residual SpMM + 3 panels and ONE pass X += sum V_i Y_i (5 + 2 panels)
    it_fused = spmm_b + 2 * pb + 3 * 4 * pb + 5 * pb
    cyc_extra_fused = spmm_b + 3 * pb + 7 * pb
    tot_ref = it * it_ref + (it / 5.0) * cyc_extra_ref
    tot_fused = it * it_fused + (it / 5.0) * cyc_extra_fused

    def fam(tag, nbytes):
        l, ms = prof[tag]
        avg = ms / max(l, 1)
        return {"launches": l, "avg_ms": avg, "bytes_per_launch": nbytes, "achieved": (nbytes / (avg * 1e-3) / 1e9) if avg > 0 else None,
                "frac": (nbytes / (avg * 1e-3) / 1e9 / HBM_PEAK_GBPS) if avg > 0 else None}
    info = A.tile_info
    out["cfg5_block_gmres"] = {
        "NOT_THE_HEADLINE": "BASELINE cfg 5, untimed for `value`: block_gmres!(memory = 5, restart = true), p = 16, on the 27-point 216^3 operator "
                            "(10,077,696 rows, 269,586,136 entries: the SuiteSparse-shaped stand-in, no network), B = A * X_true, atol = rtol = 0, "
                            "khip_block_gmres_solve_panel on an adopted workspace",
        "iterations": int(it), "seconds": dt, "ms_per_iteration": 1e3 * dt / it,
        "bytes_per_iteration_reference_sequence": tot_ref / it, "bytes_per_iteration_fused": tot_fused / it,
        "achieved_reference_sequence": tot_ref / dt / 1e9, "frac_reference_sequence": tot_ref / dt / 1e9 / HBM_PEAK_GBPS,
        "achieved_fused": tot_fused / dt / 1e9, "frac_fused": tot_fused / dt / 1e9 / HBM_PEAK_GBPS, "unit": "GB/s", "peak": HBM_PEAK_GBPS,
        "kernels_hip_events": {
            "spmm_tile (mul!(W, A, P), src/block_gmres.jl:242)": dict(fam("spmm", spmm_b), window=info["window"], groups=info["groups"]),
            "panel_nn_tn (Q -= V Psi ; Psi' = V'^T Q, :244-247)": fam("panel_nn_tn", 4 * pb),
            "panel_gemm_tn (Psi = V^T Q, :245)": fam("panel_gemm_tn", 2 * pb),
            "panel_multi_nn (X += sum V_i Y_i, :324-326; k = 5)": fam("panel_multi_nn", 7 * pb),
            "panel_gemm_nn (Q <- Q R^-1 of the QR)": fam("panel_gemm_nn", 2 * pb),
            "panel_scale_gram (scale + Gram of CholeskyQR2)": fam("panel_qr_scale_gram", 2 * pb)},
        "parity": parity, "leg_seconds": None}
    del ws, A, dB
    out["cfg5_block_gmres"]["leg_seconds"] = time.time() - t_leg
    log(f"[bench] cfg 5 leg {out['cfg5_block_gmres']['leg_seconds']:.1f} s: {out['cfg5_block_gmres']['ms_per_iteration']:.3f} ms per iteration")
    return out


def self_consistency(n1, residuals):
    """Same history against the 1-GPU run of the UNFUSED primitive sequence (GPU vs GPU: says the fused / partitioned
    path equals the plain one, nothing about the reference)."""
    path = os.path.join(ROOT, "tests", "golden", "cg512_residuals.json")
    if n1 != 512 or not os.path.exists(path):
        return None
    dev, k = _hist_dev(residuals, json.load(open(path))["residuals"])
    if dev is None:
        return None
    return {"against": "1-GPU history of the unfused primitive sequence (tests/golden/cg512_residuals.json)",
            "iterations_compared": k, "max_rel_dev": dev}


def cpu_baseline(n1, budget_s=30.0):
    """Oracle CG loop (the reference's cg! recurrence, src/cg.jl:195-268) on the host cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ctypes as C
    import oracle as ok
    try:
        ncores = len(os.sched_getaffinity(0))
    except Exception:
        ncores = os.cpu_count() or 1
    try:
        avail_gb = int([l for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0].split()[1]) / 1e6
    except Exception:
        avail_gb = 0.0
    sample_n1 = n1
    need_gb = 17.0 * (n1 / 512.0) ** 3 * 1.3
    if avail_gb < need_gb or ncores < 16:
        sample_n1 = 256 if n1 > 256 else n1     # bounded sample when the host is small
    t0 = time.time()
    ok.lib().ko_set_threads(ncores)
    A = ok.poisson3d(sample_n1)
    t_gen = time.time() - t0
    r = C.c_double()
    iters = 5 if sample_n1 >= 512 else 8          # timed iterations, after ko_cg_bench's one untimed warm-up iteration
    # The box may cap CPU time below the visible core count (cgroup quota), where 256 OpenMP threads
    # thrash: time a ladder of thread counts and report the best.  1 thread = the reference's own
    # serial SparseMatrixCSC mul!.
    try:
        quota = open("/sys/fs/cgroup/cpu.max").read().split()
        quota_cpus = None if quota[0] == "max" else float(quota[0]) / float(quota[1])
    except Exception:
        quota_cpus = None
    # most promising first (the quota when there is one), the serial reference mode last if the budget allows
    first = int(min(ncores, max(1.0, round(quota_cpus)))) if quota_cpus else ncores
    ladder = [first] + [t for t in sorted({ncores, min(ncores, 64), min(ncores, 16), min(ncores, 4)}, reverse=True) if t != first and t != 1] + [1]
    ladder = list(dict.fromkeys(ladder))
    trials = {}
    for th in ladder:
        if trials and time.time() - t0 > budget_s:
            break
        trials[th] = ok.lib().ko_cg_bench(C.byref(A.c), iters, th, C.byref(r))
    cores, best = min(trials.items(), key=lambda kv: kv[1])
    sample = (f"1 untimed warm-up + {iters} timed CG iterations of the oracle loop (cg! recurrence, OpenMP SpMV/dot/axpy) on get_div_grad({sample_n1}^3); "
              + "it/s by thread count: " + ", ".join(f"{th}: {1.0 / v:.3f}" for th, v in trials.items())
              + f"; visible cores {ncores}, cgroup cpu quota {quota_cpus}; matrix generation {t_gen:.1f} s excluded")
    return {"value": 1.0 / best, "unit": "iter/s", "cores": cores, "kind": "port", "sample": sample,
            "sample_n1": sample_n1}


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(args, argv):
    """`python bench.py --gpus N` outside a launcher: become the launcher.  Spawns N children of this very script,
    one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment -- exactly what
    `python -m torch.distributed.run --nproc-per-node N` would set, which keeps working: a process that already has
    WORLD_SIZE never comes here).  Rank 0 inherits this process's stdout (the one JSON line); the other ranks' stdout
    goes to stderr.  Refuses (rc 2) when fewer than N devices are visible: N ranks never silently share a GPU or
    shrink to one.  Returns the worst child return code.  ref: docs/src/custom_workspaces.md:583-637 (mpiexecjl -n 4)."""
    import subprocess
    n = args.gpus
    if not args.dry_launch:
        import krylov_jl_amd as K
        ndev = K.device_count() if K.gpu_available() else 0
        if ndev < n:
            log(f"bench.py: --gpus {n} asked for, {ndev} HIP device(s) visible: refusing to run "
                f"(no fallback to fewer ranks, no sharing of a GPU between ranks)")
            return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   KHIP_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rcs = [None] * n
    deadline = time.time() + float(os.environ.get("KHIP_BENCH_LAUNCH_TIMEOUT", "3600"))
    while any(rc is None for rc in rcs):
        for i, p in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = p.poll()
        failed = [rc for rc in rcs if rc not in (None, 0)]
        if (failed or time.time() > deadline) and any(rc is None for rc in rcs):
            time.sleep(5.0)                       # let the others notice (gloo errors out when a peer is gone)
            for i, p in enumerate(procs):         # then stop exactly the processes started here
                if p.poll() is None:
                    p.kill()
                rcs[i] = p.wait()
            if not failed:
                log("bench.py: launch timed out")
                return 124
            break
        time.sleep(0.05)
    worst = max((abs(rc) for rc in rcs), default=0)
    if worst:
        log(f"bench.py: rank return codes {rcs}")
    return worst


PHASES = ("halo_pack", "halo_transfer", "spmv", "spmv_boundary", "dot_allgather_combine")


def summarize_phases(profiles, steps):
    """Per-rank phase times of one iteration from the HIP-event brackets of the timed solve (Context.profile_kernels on every rank,
    gathered over the control plane): the pack kernel, the halo transfer on its own stream, the interior and the boundary launch of
    the product, each dot's 16-byte all-gather + combine.  `checks` evaluates the falsifiers DESIGN.md section 5 names for the
    8-GPU prediction, so that the one scaling run the driver may get explains itself."""
    rows = []
    for r, prof in enumerate(profiles):
        row = {"rank": r}
        for k in PHASES:
            l, ms = prof.get(k, (0, 0.0))
            row[k] = {"launches_per_iteration": l / max(steps, 1), "ms_per_iteration": ms / max(steps, 1), "avg_us": (1e3 * ms / l) if l else None}
        rows.append(row)
    dots = [x["dot_allgather_combine"]["avg_us"] for x in rows if x["dot_allgather_combine"]["avg_us"] is not None]
    xfer = [x["halo_transfer"]["avg_us"] for x in rows if x["halo_transfer"]["avg_us"] is not None]
    inter = [x["spmv"]["avg_us"] for x in rows if x["spmv"]["avg_us"] is not None]
    checks = {"dot_allgather_combine_avg_us_max": max(dots) if dots else None,
              "falsifier_allgather_above_50us": bool(dots and max(dots) > 50.0),
              "halo_transfer_avg_us_max": max(xfer) if xfer else None, "spmv_interior_avg_us_min": min(inter) if inter else None,
              "falsifier_halo_longer_than_interior_product": bool(xfer and inter and max(xfer) > min(inter)),
              "reading": "DESIGN.md 5: the N = 8 prediction (0.53-0.57 ms per iteration) fails if an 8-rank 16-byte all-gather + combine costs "
                         "more than ~50 us (two per iteration), or if the halo transfer does not finish under the interior product"}
    return {"per_rank": rows, "checks": checks}


def dry_launch(rank, world, json_fd):
    """--dry-launch: the launch path without the GPU -- every rank joins the gloo group (the control plane of the real run)
    and the ranks are counted by an all-reduce; rank 0 prints the one JSON line."""
    if os.environ.get("KHIP_BENCH_TEST_FAIL_RANK") == str(rank):      # tests/test_bench_launch.py: a rank that dies before the rendezvous
        log(f"bench.py: rank {rank} fails on request (KHIP_BENCH_TEST_FAIL_RANK)")
        os._exit(7)
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world == 1:
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    t = torch.ones(1, dtype=torch.float64)
    dist.all_reduce(t)
    ranks = [None] * world
    dist.all_gather_object(ranks, (rank, int(os.environ.get("LOCAL_RANK", "-1")), os.getpid()))
    # the per-rank phase report of the real run, on synthetic brackets (no GPU here): same gather, same summary
    steps = 10
    fake = {"spmv": (steps, 2.5 * steps), "spmv_boundary": (steps, 0.01 * steps), "halo_pack": (steps, 0.004 * steps),
            "halo_transfer": (steps, (0.2 + 0.01 * rank) * steps), "dot_allgather_combine": (2 * steps, 0.02 * (rank + 1) * 2 * steps)}
    profs = [None] * world
    dist.all_gather_object(profs, fake if world > 1 else {"spmv": (steps, 2.5 * steps)})
    dist.barrier()
    if rank == 0:
        os.write(json_fd, (json.dumps({"dry_launch": True, "n_gpus": world, "ranks_rendezvoused": int(t.item()),
                                       "ranks": [list(r) for r in ranks], "phases": summarize_phases(profs, steps),
                                       "ab": {k: None for k in ("overlap_halo_0", "comm_priority_0")},
                                       "launcher": "bench.py" if os.environ.get("KHIP_BENCH_SELF_LAUNCHED") else "external"}) + "\n").encode())
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--n1", type=int, default=512, help="grid size per dimension (default: cfg 2, 512^3)")
    ap.add_argument("--fused", type=int, default=2,
                    help="0 = reference primitive sequence, 1 = fused kernels, 2 = fused + device-resident scalars")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--compress", action="store_true",
                    help="NOT the headline: re-encode the operator as row templates (khip_csr_compress, 2 B of matrix data per row)")
    ap.add_argument("--variant", type=int, default=0,
                    help="NOT the headline: 1 = single-reduction CG (one all-reduce per iteration; different rounding)")
    ap.add_argument("--opt", action="append", default=[], help="tuning knob key=value (khip_ctx_set_option)")
    ap.add_argument("--no-full-parity", action="store_true", help="skip the (untimed) full solve to rtol 1e-8 of the parity leg")
    ap.add_argument("--no-ab", action="store_true", help="N > 1: skip the A/B legs (overlap_halo = 0, comm_priority = 0)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the (untimed for `value`) legs of BASELINE cfg 3 (gmres!) and cfg 5 (block_gmres!)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="exercise only the N-rank launch + rendezvous (gloo), no GPU work: tests/test_bench_launch.py")
    ap.add_argument("--also-variant1", action="store_true",
                    help="also time single-reduction CG and report it as the nested entry single_reduction_cg (always done with --gpus > 1)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))      # no launcher around us: be the launcher (one rank per GPU)

    # stdout carries exactly one JSON line: anything a library prints there (RCCL writes a version banner to stdout at
    # communicator creation) is sent to stderr by pointing fd 1 at fd 2 for the life of the process.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not (world == 1 and args.gpus <= 1):
        log(f"bench.py: WARNING --gpus {args.gpus} but the launcher started WORLD_SIZE = {world} ranks; "
            f"running (and reporting n_gpus =) {world}")
    if args.dry_launch:
        dry_launch(rank, world, json_fd)
        return
    dist = None
    force_comm = os.environ.get("KHIP_FORCE_COMM") == "1"      # exercise the distributed path with one rank
    if world > 1 or "RANK" in os.environ:
        import torch                      # torch FIRST: one shared HIP runtime in the process
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)   # control plane only
    import numpy as np
    import krylov_jl_amd as K

    # one rank per GPU, never two ranks on one device: device = LOCAL_RANK, unless the launcher restricted this
    # process to ONE visible device (then that device is this rank's).  KHIP_FORCE_COMM / KHIP_ALLOW_SHARED_GPU are
    # the explicit test switches for running the distributed path on a 1-GPU box.
    ndev = K.device_count() if K.gpu_available() else 0
    restricted = any(os.environ.get(v) for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
    allow_shared = os.environ.get("KHIP_ALLOW_SHARED_GPU") == "1"
    if local_rank < ndev:
        dev = local_rank
    elif ndev == 1 and (restricted or allow_shared):
        dev = 0
    else:
        log(f"bench.py: rank {rank} (local rank {local_rank}) has no device of its own: {ndev} HIP device(s) visible, "
            f"{world} ranks; refusing to share a GPU between ranks")
        sys.exit(2)
    # A *_VISIBLE_DEVICES exported to EVERY rank (not per rank) would put them all on one GPU while n_gpus = WORLD_SIZE is
    # reported: compare the PHYSICAL devices (host, PCI bus id) over the control plane and refuse duplicates.
    if dist is not None and world > 1 and not allow_shared:
        import socket
        ids = [None] * world
        dist.all_gather_object(ids, (socket.gethostname(), K.device_pci_id(dev)))
        if len(set(ids)) != world:
            if rank == 0:
                log(f"bench.py: {world} ranks on {len(set(ids))} physical GPU(s) {sorted(set(ids))}: refusing to share a GPU "
                    f"between ranks (KHIP_ALLOW_SHARED_GPU=1 is the explicit test switch)")
            sys.exit(2)
    ctx = K.Context(dev)
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    use_comm = world > 1 or force_comm
    if use_comm:
        uid = [K.Context.comm_unique_id() if rank == 0 else None]
        if dist is not None:
            dist.broadcast_object_list(uid, src=0)
        ctx.comm_init(rank, world, uid[0])

    n1 = args.n1
    n = n1 ** 3
    starts = K.row_partition(n, world)
    r0, r1 = starts[rank], starts[rank + 1]
    nloc = r1 - r0
    t_setup = time.time()
    A = K.CsrMatrix.stencil(ctx, "poisson", n1, rows=(r0, r1), distributed=use_comm)
    templates = A.compress() if args.compress else 0
    b = ctx.empty(nloc)
    K.kfill_(b, 1.0)
    ws = K.CgWorkspace(ctx, nloc, nloc)
    ctx.sync()
    log(f"[rank {rank}] setup {time.time() - t_setup:.2f} s, local rows {nloc}, nnz {A.nnz}")

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()

    # warm-up iterations (untimed)
    if args.warmup > 0:
        K.cg_(ws, A, b, atol=0.0, rtol=0.0, itmax=args.warmup, fused=args.fused, variant=args.variant)
    # Row-partitioned runs: should the halo exchange run UNDER the interior rows (overlap_halo = 1: two launches of the product and a
    # cross-stream dependency, ~30-70 us of overhead per iteration on one rank, tools/slab_iteration.py) or BEFORE one launch over
    # all rows (0)?  That depends on what the transfer costs on this node's links, which no 1-GPU box can show -- so it is probed
    # here, untimed, during warm-up: a few iterations of each, max over ranks, and the headline runs with the faster one (the default
    # unless the other is > 2 % faster).  y and the residual history are bit-identical either way; both probe times are in the line,
    # and the A/B leg after the timed region repeats the OTHER setting over the full K iterations.
    halo_probe = None
    if use_comm and args.variant == 0 and not args.no_ab and not any(kv.startswith("overlap_halo=") for kv in args.opt):
        default_overlap = ctx.get_option("overlap_halo")
        probe_iters = max(5, min(args.steps, 20))
        ms = {}
        try:
            for val in (1, 0):
                ctx.set_option("overlap_halo", val)
                K.cg_(ws, A, b, atol=0.0, rtol=0.0, itmax=3, fused=args.fused)
                barrier()
                tp = time.perf_counter()
                K.cg_(ws, A, b, atol=0.0, rtol=0.0, itmax=probe_iters, fused=args.fused)
                barrier()
                el = time.perf_counter() - tp
                if dist is not None and world > 1:
                    import torch
                    tt = torch.tensor([el], dtype=torch.float64)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    el = float(tt.item())                      # the same number on every rank: the same choice on every rank
                ms[val] = 1e3 * el / max(ws.stats.niter, 1)
            other = 1 - default_overlap
            chosen = other if ms[other] < 0.98 * ms[default_overlap] else default_overlap
            halo_probe = {"chosen_overlap_halo": chosen, "default": default_overlap, "iterations_each": probe_iters,
                          "ms_per_iteration": {f"overlap_halo_{v}": ms[v] for v in (1, 0)},
                          "rule": "the non-default setting only if it is > 2 % faster (max over ranks)"}
        except Exception as e:              # a probe must not cost the run: keep the default
            chosen = default_overlap
            halo_probe = {"chosen_overlap_halo": chosen, "default": default_overlap, "error": f"{type(e).__name__}: {e}"}
        ctx.set_option("overlap_halo", chosen)
    ctx.set_option("profile_spmv", 1)
    ctx.profile_kernels()
    barrier()
    t0 = time.perf_counter()
    done, first_hist, solves = 0, None, 0
    while done < args.steps:        # exactly K iterations: a solve that stagnates before K (K >> 1500 at 512^3) is followed by another
        K.cg_(ws, A, b, atol=0.0, rtol=0.0, itmax=args.steps - done, history=True, fused=args.fused, variant=args.variant)
        solves += 1
        if first_hist is None:
            first_hist = ws.stats.residuals.copy()
        if ws.stats.niter == 0:
            break
        done += ws.stats.niter
    barrier()
    elapsed = time.perf_counter() - t0
    st = ws.stats
    assert done == args.steps, (done, st.status)
    prof_main = ctx.profile_kernels()
    launches = prof_main["spmv"][0] + prof_main["spmv_boundary"][0]        # interior + boundary launches of a row-partitioned product
    spmv_ms = prof_main["spmv"][1] + prof_main["spmv_boundary"][1]
    ctx.set_option("profile_spmv", 0)
    # N > 1: every rank's phase times to rank 0; then the same K iterations with the halo overlap off and with the communication
    # stream at default priority (an A/B inside ONE launch: the one scaling run the driver may get should explain itself)
    phases, ab = None, None
    if use_comm:
        profs = [prof_main]
        if dist is not None and world > 1:
            profs = [None] * world
            dist.all_gather_object(profs, prof_main)
        phases = summarize_phases(profs, args.steps)
        ab = {}
        if args.variant == 0 and not args.no_ab:
            other_overlap = 1 - ctx.get_option("overlap_halo")           # the setting the headline did NOT run with
            for name, key, val in ((f"overlap_halo_{other_overlap}", "overlap_halo", other_overlap), ("comm_priority_0", "comm_priority", 0)):
                old_val = ctx.get_option(key)
                try:
                    ctx.set_option(key, val)
                    K.cg_(ws, A, b, atol=0.0, rtol=0.0, itmax=max(min(args.warmup, 5), 1), fused=args.fused)
                    barrier()
                    ta = time.perf_counter()
                    K.cg_(ws, A, b, atol=0.0, rtol=0.0, itmax=args.steps, fused=args.fused)
                    barrier()
                    el = time.perf_counter() - ta
                    its_ab = ws.stats.niter
                except Exception as e:      # never lose the headline line to a side leg (the library's errors are the same on every rank)
                    ab[name] = {"error": f"{type(e).__name__}: {e}"}
                    continue
                finally:
                    ctx.set_option(key, old_val)
                if dist is not None and world > 1:
                    import torch
                    tt = torch.tensor([el], dtype=torch.float64)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    el = float(tt.item())
                ab[name] = {"value": its_ab / el, "unit": "iter/s", "steps": int(its_ab), "ms_per_step": 1e3 * el / max(its_ab, 1),
                            "setting": f"{key} = {val} (the headline runs with {key} = {old_val})"}
    # parity leg (untimed): the first PARITY_ITERS residual norms of a fresh solve against the CPU oracle's
    parity_hist = first_hist
    if n1 == 512 and len(first_hist) <= PARITY_ITERS:
        K.cg_(ws, A, b, atol=0.0, rtol=0.0, itmax=PARITY_ITERS, history=True, fused=args.fused, variant=args.variant)
        parity_hist = ws.stats.residuals.copy()
    # ... and (untimed) the benchmark definition's full solve to convergence: iteration count, status, whole history
    full = None
    if n1 == 512 and args.variant == 0 and not args.no_full_parity:
        K.cg_(ws, A, b, atol=0.0, rtol=1e-8, itmax=n, history=True, fused=args.fused)
        full = {"niter": ws.stats.niter, "status": ws.stats.status, "residuals": ws.stats.residuals.copy()}
    # single-reduction CG (Chronopoulos-Gear: ONE all-reduce per iteration) timed the same way, reported as a nested,
    # clearly labelled entry beside the headline whenever the run is partitioned over several ranks
    sr = None
    sr_error = None
    if (world > 1 or args.also_variant1) and args.variant == 0:
        try:
            K.cg_(ws, A, b, atol=0.0, rtol=0.0, itmax=max(args.warmup, 1), fused=args.fused, variant=1)
            barrier()
            t1 = time.perf_counter()
            K.cg_(ws, A, b, atol=0.0, rtol=0.0, itmax=args.steps, fused=args.fused, variant=1)
            barrier()
            sr_elapsed = time.perf_counter() - t1
            sr_iters = ws.stats.niter
        except Exception as e:              # a side leg: report, keep the headline
            sr_error = f"{type(e).__name__}: {e}"
            sr_elapsed, sr_iters = 1.0, 0
        if dist is not None:
            import torch
            tt = torch.tensor([sr_elapsed], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            sr_elapsed = float(tt.item())
        sr = {"NOT_THE_HEADLINE": "single-reduction CG (options.variant = 1): rearranged recurrence, one all-reduce per iteration, "
                                  "different rounding (own parity budget, DESIGN.md 3.1c)",
              "value": sr_iters / sr_elapsed, "unit": "iter/s", "steps": int(sr_iters), "ms_per_step": 1e3 * sr_elapsed / max(sr_iters, 1)}
        if sr_error is not None:
            sr = {"NOT_THE_HEADLINE": sr["NOT_THE_HEADLINE"], "error": sr_error}
    code_bits, code_diags = A.code_info
    # General-CSR leg (untimed for `value`, clearly labelled): the SAME operator and fused iteration with the dictionary codes
    # switched off, i.e. the kernel that moves every algorithmic byte of SURVEY 8(d) (12 B per entry: Float64 value + int32
    # column).  HIP-event average over the same number of steps.
    int32_leg = None
    if world == 1 and code_bits != 32 and not templates and args.variant == 0:
        saved = {k: ctx.get_option(k) for k in ("spmv_codes", "profile_spmv")}     # a user's --opt spmv_codes=... survives the leg
        ctx.set_option("spmv_codes", 0)
        try:
            K.cg_(ws, A, b, atol=0.0, rtol=0.0, itmax=max(args.warmup, 1), fused=args.fused)
            ctx.set_option("profile_spmv", 1)
            ctx.profile_spmv()
            barrier()
            t2 = time.perf_counter()
            K.cg_(ws, A, b, atol=0.0, rtol=0.0, itmax=args.steps, fused=args.fused)
            barrier()
            el32 = time.perf_counter() - t2
            l32, ms32 = ctx.profile_spmv()
            ctx.set_option("profile_spmv", 0)
            it32 = ws.stats.niter
            avg32 = ms32 / max(it32, 1)
            int32_leg = {"NOT_THE_HEADLINE": "same operator, same fused cg! iteration, ctx option spmv_codes = 0: the staged SpMV reads the "
                                             "int32 column stream (12 B per entry) -- the general-CSR kernel, for the '>= 70 % on CSR SpMV' target",
                         "bound": "hbm", "kernel": (f"SpMV kernel {A.spmv_kernel_choice} (4 = the staged family: " +
                                                    ("spmv_sell_kernel on the sliced copy with int32 columns, khip_csr_sell32_info " + str(list(A.sell32_info))
                                                     if A.sell32_info[0] == 1 and ctx.get_option("spmv_sell") else "spmv_stage_kernel") +
                                                    ") fused with p.Ap, int32 columns"),
                         "achieved": A.spmv_bytes / (avg32 * 1e-3) / 1e9 if avg32 > 0 else 0.0, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": (A.spmv_bytes / (avg32 * 1e-3) / 1e9 / HBM_PEAK_GBPS) if avg32 > 0 else 0.0,
                         "bytes_per_launch": A.spmv_bytes, "avg_ms": avg32, "launches_per_iteration": l32 / max(it32, 1),
                         "steps": int(it32), "cg_iters_per_sec": it32 / el32, "ms_per_step": 1e3 * el32 / max(it32, 1)}
        finally:
            for k, v in saved.items():
                ctx.set_option(k, v)
    others = None
    if world == 1 and n1 == 512 and args.variant == 0 and not args.no_other_configs and not templates:
        try:
            others = other_configs(K, ctx, log)
        except Exception as e:          # never lose the headline line to a side leg
            others = {"error": f"{type(e).__name__}: {e}"}
    rccl_ranks = ctx.comm_info()["rccl_ranks"] if use_comm else 0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        its = args.steps / elapsed
        spmv_bytes_local = A.spmv_bytes_stored if templates else A.spmv_bytes     # compressed: what that format moves
        spmv_bytes_moved = A.spmv_bytes_stored          # bytes the kernel streams in the handle's current representation
        # algorithmic bytes of one fused iteration on this rank: SpMV(+dot) + (r update + r.r: 24n) + (x and p update: 40n)
        iter_bytes_local = spmv_bytes_local + ((72 if args.variant == 1 else 64) if args.fused else 104) * nloc
        iter_bytes_unfused_local = spmv_bytes_local + 104 * nloc       # as the reference issues it
        spmv_per_iter = launches / max(args.steps, 1)
        avg_spmv_ms = spmv_ms / max(args.steps, 1)                    # all SpMV launches of one iteration
        spmv_gbps = spmv_bytes_local / (avg_spmv_ms * 1e-3) / 1e9 if avg_spmv_ms > 0 else 0.0
        moved_gbps = spmv_bytes_moved / (avg_spmv_ms * 1e-3) / 1e9 if avg_spmv_ms > 0 else 0.0
        sliced = code_bits == 8 and A.sell_info[0] == 1 and ctx.get_option("spmv_sell") != 0     # the sliced form of the coded operator is in use
        kern = "spmv_template_kernel" if templates else ("spmv_sell_kernel" if sliced else ("spmv_code_kernel" if code_bits != 32 else "spmv_stage_kernel"))
        pmc_key = {8: "spmv_code_kernel<unsigned char, true, true", 16: "spmv_code_kernel<unsigned short, true, true",
                   32: "spmv_stage_kernel<256, false, true, true"}[code_bits]
        if sliced:
            pmc_key = "spmv_sell_kernel<true, true, false, " + ("true" if ctx.get_option("spmv_sell") == 2 else "false") + ", false, "   # DOT, COMP, DIST, non-temporal loads, int32 columns = false; the layout flags (narrow codes, 16-byte pairs) follow
        traffic, traffic_note = pmc_traffic(n1, pmc_key) if (world == 1 and not templates) else (None, "single-GPU CSR runs only")
        col_note = ("int32 columns" if code_bits == 32 else
                    f"{code_bits}-bit diagonal codes ({code_diags} distinct column - row offsets, csrc/colcode.hip)")
        if sliced:
            col_note += ("; values and codes read from the sliced copy of the operator (every 64 rows transposed: per row one word of eight codes + "
                         "its values, 64 B per 7-entry row; khip_csr_sell_info " + str(list(A.sell_info)) + ")")
        out = {
            "metric": "cg_iters_per_sec_poisson3d_csr_512cubed",
            "value": its, "unit": "iter/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"cg! on get_div_grad({n1},{n1},{n1}) CSR (cfg 2), b=ones, Float64, int32 indices",
                       "n": n, "nnz_global": 7 * n - 6 * n1 * n1, "fused": args.fused,
                       "partition": f"1-D rows over {world} GPU(s)", "atol": 0.0, "rtol": 0.0,
                       "operator_format": f"row templates ({templates})" if templates else f"CSR; column stream read as {col_note}",
                       "recurrence": "single-reduction CG (Chronopoulos-Gear)" if args.variant == 1 else "cg! (src/cg.jl)",
                       "entry": ("khip_cg_solve via khip_cg_workspace_adopt (x, r, p, Ap owned by the caller, as a Krylov.jl "
                                 "CgWorkspace{Float64,Float64,HIPVector} holds them: julia/KrylovHIP/src/KrylovHIP.jl cg!)")
                                if ws.adopted else "khip_cg_solve on a khip_cg_workspace_create workspace"},
            "hbm_gbps_iteration": its * iter_bytes_local * world / 1e9,
            "bytes_per_iteration_algorithmic_fused": iter_bytes_local * world,
            "bytes_per_iteration_reference_sequence": iter_bytes_unfused_local * world,
            "final_residual_norm": float(first_hist[-1]), "solves_in_timed_region": solves,
            "parity": parity_vs_oracle(n1, parity_hist, full),
            "self_consistency": self_consistency(n1, parity_hist),
            "rccl_ranks_seen": rccl_ranks,
            "comm": (dict(ctx.comm_info(), halo=dict(zip(("gather_mode", "n_ghost", "n_send"), A.halo_info))) if use_comm else None),
            "phases": phases, "ab": ab, "halo_probe": halo_probe,
            "single_reduction_cg": sr,
            "roofline_int32_csr": int32_leg,
            "cfg3_gmres": (others or {}).get("cfg3_gmres"), "cfg5_block_gmres": (others or {}).get("cfg5_block_gmres"),
            "other_configs_error": (others or {}).get("error"),
            "roofline": {"bound": "hbm", "kernel": kern + " (SpMV fused with p.Ap)",
                         "achieved": spmv_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": spmv_gbps / HBM_PEAK_GBPS, "traffic": traffic, "traffic_note": traffic_note,
                         "bytes_per_launch": spmv_bytes_local,
                         "bytes_per_launch_note": "ALGORITHMIC bytes, SURVEY 8(d): 12 nnz + 4 (m+1) + 8 n + 8 m (CSR with int32 columns)",
                         "bytes_moved_per_launch": spmv_bytes_moved, "achieved_moved": moved_gbps,
                         "frac_moved": moved_gbps / HBM_PEAK_GBPS,
                         "bytes_moved_note": "what the kernel actually streams: " + col_note,
                         "avg_ms": avg_spmv_ms, "launches_per_iteration": spmv_per_iter,
                         "kernel_source_sha": kernel_source_sha()},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(n1)
            except Exception as e:          # never lose the GPU line to a host-side problem
                out["cpu_baseline"] = {"value": None, "unit": "iter/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": f"failed: {e}"}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
