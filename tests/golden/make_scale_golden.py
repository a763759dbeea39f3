"""Oracle residual histories AT THE BASELINE SIZES -> tests/golden/oracle_cfg{2,3,5}.json.

  cfg 2: cg! on get_div_grad(512,512,512), b = ones, atol = rtol = 0, 100 iterations        (src/cg.jl:195-268)
  cfg 3: gmres!(memory = 30, restart = true) on kron_unsymmetric(256), b = A*ones, 45 inner
         iterations = one full cycle, the restart, and half of the second cycle            (src/gmres.jl:237-330)
  cfg 5: block_gmres!(memory = 5, restart = true) on the 27-point 216^3 operator, p = 16,
         B = A*X_true, 7 iterations = one cycle, the restart, two more                     (src/block_gmres.jl:236-310)

Everything is the oracle's own arithmetic (oracle/krylov_oracle.c: serial extended-precision dots, fma axpys,
unblocked Householder QR).  The operator products run row-parallel (ko_spmv_omp): rows are independent, so each y value
is the serial loop's; likewise the p x p entries of the oracle's panel products and the columns its reflectors update
are independent, so the thread count changes no value.  These are golden vectors OF THE ORACLE: "parity unpinned" with
respect to Krylov.jl itself (no Julia in the image), exactly like tests/golden/oracle_histories.json.

  leg 4 (not a BASELINE config; bicgstab! is the fourth north-star solver): bicgstab! on cfg 3's operator kron_unsymmetric(256),
         b = A*ones, 25 iterations, together with the binary128 history of the same recurrence (oracle/quad_reference.c,
         make -C oracle quadref; ~20 minutes) and the double-precision oracle's distance to it: the tolerance of the GPU test
         is derived from that distance                                     (src/bicgstab.jl:125-277)

Run (needs ~20 GB of RAM for cfg 2, a few minutes on 8 cores):  python tests/golden/make_scale_golden.py [2] [3] [5] [4]
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import oracle as ok  # noqa: E402

ok.PARALLEL_MATVEC = True
ok.lib().ko_set_threads(len(os.sched_getaffinity(0)))
which = [int(a) for a in sys.argv[1:]] or [2, 3, 5]
SAMPLE = 16          # solution entries kept as a second, x-level check


def sample_idx(n):
    return [int(i) for i in np.linspace(0, n - 1, SAMPLE).astype(np.int64)]


def cfg5_xtrue(n, p):
    """The well-conditioned trigonometric family of tests/test_gpu_block.py::_rhs."""
    t = (np.arange(n) + 1.0) / n
    return np.stack([np.cos(j * np.pi * t) + 0.1 * j for j in range(p)], axis=1)


def dump(name, d):
    path = os.path.join(HERE, name)
    json.dump(d, open(path, "w"), indent=1)
    print("wrote", path, flush=True)


if 2 in which:
    t0 = time.time()
    n1, iters = 512, 100
    A = ok.poisson3d(n1)
    b = np.ones(A.n)
    res = ok.cg(A, b, atol=0.0, rtol=0.0, itmax=iters, history=True)
    assert res.niter == iters, (res.niter, res.status)
    idx = sample_idx(A.n)
    dump("oracle_cfg2_cg512.json", dict(
        generator="tests/golden/make_scale_golden.py", oracle="oracle/krylov_oracle.c ko_cg (src/cg.jl:120-291)",
        config="BASELINE cfg 2: cg! on get_div_grad(512,512,512), b = ones, x0 = 0, atol = rtol = 0",
        n=A.n, nnz=A.nnz, niter=res.niter, status=res.status,
        residuals=[float(v) for v in res.residuals], x_index=idx, x_sample=[float(res.x[i]) for i in idx],
        seconds=time.time() - t0))
    del A, b, res

if 3 in which:
    t0 = time.time()
    n1, mem, iters = 256, 30, 45
    A = ok.kron_unsymmetric(n1)
    b = A.matvec(np.ones(A.n))
    res = ok.gmres(A, b, memory=mem, restart=True, atol=0.0, rtol=0.0, itmax=iters, history=True)
    assert res.niter == iters, (res.niter, res.status)
    idx = sample_idx(A.n)
    dump("oracle_cfg3_gmres256.json", dict(
        generator="tests/golden/make_scale_golden.py", oracle="oracle/krylov_oracle.c ko_gmres (src/gmres.jl:121-384)",
        config="BASELINE cfg 3: gmres!(memory = 30, restart = true) on kron_unsymmetric(256), b = A*ones, atol = rtol = 0",
        n=A.n, nnz=A.nnz, memory=mem, niter=res.niter, status=res.status,
        residuals=[float(v) for v in res.residuals], x_index=idx, x_sample=[float(res.x[i]) for i in idx],
        seconds=time.time() - t0))
    del A, b, res

if 4 in which:
    import struct
    import subprocess
    import tempfile
    n1, iters = 256, 25
    A = ok.kron_unsymmetric(n1)
    b = A.matvec(np.ones(A.n))
    ref = ok.bicgstab(A, b, atol=0.0, rtol=0.0, itmax=iters, history=True)
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:      # case file of oracle/quad_reference.c
        f.write(struct.pack("8i", 2, 1, n1, 1, 0, 0, 0, iters))
        f.write(struct.pack("2d", 0.0, 0.0))
        f.write(b.tobytes())
        case = f.name
    out = json.loads(subprocess.check_output([os.path.join(HERE, "..", "..", "oracle", "_ref", "quad_reference"), case]))
    os.unlink(case)
    hq, hd = np.array(out["residuals"]), ref.residuals
    idx = sample_idx(A.n)
    dump("oracle_bicgstab256.json", dict(
        generator="tests/golden/make_scale_golden.py (leg 4)",
        config="bicgstab! on kron_unsymmetric(256), b = A*ones, atol = rtol = 0, 25 iterations (cfg 3's operator)",
        n=A.n, nnz=A.nnz, niter=ref.niter, status=ref.status, residuals=[float(v) for v in hd], x_index=idx,
        x_sample=[float(ref.x[i]) for i in idx], quad_residuals=out["residuals"],
        oracle_double_max_rel_dev=float(np.max(np.abs(hd - hq) / hq))))
    del A, b, ref

if 5 in which:
    t0 = time.time()
    n1, p, mem, iters = 216, 16, 5, 7
    A = ok.stencil27_unsym(n1)
    Xt = cfg5_xtrue(A.n, p)
    B = np.stack([A.matvec(np.ascontiguousarray(Xt[:, j])) for j in range(p)], axis=1)
    res = ok.block_gmres(A, B, memory=mem, restart=True, atol=0.0, rtol=0.0, itmax=iters, history=True)
    assert res.niter == iters, (res.niter, res.status)
    idx = sample_idx(A.n)
    dump("oracle_cfg5_block216.json", dict(
        generator="tests/golden/make_scale_golden.py",
        oracle="oracle/krylov_oracle.c ko_block_gmres (src/block_gmres.jl:110-358)",
        config="BASELINE cfg 5: block_gmres!(memory = 5, restart = true), p = 16, 27-point 216^3 operator "
               "(ko_csr_stencil27_unsym), B = A*X_true with X_true[i, j] = cos(j pi (i+1)/n) + 0.1 j, atol = rtol = 0",
        n=A.n, nnz=A.nnz, p=p, memory=mem, niter=res.niter, status=res.status,
        residuals=[float(v) for v in res.residuals], x_index=idx,
        x_sample=[[float(v) for v in res.x[i]] for i in idx], seconds=time.time() - t0))

# ---------------------------------------------------------------------------------------------------------------------
# Full solves TO CONVERGENCE at the BASELINE sizes: the benchmark definition of the reference,
# cg(A, b, atol = 0.0, rtol = 1.0e-8, itmax = n) (benchmark/benchmarks.jl:14-21), applied to all three configs.
# Legs 12 / 13 / 15 = cfg 2 / 3 / 5.  One-off: ~40 / ~30 / ~15 minutes on 8 cores.  The whole history is kept so the GPU
# test can compare every iterate, the iteration count and the status string.
# ---------------------------------------------------------------------------------------------------------------------
FULL_RTOL = 1.0e-8

if 12 in which:
    t0 = time.time()
    A = ok.poisson3d(512)
    b = np.ones(A.n)
    res = ok.cg(A, b, atol=0.0, rtol=FULL_RTOL, itmax=A.n, history=True)
    idx = sample_idx(A.n)
    dump("oracle_cfg2_cg512_full.json", dict(
        generator="tests/golden/make_scale_golden.py 12", oracle="oracle/krylov_oracle.c ko_cg (src/cg.jl:120-291)",
        config="BASELINE cfg 2 to convergence: cg!(get_div_grad(512,512,512), ones; atol = 0, rtol = 1e-8, itmax = n) "
               "(benchmark/benchmarks.jl:14-21)",
        n=A.n, nnz=A.nnz, atol=0.0, rtol=FULL_RTOL, niter=res.niter, solved=bool(res.solved), status=res.status,
        residuals=[float(v) for v in res.residuals], x_index=idx, x_sample=[float(res.x[i]) for i in idx],
        seconds=time.time() - t0))
    del A, b, res

if 13 in which:
    t0 = time.time()
    A = ok.kron_unsymmetric(256)
    b = A.matvec(np.ones(A.n))
    res = ok.gmres(A, b, memory=30, restart=True, atol=0.0, rtol=FULL_RTOL, itmax=A.n, history=True)
    idx = sample_idx(A.n)
    dump("oracle_cfg3_gmres256_full.json", dict(
        generator="tests/golden/make_scale_golden.py 13", oracle="oracle/krylov_oracle.c ko_gmres (src/gmres.jl:121-384)",
        config="BASELINE cfg 3 to convergence: gmres!(memory = 30, restart = true) on kron_unsymmetric(256), b = A*ones, "
               "atol = 0, rtol = 1e-8, itmax = n",
        n=A.n, nnz=A.nnz, memory=30, atol=0.0, rtol=FULL_RTOL, niter=res.niter, solved=bool(res.solved), status=res.status,
        residuals=[float(v) for v in res.residuals], x_index=idx, x_sample=[float(res.x[i]) for i in idx],
        seconds=time.time() - t0))
    del A, b, res

if 15 in which:
    t0 = time.time()
    n1, p, mem = 216, 16, 5
    A = ok.stencil27_unsym(n1)
    Xt = cfg5_xtrue(A.n, p)
    B = np.stack([A.matvec(np.ascontiguousarray(Xt[:, j])) for j in range(p)], axis=1)
    res = ok.block_gmres(A, B, memory=mem, restart=True, atol=0.0, rtol=FULL_RTOL, itmax=A.n, history=True)
    idx = sample_idx(A.n)
    dump("oracle_cfg5_block216_full.json", dict(
        generator="tests/golden/make_scale_golden.py 15",
        oracle="oracle/krylov_oracle.c ko_block_gmres (src/block_gmres.jl:110-358)",
        config="BASELINE cfg 5 to convergence: block_gmres!(memory = 5, restart = true), p = 16, 27-point 216^3 operator "
               "(ko_csr_stencil27_unsym), B = A*X_true, X_true[i, j] = cos(j pi (i+1)/n) + 0.1 j, atol = 0, rtol = 1e-8",
        n=A.n, nnz=A.nnz, p=p, memory=mem, atol=0.0, rtol=FULL_RTOL, niter=res.niter, solved=bool(res.solved),
        status=res.status, residuals=[float(v) for v in res.residuals], x_index=idx,
        x_sample=[[float(v) for v in res.x[i]] for i in idx], seconds=time.time() - t0))

# ---------------------------------------------------------------------------------------------------------------------
# BASELINE cfg 4: cg! on get_div_grad(1024,1024,1024), b = ones, the config that is row-partitioned
# over 8 GPUs.  The CSR arrays alone would be 94 GB, so the oracle's cg! (ko_cg, unchanged) runs on the MATRIX-FREE
# operator ko_stencil7_matvec (oracle/krylov_oracle.c; tests/test_oracle.py pins it bit for bit to the CSR operator at
# 32^3 / 64^3 and on non-cubic grids).  Leg 40: the first 100 iterations (atol = rtol = 0), x samples included.
# 5 vectors = 43 GB of host memory, ~10 s per iteration on 8 cores (the dots are serial extended-precision sums).
# ---------------------------------------------------------------------------------------------------------------------
if 40 in which:
    t0 = time.time()
    n1 = int(os.environ.get("CFG4_N1", "1024"))
    iters = int(os.environ.get("CFG4_ITERS", "100"))
    n = n1 ** 3
    idx = sample_idx(n)

    def tick(k):
        if k % 5 == 0:
            print(f"  cfg 4: iteration {k}  ({time.time() - t0:.0f} s)", flush=True)
    # "41": only add the second history (the same recurrence with Dot2 dots) to the existing file
    gname = "oracle_cfg4_cg1024.json" if n1 == 1024 else f"oracle_cfg4_cg{n1}.json"
    if 41 in which:
        d = json.load(open(os.path.join(HERE, gname)))
        ok.lib().ko_set_dot_mode(1)
        res2 = ok.cg_stencil7(n1, x_index=idx, progress=tick, atol=0.0, rtol=0.0, itmax=iters, history=True)
        ok.lib().ko_set_dot_mode(0)
        assert res2.rc == 0 and res2.niter == iters
        h1, h2 = np.array(d["residuals"]), res2.residuals
        d.update(residuals_exact_dots=[float(v) for v in h2], x_sample_exact_dots=[float(v) for v in res2.x],
                 oracle_vs_exact_dots_max_rel_dev=float(np.max(np.abs(h1 - h2) / h2)),
                 exact_dots="the same ko_cg with ko_set_dot_mode(1): every dot is Dot2 (double-double accumulation, krylov_oracle.c ko_dot2) "
                            "instead of the documented sequential extended-precision sum; measures the documented oracle's own rounding at this size")
        dump(gname, d)
        sys.exit(0)
    res = ok.cg_stencil7(n1, x_index=idx, progress=tick, atol=0.0, rtol=0.0, itmax=iters, history=True)
    assert res.rc == 0 and res.niter == iters, (res.rc, res.niter, res.status)
    dump("oracle_cfg4_cg1024.json" if n1 == 1024 else f"oracle_cfg4_cg{n1}.json", dict(
        generator="tests/golden/make_scale_golden.py 40",
        oracle="oracle/krylov_oracle.c ko_cg_stencil7 = ko_cg (src/cg.jl:120-291) on the matrix-free get_div_grad "
               "(test/get_div_grad.jl:8-25), bit-identical to the CSR operator (tests/test_oracle.py)",
        config=f"BASELINE cfg 4: cg! on get_div_grad({n1},{n1},{n1}), b = ones, x0 = 0, atol = rtol = 0",
        n=n, nnz=7 * n - 6 * n1 * n1, niter=res.niter, status=res.status,
        residuals=[float(v) for v in res.residuals], x_index=idx, x_sample=[float(v) for v in res.x],
        seconds=time.time() - t0))

# ---------------------------------------------------------------------------------------------------------------------
# Leg 22: cfg 2 (512^3) once more with EXACT DOTS (ko_set_dot_mode(1): Dot2, krylov_oracle.c) -- the full solve of the benchmark
# definition (atol = 0, rtol = 1e-8) and the 100-iteration prefix (atol = rtol = 0).  Everything else of ko_cg is unchanged
# (serial-order products, fma axpys).  Measures how much of the HIP path's distance to the documented oracle is the documented
# oracle's own extended-precision summation (tests/test_gpu_scale_parity.py).  ~10 minutes on 8 cores.
# ---------------------------------------------------------------------------------------------------------------------
if 22 in which:
    t0 = time.time()
    ok.lib().ko_set_dot_mode(1)
    try:
        A = ok.poisson3d(512)
        b = np.ones(A.n)
        idx = sample_idx(A.n)
        pre = ok.cg(A, b, atol=0.0, rtol=0.0, itmax=100, history=True)
        full = ok.cg(A, b, atol=0.0, rtol=FULL_RTOL, itmax=A.n, history=True)
    finally:
        ok.lib().ko_set_dot_mode(0)
    dump("oracle_cfg2_cg512_exact_dots.json", dict(
        generator="tests/golden/make_scale_golden.py 22",
        oracle="oracle/krylov_oracle.c ko_cg (src/cg.jl:120-291) with ko_set_dot_mode(1): every dot is Dot2 (double-double accumulation)",
        config="BASELINE cfg 2: cg! on get_div_grad(512,512,512), b = ones; prefix: atol = rtol = 0, 100 iterations; full: atol = 0, rtol = 1e-8, itmax = n",
        n=A.n, nnz=A.nnz, prefix_residuals=[float(v) for v in pre.residuals], prefix_x_sample=[float(pre.x[i]) for i in idx],
        niter=full.niter, solved=bool(full.solved), status=full.status, residuals=[float(v) for v in full.residuals],
        x_index=idx, x_sample=[float(full.x[i]) for i in idx], seconds=time.time() - t0))

# ---------------------------------------------------------------------------------------------------------------------
# Legs 23 / 24: cfg 3 (gmres!(30, restart) on kron_unsymmetric(256): 45-iteration prefix and the full solve to rtol 1e-8) and
# bicgstab! on the same operator (25 iterations) with EXACT DOTS on the oracle's side (ko_set_dot_mode(1)); everything else of
# ko_gmres / ko_bicgstab unchanged.  ~15 minutes on 8 cores.
# ---------------------------------------------------------------------------------------------------------------------
if 23 in which or 24 in which:
    t0 = time.time()
    A = ok.kron_unsymmetric(256)
    b = A.matvec(np.ones(A.n))
    idx = sample_idx(A.n)
    ok.lib().ko_set_dot_mode(1)
    try:
        if 24 in which:
            r = ok.bicgstab(A, b, atol=0.0, rtol=0.0, itmax=25, history=True)
            dump("oracle_bicgstab256_exact_dots.json", dict(
                generator="tests/golden/make_scale_golden.py 24",
                oracle="oracle/krylov_oracle.c ko_bicgstab (src/bicgstab.jl:125-277) with ko_set_dot_mode(1): Dot2 dots",
                config="bicgstab! on kron_unsymmetric(256), b = A*ones, atol = rtol = 0, 25 iterations",
                n=A.n, nnz=A.nnz, niter=r.niter, status=r.status, residuals=[float(v) for v in r.residuals], x_index=idx,
                x_sample=[float(r.x[i]) for i in idx], seconds=time.time() - t0))
        if 23 in which:
            pre = ok.gmres(A, b, memory=30, restart=True, atol=0.0, rtol=0.0, itmax=45, history=True)
            full = ok.gmres(A, b, memory=30, restart=True, atol=0.0, rtol=FULL_RTOL, itmax=A.n, history=True)
            dump("oracle_cfg3_gmres256_exact_dots.json", dict(
                generator="tests/golden/make_scale_golden.py 23",
                oracle="oracle/krylov_oracle.c ko_gmres (src/gmres.jl:121-384) with ko_set_dot_mode(1): Dot2 dots",
                config="BASELINE cfg 3: gmres!(memory = 30, restart = true) on kron_unsymmetric(256), b = A*ones; prefix: atol = rtol = 0, "
                       "45 iterations; full: atol = 0, rtol = 1e-8, itmax = n",
                n=A.n, nnz=A.nnz, memory=30, prefix_residuals=[float(v) for v in pre.residuals], prefix_x_sample=[float(pre.x[i]) for i in idx],
                niter=full.niter, solved=bool(full.solved), status=full.status, residuals=[float(v) for v in full.residuals],
                x_index=idx, x_sample=[float(full.x[i]) for i in idx], seconds=time.time() - t0))
    finally:
        ok.lib().ko_set_dot_mode(0)

# ---------------------------------------------------------------------------------------------------------------------
# Leg 25: cfg 5 BEYOND THE STENCIL -- block_gmres!(memory = 5, restart = true), p = 16, on the
# "banded + random, fixed seed" operator of tools/bench_irregular.py (10 x 2^20 rows, ~26 entries per row, symmetric, seed 1;
# ko_csr_banded_random == csrc/gen_irregular.cpp entry for entry, tests/test_abi.py), B = A X_true, atol = rtol = 0,
# 20 iterations = four cycles: the non-stencil parity pin at full size, and the evidence that the slow decay of the residual on
# this operator (the HIP path's 400-iteration stall, profiles/r04_bench_irregular.jsonl) is the ALGORITHM's.
# ---------------------------------------------------------------------------------------------------------------------
if 25 in which:
    t0 = time.time()
    n, p, mem, iters = 10 * (1 << 20), 16, 5, 20
    A = ok.banded_random(n, seed=1)
    Xt = cfg5_xtrue(A.n, p)
    B = np.stack([A.matvec(np.ascontiguousarray(Xt[:, j])) for j in range(p)], axis=1)
    res = ok.block_gmres(A, B, memory=mem, restart=True, atol=0.0, rtol=0.0, itmax=iters, history=True)
    assert res.niter == iters, (res.niter, res.status)
    idx = sample_idx(A.n)
    dump("oracle_cfg5_banded_block.json", dict(
        generator="tests/golden/make_scale_golden.py 25",
        oracle="oracle/krylov_oracle.c ko_block_gmres (src/block_gmres.jl:110-358)",
        config="cfg 5 on the non-stencil operator: block_gmres!(memory = 5, restart = true), p = 16, banded + random (n = 10 * 2^20, "
               "half_band 13, links 3, seed 1, symmetric; ko_csr_banded_random), B = A*X_true with X_true[i, j] = cos(j pi (i+1)/n) + 0.1 j, "
               "atol = rtol = 0, 20 iterations",
        n=A.n, nnz=A.nnz, p=p, memory=mem, niter=res.niter, status=res.status,
        residuals=[float(v) for v in res.residuals], x_index=idx,
        x_sample=[[float(v) for v in res.x[i]] for i in idx],
        max_err_after_20=float(np.abs(res.x - Xt).max()), seconds=time.time() - t0))

# ---------------------------------------------------------------------------------------------------------------------
# Leg 26: how sensitive IS the recurrence of leg 25?  The same oracle solve with every entry of B moved by ONE ULP (factor
# 1 +- 2^-52, seeded signs), 8 iterations; the relative change of each residual norm is stored in the golden of leg 25 as
# `one_ulp_sensitivity`.  (Measured: 1.6e-16 at iteration 1, 1.8e-8 at iteration 2, 1e-7 .. 1e-6 up to iteration 7, 2.6e-5 at
# iteration 8 -- block-GMRES with 16 trigonometric right-hand sides on this operator amplifies rounding by 1e8 .. 1e11 within
# one cycle, in the ORACLE.  binary128 is out of reach at 10.5 M rows x 16, so this is the yardstick the GPU test's tolerance
# is derived from.)  3 minutes on 8 cores.
# ---------------------------------------------------------------------------------------------------------------------
if 26 in which:
    t0 = time.time()
    n, p, mem, iters = 10 * (1 << 20), 16, 5, 8
    A = ok.banded_random(n, seed=1)
    Xt = cfg5_xtrue(A.n, p)
    B = np.stack([A.matvec(np.ascontiguousarray(Xt[:, j])) for j in range(p)], axis=1)
    rng = np.random.default_rng(11)
    Bp = B * (1.0 + (rng.integers(0, 2, size=B.shape) * 2 - 1) * 2.0 ** -52)
    res = ok.block_gmres(A, Bp, memory=mem, restart=True, atol=0.0, rtol=0.0, itmax=iters, history=True)
    path = os.path.join(HERE, "oracle_cfg5_banded_block.json")
    g = json.load(open(path))
    h, href = np.array(res.residuals), np.array(g["residuals"][:iters + 1])
    sv = np.linalg.svd(np.linalg.qr(B, mode="r"), compute_uv=False)
    g["one_ulp_sensitivity"] = [float(abs(a - b) / b) for a, b in zip(h, href)]
    g["one_ulp_sensitivity_note"] = ("relative change of the ORACLE's residual norms (iterations 0..8) when every entry of B is moved by one ulp "
                                     "(make_scale_golden.py leg 26); cond_2(B) = %.1f" % float(sv[0] / sv[-1]))
    g["one_ulp_sensitivity_seconds"] = time.time() - t0
    dump("oracle_cfg5_banded_block.json", g)

# ---------------------------------------------------------------------------------------------------------------------
# Legs 27 / 28: the VECTOR solvers on the non-stencil operators at full size (10 x 2^20 rows), documented dots and exact (Dot2) dots:
#   27: cg! on banded + random (symmetric, seed 1), b = A x_true with x_true = cos(1e-3 i) + 0.5 (tools/bench_irregular.py), atol = rtol = 0, 100 iterations;
#       and the full solve to rtol 1e-8 (the oracle needs ~210 iterations)
#   28: gmres!(30, restart) and bicgstab! on the nonsymmetric variant with four rows of 3000 further entries, b = A x_true, 45 / 25 iterations
# These products run through the LDS stream kernel / the strided vector kernel for the dense rows, not the staged stencil kernels.
# ~10 minutes on 8 cores.
# ---------------------------------------------------------------------------------------------------------------------
def _irregular_case(unsym):
    n = 10 * (1 << 20)
    A = ok.banded_random(n, seed=1, unsym=unsym, dense_rows=4 if unsym else 0)
    xt = np.cos(np.arange(n) * 1e-3) + 0.5
    return A, A.matvec(xt)


if 27 in which:
    t0 = time.time()
    A, b = _irregular_case(False)
    idx = sample_idx(A.n)
    out = dict(generator="tests/golden/make_scale_golden.py 27", oracle="oracle/krylov_oracle.c ko_cg (src/cg.jl:120-291); *_exact = with ko_set_dot_mode(1) (Dot2 dots)",
               config="cg! on banded + random (n = 10 * 2^20, half_band 13, links 3, seed 1, symmetric), b = A (cos(1e-3 i) + 0.5); prefix: atol = rtol = 0, "
                      "100 iterations; full: atol = 0, rtol = 1e-8", n=A.n, nnz=A.nnz, x_index=idx)
    for tag, mode in (("", 0), ("_exact", 1)):
        ok.lib().ko_set_dot_mode(mode)
        try:
            pre = ok.cg(A, b, atol=0.0, rtol=0.0, itmax=100, history=True)
            full = ok.cg(A, b, atol=0.0, rtol=FULL_RTOL, itmax=A.n, history=True)
        finally:
            ok.lib().ko_set_dot_mode(0)
        out["prefix_residuals" + tag] = [float(v) for v in pre.residuals]
        out["prefix_x_sample" + tag] = [float(pre.x[i]) for i in idx]
        out["niter" + tag], out["status" + tag] = full.niter, full.status
        out["residuals" + tag] = [float(v) for v in full.residuals]
    out["seconds"] = time.time() - t0
    dump("oracle_irregular_cg.json", out)

if 28 in which:
    t0 = time.time()
    A, b = _irregular_case(True)
    idx = sample_idx(A.n)
    out = dict(generator="tests/golden/make_scale_golden.py 28",
               oracle="oracle/krylov_oracle.c ko_gmres (src/gmres.jl:121-384), ko_bicgstab (src/bicgstab.jl:125-277); *_exact = with ko_set_dot_mode(1) (Dot2 dots)",
               config="banded + random, nonsymmetric, four rows of 3000 further entries (n = 10 * 2^20, seed 1), b = A (cos(1e-3 i) + 0.5), atol = rtol = 0: "
                      "gmres!(memory = 30, restart = true) 45 iterations; bicgstab! 25 iterations", n=A.n, nnz=A.nnz, x_index=idx)
    for tag, mode in (("", 0), ("_exact", 1)):
        ok.lib().ko_set_dot_mode(mode)
        try:
            g = ok.gmres(A, b, memory=30, restart=True, atol=0.0, rtol=0.0, itmax=45, history=True)
            bi = ok.bicgstab(A, b, atol=0.0, rtol=0.0, itmax=25, history=True)
        finally:
            ok.lib().ko_set_dot_mode(0)
        out["gmres_residuals" + tag] = [float(v) for v in g.residuals]
        out["gmres_x_sample" + tag] = [float(g.x[i]) for i in idx]
        out["bicgstab_residuals" + tag] = [float(v) for v in bi.residuals]
        out["bicgstab_x_sample" + tag] = [float(bi.x[i]) for i in idx]
        out["bicgstab_niter" + tag], out["bicgstab_status" + tag] = bi.niter, bi.status
    out["seconds"] = time.time() - t0
    dump("oracle_irregular_gmres_bicgstab.json", out)
