"""How far the ORACLE'S OWN double-precision history drifts from the exact recurrence over a full solve of the benchmark
definition (atol = 0, rtol = 1e-8, itmax = n; benchmark/benchmarks.jl:14-21) -> tests/golden/full_solve_tolerance.json.

The full-convergence parity tests at the BASELINE sizes (tests/test_gpu_scale_parity.py) cannot hold the
histories to 1e-12 over ~1000 iterations: two correct double-precision implementations of cg! / gmres!(restart) drift apart
as fast as each of them drifts from exact arithmetic.  This script measures that drift where binary128 is affordable
(oracle/quad_reference.c: the oracle's source compiled with __float128) on a ladder of sizes of the SAME operators and
settings, so that the tolerance of the full-size tests is derived, not guessed:
  d(size) = max over the history of |oracle_double - binary128| / binary128, together with the iteration counts.

Run:  make -C oracle quadref && python tests/golden/make_full_tolerance.py        (~10 minutes)
"""
import json
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as ok  # noqa: E402

BIN = os.path.join(ROOT, "oracle", "_ref", "quad_reference")
SOLVER = {"cg": 0, "gmres": 1, "bicgstab": 2, "block_gmres": 3}
KIND = {"poisson3d": 0, "kron_unsymmetric": 1, "stencil27_unsym": 2}
RTOL = 1.0e-8


def quad(solver, matrix, n1, p, memory, restart, b):
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
        f.write(struct.pack("8i", SOLVER[solver], KIND[matrix], n1, p, memory, int(restart), 0, 0))
        f.write(struct.pack("2d", 0.0, RTOL))
        f.write(np.asfortranarray(b, dtype=np.float64).tobytes(order="F"))
        path = f.name
    out = json.loads(subprocess.check_output([BIN, path]))
    os.unlink(path)
    return out


def entry(solver, matrix, n1, ref, out):
    hq, hd = np.array(out["residuals"]), np.asarray(ref.residuals)
    k = min(len(hq), len(hd))
    devs = np.abs(hd[:k] - hq[:k]) / hq[:k]
    e = dict(solver=solver, matrix=matrix, n1=n1, niter_quad=out["niter"], niter_double=int(ref.niter), max_rel_dev=float(devs.max()),
             dev_at_quarters=[float(devs[: max(1, (k * q) // 4)].max()) for q in (1, 2, 3, 4)])
    print(e, flush=True)
    return e


res = []
for n1 in (16, 32, 48, 64):
    A = ok.poisson3d(n1)
    b = np.ones(A.n)
    res.append(entry("cg", "poisson3d", n1, ok.cg(A, b, atol=0.0, rtol=RTOL, itmax=A.n, history=True), quad("cg", "poisson3d", n1, 1, 0, False, b)))
for n1 in (12, 16, 24, 32):
    A = ok.kron_unsymmetric(n1)
    b = A.matvec(np.ones(A.n))
    res.append(entry("gmres(30, restart)", "kron_unsymmetric", n1,
                     ok.gmres(A, b, memory=30, restart=True, atol=0.0, rtol=RTOL, itmax=A.n, history=True),
                     quad("gmres", "kron_unsymmetric", n1, 1, 30, True, b)))
for n1 in (10, 14, 18):
    A = ok.stencil27_unsym(n1)
    t = (np.arange(A.n) + 1.0) / A.n
    Xt = np.stack([np.cos(j * np.pi * t) + 0.1 * j for j in range(16)], axis=1)
    B = np.stack([A.matvec(np.ascontiguousarray(Xt[:, j])) for j in range(16)], axis=1)
    res.append(entry("block_gmres(5, restart), p = 16", "stencil27_unsym", n1,
                     ok.block_gmres(A, B, memory=5, restart=True, atol=0.0, rtol=RTOL, itmax=A.n, history=True),
                     quad("block_gmres", "stencil27_unsym", n1, 16, 5, True, B)))
json.dump(dict(generator="tests/golden/make_full_tolerance.py", reference="oracle/quad_reference.c (__float128 build of oracle/krylov_oracle.c)",
               setting="atol = 0, rtol = 1e-8, itmax = n (benchmark/benchmarks.jl:14-21)", ladder=res),
          open(os.path.join(HERE, "full_solve_tolerance.json"), "w"), indent=1)
