"""Residual histories of the oracle's recurrences in IEEE binary128 -> tests/golden/quad_histories.json.

oracle/quad_reference.c compiles the oracle's own source with __float128 in place of double (make -C oracle quadref);
this script feeds it the SAME double-precision inputs the parity tests use and stores the histories.  They are the
yardstick of VERDICT r01 item 6: the distance of a double-precision run (the oracle's or the HIP path's) to these
histories is that run's own accumulated rounding; the tests bound the HIP path's distance and record the oracle's.
Golden vectors of the ORACLE'S ALGORITHM (parity unpinned with respect to Krylov.jl, like every history here).

Run:  make -C oracle quadref && python tests/golden/make_quad_golden.py
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
NAN = float("nan")


def block_rhs(A, p):
    """tests/test_gpu_block.py::_rhs"""
    S = A.to_scipy()
    t = (np.arange(A.n) + 1.0) / A.n
    if p <= 4:
        Xt = np.stack([t ** j for j in range(p)], axis=1)
    else:
        Xt = np.stack([np.cos(j * np.pi * t) + 0.1 * j for j in range(p)], axis=1)
    return S @ Xt


CASES = [
    dict(name="cg_poisson16", solver="cg", matrix="poisson3d", n1=16, rhs="ones"),
    dict(name="cg_poisson32", solver="cg", matrix="poisson3d", n1=32, rhs="ones"),
    dict(name="bicgstab_kron8", solver="bicgstab", matrix="kron_unsymmetric", n1=8, rhs="A*ones"),
    dict(name="bicgstab_kron12", solver="bicgstab", matrix="kron_unsymmetric", n1=12, rhs="A*ones"),
    dict(name="bicgstab_kron16", solver="bicgstab", matrix="kron_unsymmetric", n1=16, rhs="A*ones"),
    dict(name="gmres_kron8", solver="gmres", matrix="kron_unsymmetric", n1=8, rhs="A*ones", memory=10),
    dict(name="gmres_kron16", solver="gmres", matrix="kron_unsymmetric", n1=16, rhs="A*ones", memory=10),
    dict(name="gmres_kron16_restart", solver="gmres", matrix="kron_unsymmetric", n1=16, rhs="A*ones", memory=10, restart=True),
    dict(name="gmres_kron16_restart_reorth", solver="gmres", matrix="kron_unsymmetric", n1=16, rhs="A*ones", memory=10,
         restart=True, reorthogonalization=True),
    dict(name="gmres_kron20_mem30", solver="gmres", matrix="kron_unsymmetric", n1=20, rhs="A*ones", memory=30, restart=True,
         atol=1e-10, rtol=1e-10),
    dict(name="block_kron8_p4", solver="block_gmres", matrix="kron_unsymmetric", n1=8, p=4, rhs="block", memory=8),
    dict(name="block_kron8_p4_restart", solver="block_gmres", matrix="kron_unsymmetric", n1=8, p=4, rhs="block", memory=8, restart=True),
    dict(name="block_kron10_p16_restart", solver="block_gmres", matrix="kron_unsymmetric", n1=10, p=16, rhs="block", memory=8,
         restart=True),
    dict(name="block_kron12_p16", solver="block_gmres", matrix="kron_unsymmetric", n1=12, p=16, rhs="block", memory=8),
]


def run_case(c):
    A = getattr(ok, c["matrix"])(c["n1"])
    p = c.get("p", 1)
    if c["rhs"] == "ones":
        b = np.ones(A.n)
    elif c["rhs"] == "A*ones":
        b = A.matvec(np.ones(A.n))
    else:
        b = block_rhs(A, p)
    kw = dict(restart=bool(c.get("restart", False)), reorthogonalization=bool(c.get("reorthogonalization", False)))
    if "atol" in c:
        kw.update(atol=c["atol"], rtol=c["rtol"])
    # the double-precision oracle on the same inputs
    if c["solver"] == "cg":
        ref = ok.cg(A, b, history=True)
    elif c["solver"] == "bicgstab":
        ref = ok.bicgstab(A, b, history=True)
    elif c["solver"] == "gmres":
        ref = ok.gmres(A, b, memory=c["memory"], history=True, **kw)
    else:
        ref = ok.block_gmres(A, b, memory=c["memory"], history=True, **kw)
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
        f.write(struct.pack("8i", SOLVER[c["solver"]], KIND[c["matrix"]], c["n1"], p, c.get("memory", 0),
                            int(kw["restart"]), int(kw["reorthogonalization"]), 0))
        f.write(struct.pack("2d", c.get("atol", NAN), c.get("rtol", NAN)))
        f.write(np.asfortranarray(b, dtype=np.float64).tobytes(order="F"))
        path = f.name
    out = json.loads(subprocess.check_output([BIN, path]))
    os.unlink(path)
    hq, hd = np.array(out["residuals"]), ref.residuals
    k = min(len(hq), len(hd))
    dev = float(np.max(np.abs(hd[:k] - hq[:k]) / hq[:k]))
    print(f"{c['name']:32s} niter quad {out['niter']:4d} double {ref.niter:4d}   oracle(double) vs quad: max rel dev {dev:.2e}", flush=True)
    d = dict(c)
    d.update(niter=out["niter"], status=out["status"], residuals=out["residuals"], oracle_double_niter=ref.niter,
             oracle_double_max_rel_dev=dev)
    return d


if __name__ == "__main__":
    res = {"generator": "tests/golden/make_quad_golden.py", "reference": "oracle/quad_reference.c (__float128 build of oracle/krylov_oracle.c)",
           "cases": [run_case(c) for c in CASES]}
    json.dump(res, open(os.path.join(HERE, "quad_histories.json"), "w"), indent=1)
