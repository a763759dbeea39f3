"""Generates tests/golden/oracle_histories.json from the CPU oracle (oracle/krylov_oracle.c).

These are golden vectors OF THE ORACLE (iteration counts, status strings, residual histories): the
reference itself cannot run here (no Julia) and ships no residual histories, so they are "parity
unpinned" with respect to Krylov.jl beyond what tests/test_oracle.py pins.
Run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import oracle as ok  # noqa: E402

CASES = [
    dict(name="cg_poisson16_default", solver="cg", matrix="poisson3d", n1=16, rhs="ones", kwargs={}),
    dict(name="cg_poisson32_benchmark", solver="cg", matrix="poisson3d", n1=32, rhs="ones",
         kwargs=dict(atol=0.0, rtol=1e-8, itmax=32 ** 3)),
    dict(name="cg_poisson64_default", solver="cg", matrix="poisson3d", n1=64, rhs="ones", kwargs={}),
    dict(name="gmres_kron12_restart10", solver="gmres", matrix="kron_unsymmetric", n1=12, rhs="A*ones",
         kwargs=dict(memory=10, restart=True)),
    dict(name="gmres_kron12_reorth", solver="gmres", matrix="kron_unsymmetric", n1=12, rhs="A*ones",
         kwargs=dict(memory=10, restart=True, reorthogonalization=True)),
    dict(name="bicgstab_kron12", solver="bicgstab", matrix="kron_unsymmetric", n1=12, rhs="A*ones", kwargs={}),
]

out = {"generator": "tests/golden/make_golden.py", "oracle": "oracle/krylov_oracle.c", "cases": []}
for c in CASES:
    A = getattr(ok, c["matrix"])(c["n1"])
    b = np.ones(A.n) if c["rhs"] == "ones" else A.matvec(np.ones(A.n))
    res = getattr(ok, c["solver"])(A, b, history=True, **c["kwargs"])
    d = dict(c)
    d.update(niter=res.niter, status=res.status, solved=res.solved,
             residuals=[float(v) for v in res.residuals])
    out["cases"].append(d)
    print(c["name"], res.niter, res.status)
json.dump(out, open(os.path.join(HERE, "oracle_histories.json"), "w"), indent=1)
