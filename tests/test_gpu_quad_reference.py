"""Whose rounding is it?  

tests/golden/quad_histories.json holds the residual histories of the ORACLE'S OWN SOURCE compiled in IEEE binary128
(oracle/quad_reference.c, tests/golden/make_quad_golden.py): the recurrences of src/cg.jl, src/gmres.jl, src/bicgstab.jl
and src/block_gmres.jl evaluated essentially exactly on the double-precision inputs.  The distance of a double-precision
run to that history is that run's own accumulated rounding.  Measured for the CPU oracle (stored in the file):
CG 1e-15, GMRES without restart 1e-13, but BiCGSTAB 16^3 1.5e-8, restarted GMRES 2.4e-9, restarted block-GMRES p = 16
1.1e-3 -- the recurrences amplify eps that much (ratios of cancelling dots; b - A x recomputed at a restart when
r_k ~ 1e-8 r_0; cond of the residual block's R factor).  Two double-precision implementations therefore cannot agree
better than that, however each is written, and a GPU-vs-oracle tolerance below it would test luck.

Stated tolerance, from the analysis and not from the measured gap: the HIP path's distance to the binary128 history is at
most FACTOR = 8 times the CPU oracle's own distance (plus 1e-13), with equal iteration counts and status.  The measured
ratios are logged (gpurun_out/parity_log.jsonl) and quoted in DESIGN.md section 3.2.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FACTOR = 8.0
FLOOR = 1e-13

with open(os.path.join(ROOT, "tests", "golden", "quad_histories.json")) as _f:
    CASES = json.load(_f)["cases"]


def _block_rhs(A, p):
    S = A.to_scipy()
    t = (np.arange(A.n) + 1.0) / A.n
    if p <= 4:
        Xt = np.stack([t ** j for j in range(p)], axis=1)
    else:
        Xt = np.stack([np.cos(j * np.pi * t) + 0.1 * j for j in range(p)], axis=1)
    return S @ Xt


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_hip_path_is_as_close_to_the_exact_recurrence_as_the_cpu_oracle(K, ctx, oracle, parity_log, case):
    c = case
    A = getattr(oracle, c["matrix"])(c["n1"])
    dA = K.CsrMatrix.from_host(ctx, A.rowptr, A.col, A.val, (A.n, A.n))
    kw = dict(restart=bool(c.get("restart", False)), reorthogonalization=bool(c.get("reorthogonalization", False)))
    if "atol" in c:
        kw.update(atol=c["atol"], rtol=c["rtol"])
    if c["rhs"] == "block":
        B = _block_rhs(A, c["p"])
        _, st, _ = K.block_gmres(dA, B, memory=c["memory"], history=True, **kw)
    else:
        bh = np.ones(A.n) if c["rhs"] == "ones" else A.matvec(np.ones(A.n))
        b = ctx.array(bh)
        if c["solver"] == "cg":
            _, st, _ = K.cg(dA, b, history=True)
        elif c["solver"] == "bicgstab":
            _, st, _ = K.bicgstab(dA, b, history=True)
        else:
            _, st, _ = K.gmres(dA, b, memory=c["memory"], history=True, **kw)
    hq = np.array(c["residuals"])
    assert st.niter == c["niter"] and st.status == c["status"] and len(st.residuals) == len(hq)
    d_gpu = float(np.max(np.abs(st.residuals - hq) / hq))
    d_cpu = float(c["oracle_double_max_rel_dev"])
    parity_log(test="vs_binary128", case=c["name"], niter=st.niter, gpu_vs_quad=d_gpu, cpu_oracle_vs_quad=d_cpu,
               ratio=d_gpu / max(d_cpu, 1e-300))
    assert d_gpu <= FACTOR * d_cpu + FLOOR, (c["name"], d_gpu, d_cpu)
