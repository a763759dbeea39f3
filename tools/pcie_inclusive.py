#!/usr/bin/env python3
"""The boundary also accepts HOST arrays (khip_csr_create with on_device = 0, khip_memcpy_h2d / d2h).  What a cg! solve of
cfg 2 costs when operator, right-hand side and solution cross PCIe: upload of the plain CSR arrays (pageable host memory),
the solve to rtol 1e-8, download of x.  The operator's host arrays come from the oracle's generator (used here as a data
source only; nothing is checked against it).  Usage: python tools/pcie_inclusive.py [n1]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import krylov_jl_amd as K
import oracle as ok

n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ok.lib().ko_set_threads(min(64, len(os.sched_getaffinity(0))))
t0 = time.perf_counter()
A = ok.poisson3d(n1)
rowptr, col, val = np.ascontiguousarray(A.rowptr), np.ascontiguousarray(A.col, dtype=np.int32), np.ascontiguousarray(A.val)
b = np.ones(A.n)
t_gen = time.perf_counter() - t0
ctx = K.Context(0)
ctx.sync()
t0 = time.perf_counter()
dA = K.CsrMatrix.from_host(ctx, rowptr, col, val, (A.n, A.n))
db = ctx.array(b)
ctx.sync()
t_up = time.perf_counter() - t0
t0 = time.perf_counter()
x, stats, ws = K.cg(dA, db, rtol=1e-8, atol=0.0, itmax=A.n)
ctx.sync()
t_solve = time.perf_counter() - t0
t0 = time.perf_counter()
hx = x.to_host()
t_down = time.perf_counter() - t0
nbytes = rowptr.nbytes + col.nbytes + val.nbytes + b.nbytes
total = t_up + t_solve + t_down
print(json.dumps(dict(n1=n1, niter=stats.niter, solved=bool(stats.solved), host_generation_s=round(t_gen, 2),
                      upload_s=round(t_up, 3), upload_gb=round(nbytes / 1e9, 2), upload_gbps=round(nbytes / t_up / 1e9, 1),
                      solve_s=round(t_solve, 3), first_solve_includes="column codes built once per handle",
                      download_s=round(t_down, 3), download_gbps=round(hx.nbytes / t_down / 1e9, 1),
                      iters_per_s_device_resident=round(stats.niter / t_solve, 1),
                      iters_per_s_pcie_inclusive=round(stats.niter / total, 1))))
ctx.close()
