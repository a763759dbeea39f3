#!/usr/bin/env python3
"""The boundary also accepts HOST arrays (khip_csr_create with on_device = 0, khip_memcpy_h2d / d2h).  What a cg! solve of
cfg 2 costs when operator, right-hand side and solution cross PCIe: upload of the plain CSR arrays (pageable host memory),
the solve to rtol 1e-8, download of x.  The host arrays are the device generator's output copied back beforehand (untimed).
Usage: python tools/pcie_inclusive.py [n1]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np
import krylov_jl_amd as K

n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = K.Context(0)
t0 = time.perf_counter()
n = n1 ** 3
rp, cl, vl, nnz = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int64()
K._ck(K.lib().khip_gen_stencil(ctx._h, 0, n1, n1, n1, 0, n, C.byref(rp), C.byref(cl), C.byref(vl), C.byref(nnz)))
rowptr, col, val = np.empty(n + 1, dtype=np.int32), np.empty(nnz.value, dtype=np.int32), np.empty(nnz.value, dtype=np.float64)
for host, dev in ((rowptr, rp), (col, cl), (val, vl)):
    K._ck(K.lib().khip_memcpy_d2h(ctx._h, host.ctypes.data, dev, host.nbytes))
    K.lib().khip_free(ctx._h, dev)
b = np.ones(n)
t_gen = time.perf_counter() - t0
ctx.sync()
t0 = time.perf_counter()
dA = K.CsrMatrix.from_host(ctx, rowptr, col, val, (n, n))
db = ctx.array(b)
ctx.sync()
t_up = time.perf_counter() - t0
t0 = time.perf_counter()
x, stats, ws = K.cg(dA, db, rtol=1e-8, atol=0.0, itmax=n)
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
