#!/usr/bin/env python3
"""The sliced form of the coded SpMV (ctx option spmv_sell = 1, csrc/colcode.hip csr_build_sell + spmv_sell_kernel) against the coded
kernel: y and the fused dots bit for bit on several operators, then the times of the plain and the fused product at n1^3 (default 512)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
ctx.set_option("spmv_codes", 2)
for kind, n1 in (("poisson", 40), ("poisson", 67), ("kron_unsymmetric", 33), ("stencil27", 30)):
    A = K.CsrMatrix.stencil(ctx, kind, n1)
    n = A.n
    x = ctx.array(np.cos(np.arange(n) * 0.37) + 0.1)
    out = {}
    for sell in (0, 1):
        ctx.set_option("spmv_sell", sell)
        y = ctx.empty(n)
        A.matvec(x, y)
        out[sell] = (y.to_host(), K.spmv_dot(A, x, ctx.empty(n)), A.spmv_bytes_stored)
    ok = bool(np.array_equal(out[0][0], out[1][0])) and out[0][1] == out[1][1]
    print(json.dumps({"operator": kind, "n1": n1, "y_and_dot_bit_identical": ok, "bytes_coded": out[0][2], "bytes_sliced": out[1][2]}), flush=True)
    assert ok
    del A
# the int32 column stream (spmv_codes = 0): the sliced form against the staged CSR kernel
ctx.set_option("spmv_codes", 0)
for kind, n1 in (("poisson", 40), ("poisson", 67), ("kron_unsymmetric", 33)):
    A = K.CsrMatrix.stencil(ctx, kind, n1)
    n = A.n
    x = ctx.array(np.cos(np.arange(n) * 0.37) + 0.1)
    out = {}
    for sell in (0, 3):
        ctx.set_option("spmv_sell", sell)
        y = ctx.empty(n)
        A.matvec(x, y)
        out[sell] = (y.to_host(), K.spmv_dot(A, x, ctx.empty(n)), A.spmv_bytes_stored, A.sell32_info)
    ok = bool(np.array_equal(out[0][0], out[3][0])) and out[0][1] == out[3][1]
    print(json.dumps({"operator": kind, "n1": n1, "int32_columns": True, "y_and_dot_bit_identical": ok, "bytes_csr": out[0][2], "bytes_sliced": out[3][2], "sell32_info": out[3][3]}), flush=True)
    assert ok and out[3][3][0] == 1
    del A
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx.set_option("spmv_codes", 1)
A = K.CsrMatrix.stencil(ctx, "poisson", n1)
n = A.n
x = ctx.empty(n); K.kfill_(x, 1.0)
y = ctx.empty(n)
for codes in (1, 0):
  ctx.set_option("spmv_codes", codes)
  for rnd in range(2):
    for sell in (0, 1, 2):
          ctx.set_option("spmv_sell", sell)
          A.matvec(x, y); K.spmv_dot(A, x, y); ctx.sync()
          t0 = time.perf_counter()
          for _ in range(20): A.matvec(x, y)
          ctx.sync(); t_plain = (time.perf_counter() - t0) / 20
          t0 = time.perf_counter()
          for _ in range(20): K.spmv_dot(A, x, y)
          ctx.sync(); t_dot = (time.perf_counter() - t0) / 20
          print(json.dumps({"n1": n1, "spmv_codes": codes, "spmv_sell": sell, "plain_ms": round(1e3 * t_plain, 4), "fused_dot_ms_incl_host_sync": round(1e3 * t_dot, 4),
                            "bytes_stored": A.spmv_bytes_stored, "frac_algorithmic_plain": round(A.spmv_bytes / t_plain / 8e12, 4)}), flush=True)
ctx.close()
