#!/usr/bin/env python3
"""Phases of one block of the ILU(0) block schedule (lower solve), from shader-clock stamps.  Needs the instrumented library:
   KHIP_OUT=$PWD/tools/_trace/libkrylov_hip_ilutrace.so KHIP_BUILD_DIR=/tmp/khip_ilutrace KHIP_EXTRA_FLAGS=-DKHIP_ILU_TRACE bash krylov.jl_amd/build.sh
   KHIP_LIBRARY=$PWD/tools/_trace/libkrylov_hip_ilutrace.so python tools/archive/ilu_trace.py [n1]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = n1 ** 3
A = K.CsrMatrix.stencil(ctx, sys.argv[2] if len(sys.argv) > 2 else "poisson", n1)
P = K.Ilu0(A)
x, y = ctx.empty(n), ctx.empty(n); K.kfill_(x, 1.0)
for _ in range(3): P(x, y)
ctx.sync()
buf = (C.c_ulonglong * 512)()
assert K.lib().khip_debug_ilu_trace(buf) == 0
t = np.array(list(buf), dtype=np.int64).reshape(64, 8)[:, :6]
d = np.diff(t, axis=1)
names = ["stage (entries, rhs -> LDS)", "wait for the flags", "fetch the faces' y", "levels out of LDS", "write y through"]
print(f"{n1}^3, blocks {P.block_info()[1]}: median shader-clock ticks per phase over 64 blocks in the middle of the schedule")
for k, nm in enumerate(names):
    print(f"  {nm:30s} {int(np.median(d[:, k])):8d}   (min {int(d[:, k].min())}, max {int(d[:, k].max())})")
print(f"  {'whole block':30s} {int(np.median(t[:, 5] - t[:, 0])):8d}")
ctx.close()
