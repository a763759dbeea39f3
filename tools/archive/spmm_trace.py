#!/usr/bin/env python3
"""Where does a wave of spmm_window_kernel spend its iteration?  Needs the instrumented variant of the library:
   KHIP_OUT=$PWD/gpurun_out/libkrylov_hip_trace.so KHIP_BUILD_DIR=/tmp/khip_trace KHIP_EXTRA_FLAGS=-DKHIP_WIN_TRACE bash krylov.jl_amd/build.sh
   KHIP_LIBRARY=$PWD/gpurun_out/libkrylov_hip_trace.so python tools/archive/spmm_trace.py
Prints, per traced iteration of wave 0 of one workgroup, the shader-clock ticks between the phase stamps:
   0 iteration start | 1 window written to LDS | 2 barrier passed | 3 prefetch loads issued | 4 products done | 5 y stored | 6 barrier passed"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 216
p = 16
n = n1 ** 3
A = K.CsrMatrix.stencil(ctx, "stencil27", n1)
X, Y = K.Panel(ctx, n, p), K.Panel(ctx, n, p)
K.kfill_(X.buf, 1.0)
for _ in range(3):
    K.spmm_(A, X, Y)
ctx.sync()
buf = (C.c_ulonglong * 128)()
assert K.lib().khip_debug_win_trace(buf) == 0
t = np.array(list(buf), dtype=np.int64).reshape(16, 8)
names = ["lds_write", "barrier1", "issue_prefetch", "products", "y_store", "barrier2", "next_start"]
print("ticks per phase (s_memtime), iterations 8..23 of wave 0 of workgroup 5:")
print("  " + "  ".join(f"{nm:>14s}" for nm in names) + "           total")
for i in range(15):
    row = t[i]
    d = [row[1] - row[0], row[2] - row[1], row[3] - row[2], row[4] - row[3], row[5] - row[4], row[6] - row[5], t[i + 1][0] - row[6]]
    print("  " + "  ".join(f"{int(v):14d}" for v in d) + f"   {int(t[i + 1][0] - row[0]):10d}")
ctx.close()
