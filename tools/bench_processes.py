"""ms per step of the device Krylov processes on get_div_grad(n1^3).  Usage: python tools/bench_processes.py [n1] [k]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, krylov_jl_amd as K

n1 = int(sys.argv[1]) if len(sys.argv) > 1 else 256
k = int(sys.argv[2]) if len(sys.argv) > 2 else 30
ctx = K.Context(0)
A = K.CsrMatrix.stencil(ctx, "poisson", n1)
n = A.n
b = ctx.empty(n); K.kfill_(b, 1.0)
At = A.transpose()
V, U = K.DeviceMatrix(ctx, n, k + 1), K.DeviceMatrix(ctx, n, k + 1)      # storage reused: no hipMalloc / hipFree in the timed region
out = {"n1": n1, "n": n, "k": k}
for name, fn in [("hermitian_lanczos", lambda: K.hermitian_lanczos(A, b, k, V=V)),
                 ("hermitian_lanczos_reorth", lambda: K.hermitian_lanczos(A, b, k, reorthogonalization=True, V=V)),
                 ("arnoldi", lambda: K.arnoldi(A, b, k, V=V)),
                 ("arnoldi_reorth", lambda: K.arnoldi(A, b, k, reorthogonalization=True, V=V)),
                 ("golub_kahan", lambda: K.golub_kahan(A, b, k, At=At, V=V, U=U))] * 2:      # second pass is the one kept (clocks ramped)
    fn(); ctx.sync()
    t = time.perf_counter(); fn(); ctx.sync(); dt = time.perf_counter() - t
    out[name + "_ms_per_step"] = 1e3 * dt / k
print(json.dumps(out))
