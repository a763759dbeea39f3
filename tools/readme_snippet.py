import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, krylov_jl_amd as K
ctx = K.Context(0)
A = K.CsrMatrix.stencil(ctx, "poisson", 48)
b = ctx.array(np.ones(A.n))
x, stats, ws = K.cg(A, b, rtol=1e-8, history=True); print(stats.niter, stats.status)
x, stats, ws = K.cg(A, b, M=K.Ilu0(A)); print(stats.niter)
x, stats, ws = K.gmres(A, b, memory=30, restart=True); print(stats.niter, stats.solved)
V, beta, H = K.arnoldi(A, b, 20, reorthogonalization=True); print(V.shape, H.shape, beta)
