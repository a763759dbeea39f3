"""One rank of tests/test_gpu_rccl_multi.py: a process with its own GPU and a REAL RCCL communicator.

usage: python tests/rccl_worker.py <rank> <world> <uid_file> <out_dir>
Rank 0 creates the ncclUniqueId and publishes it through <uid_file>; every rank runs the distributed SpMV, cg!, gmres!,
bicgstab! and block_gmres! on its row slab in BOTH halo modes and writes what it got to <out_dir>/rank<r>.npz; the parent
compares with the CPU oracle's solve of the global system."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import krylov_jl_amd as K  # noqa: E402

rank, world, uid_file, out_dir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
ctx = K.Context(rank % max(K.device_count(), 1))
if rank == 0:
    uid = K.Context.comm_unique_id()
    with open(uid_file + ".tmp", "wb") as f:
        f.write(uid)
    os.replace(uid_file + ".tmp", uid_file)
else:
    t0 = time.time()
    while not os.path.exists(uid_file):
        if time.time() - t0 > 120:
            raise SystemExit("no unique id from rank 0")
        time.sleep(0.05)
    uid = open(uid_file, "rb").read()
ctx.comm_init(rank, world, uid)
info = ctx.comm_info()
out = {"rccl_ranks": info["rccl_ranks"], "halo_comm_separate": info["halo_comm_separate"]}

N1_CG, N1_NS, P = 24, 14, 4
n = N1_CG ** 3
starts = K.row_partition(n, world)
r0, r1 = starts[rank], starts[rank + 1]
x = np.linspace(-1, 1, n) ** 3 + 0.25
for mode in (1, 2):
    ctx.set_option("halo_mode", mode)
    A = K.CsrMatrix.stencil(ctx, "poisson", N1_CG, rows=(r0, r1), distributed=True)
    out[f"gather{mode}"] = A.halo_info[0]
    for overlap in (1, 0):
        ctx.set_option("overlap_halo", overlap)
        out[f"y{mode}{overlap}"] = A.matvec(ctx.array(x[r0:r1])).to_host()
    ctx.set_option("overlap_halo", 1)
    b = ctx.empty(r1 - r0)
    K.kfill_(b, 1.0)
    for fused in (2, 1, 0):
        xs, st, _ = K.cg(A, b, history=True, fused=fused)
        out[f"cg{mode}{fused}_hist"] = st.residuals.copy()
        out[f"cg{mode}{fused}_x"] = xs.to_host()
    xs, st, _ = K.cg(A, b, history=True, variant=1)
    out[f"cgv{mode}_hist"] = st.residuals.copy()
    # a time limit every rank must honour together (the clock is the one rank-local stopping test)
    xs, st, _ = K.cg(A, b, atol=0.0, rtol=0.0, itmax=10 ** 6, timemax=1e-5)
    out[f"timed{mode}"] = np.array([st.niter, int(st.status == "time limit exceeded")])
    del A

n2 = N1_NS ** 3
starts2 = K.row_partition(n2, world)
q0, q1 = starts2[rank], starts2[rank + 1]
ones = np.ones(n2)
for mode in (1, 2):
    ctx.set_option("halo_mode", mode)
    A = K.CsrMatrix.stencil(ctx, "kron_unsymmetric", N1_NS, rows=(q0, q1), distributed=True)
    t = ctx.empty(q1 - q0)
    A.matvec(ctx.array(ones[q0:q1]), t)
    bh = t.to_host()                                            # slab of b = A * ones
    out[f"b{mode}"] = bh
    _, stg, _ = K.gmres(A, ctx.array(bh), memory=10, restart=True, history=True)
    out[f"gmres{mode}"] = stg.residuals.copy()
    _, stb, _ = K.bicgstab(A, ctx.array(bh), history=True)
    out[f"bicgstab{mode}"] = stb.residuals.copy()
    tt = (np.arange(n2) + 1.0) / n2
    Xt = np.stack([tt ** j for j in range(P)], axis=1)
    Y = K.Panel(ctx, q1 - q0, P)
    K.spmm_(A, K.Panel.from_host(ctx, Xt[q0:q1]), Y)
    Bloc = Y.to_host()
    out[f"B{mode}"] = Bloc
    X, stk, _ = K.block_gmres(A, Bloc, memory=8, history=True, ctx=ctx)
    out[f"block{mode}"] = stk.residuals.copy()
    out[f"blockX{mode}"] = X
    # A' of the partitioned operator (one all-to-all of the entries over RCCL), applied to the slab of `ones`-like data
    At = A.transpose()
    xt = np.cos(np.arange(n2) * 0.01)
    out[f"At{mode}"] = At.matvec(ctx.array(xt[q0:q1])).to_host()
    del A, At
ctx.barrier()
np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **out)
ctx.close()
print(f"rank {rank} of {world} done", flush=True)
