#!/usr/bin/env python3
"""Can two RCCL ranks share the single GPU of a gpurun box?  (Would let the exchange path be tested.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
import numpy as np
import krylov_jl_amd as K
ctx = K.Context(0)
uid = [K.Context.comm_unique_id() if rank == 0 else None]
dist.broadcast_object_list(uid, src=0)
try:
    ctx.comm_init(rank, world, uid[0])
    n1 = 16; n = n1 ** 3
    starts = K.row_partition(n, world)
    A = K.CsrMatrix.stencil(ctx, "poisson", n1, rows=(starts[rank], starts[rank + 1]), distributed=True)
    b = ctx.empty(starts[rank + 1] - starts[rank]); K.kfill_(b, 1.0)
    x, st, _ = K.cg(A, b, history=True)
    print(f"rank {rank}: cg niter {st.niter} last {st.residuals[-1]:.6e}", flush=True)
except Exception as e:
    print(f"rank {rank}: FAILED {e}", flush=True)
dist.barrier(); dist.destroy_process_group()
