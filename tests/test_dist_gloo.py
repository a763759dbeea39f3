"""world_size-2 (and 3) `gloo` tests of the N>1 path on CPU: the row partition, the ghost-column
analysis and the halo plan that libkrylov_hip uses for its RCCL exchange (csrc/comm.cpp) are driven
through the same host entry points (khip_ghost_columns_host / khip_halo_plan_host), the exchange is
performed with real point-to-point messages between processes, and a distributed CG (local BLAS-1 in
numpy, all-reduced dots -- the decomposition of docs/src/custom_workspaces.md:477-586) must reproduce
the single-process oracle: same iteration count, residual history within 1e-10.

The arithmetic here is numpy TEST code standing in for the device kernels (no GPU in this
container); what is under test is the distributed plumbing, which is device independent."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n1, kind, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import torch
        import torch.distributed as dist
        import krylov_jl_amd as K
        import oracle as ok
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        A = getattr(ok, kind)(n1)
        n = A.n
        starts = K.row_partition(n, world)
        r0, r1 = starts[rank], starts[rank + 1]
        m = r1 - r0
        sl = A.row_slice(r0, r1)
        ghost = K.ghost_columns_host(sl.rowptr, sl.col, r0)
        all_ghost = [None] * world
        dist.all_gather_object(all_ghost, ghost)
        recv_off, send_off, send_idx = K.halo_plan_host(rank, world, starts, all_ghost)
        # [owned | ghost] column numbering, as csrc/spmv.hip:col_remap_kernel does on the device
        cols = sl.col.astype(np.int64)
        own = (cols >= r0) & (cols < r1)
        loc = np.where(own, cols - r0, m + np.searchsorted(ghost, cols))
        rows = np.repeat(np.arange(m), np.diff(sl.rowptr))
        vals = sl.val.copy()

        def spmv(x_owned):
            gh = np.zeros(len(ghost))
            reqs = []
            for r in range(world):        # grouped send/recv, like ncclGroupStart/End in comm.cpp
                if r == rank:
                    continue
                ns = send_off[r + 1] - send_off[r]
                nr = recv_off[r + 1] - recv_off[r]
                if ns:
                    buf = torch.from_numpy(np.ascontiguousarray(x_owned[send_idx[send_off[r]:send_off[r + 1]]]))
                    reqs.append(dist.isend(buf, r))
                if nr:
                    rb = torch.empty(nr, dtype=torch.float64)
                    reqs.append((dist.irecv(rb, r), rb, r))
            for it in reqs:
                if isinstance(it, tuple):
                    it[0].wait()
                    gh[recv_off[it[2]]:recv_off[it[2] + 1]] = it[1].numpy()
                else:
                    it.wait()
            xe = np.concatenate([x_owned, gh])
            y = np.zeros(m)
            np.add.at(y, rows, vals * xe[loc])
            return y

        def dot(a, b):
            t = torch.tensor([float(a @ b)], dtype=torch.float64)
            dist.all_reduce(t)
            return float(t.item())

        # distributed product equals the global one
        xg = np.linspace(-1, 1, n) ** 3 + 0.1
        y = spmv(xg[r0:r1])
        assert np.allclose(y, A.matvec(xg)[r0:r1], rtol=0, atol=1e-13)
        # distributed CG (src/cg.jl:153-268, M = I) on b = ones
        b = np.ones(m)
        x = np.zeros(m)
        r = b.copy()
        p = r.copy()
        gamma = dot(r, r)
        hist = [np.sqrt(gamma)]
        eps_tol = np.sqrt(np.finfo(float).eps) * (1 + hist[0])
        it = 0
        while hist[-1] > eps_tol and it < 2 * n:
            Ap = spmv(p)
            alpha = gamma / dot(p, Ap)
            x += alpha * p
            r -= alpha * Ap
            gn = dot(r, r)
            hist.append(np.sqrt(gn))
            if hist[-1] > eps_tol:
                beta = gn / gamma
                gamma = gn
                p = r + beta * p
            it += 1
        xs = [None] * world
        dist.all_gather_object(xs, x)
        if rank == 0:
            q.put(("ok", it, hist, np.concatenate(xs)))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put(("err", f"rank {rank}: {e}\n{traceback.format_exc()}"))


@pytest.mark.parametrize("world,kind,n1", [(2, "poisson3d", 10), (3, "poisson3d", 9), (2, "kron_unsymmetric", 6)])
def test_distributed_plan_and_cg_gloo(oracle, world, kind, n1):
    import multiprocessing as mp
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = _free_port()
    procs = [mpctx.Process(target=_worker, args=(r, world, port, n1, kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        msg = q.get(timeout=240)
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    assert msg[0] == "ok", msg[1]
    _, it, hist, x = msg
    if kind == "poisson3d":
        A = oracle.poisson3d(n1)
        ref = oracle.cg(A, np.ones(A.n), history=True)
        assert it == ref.niter
        assert np.allclose(hist, ref.residuals, rtol=1e-10)
        assert np.allclose(x, ref.x, atol=1e-11)
