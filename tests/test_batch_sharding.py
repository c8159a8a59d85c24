"""The N > 1 path of the batched solve on CPU: world_size 2 over `gloo`.  The
per-rank solver here is the CPU oracle (tests may use it); what is under test is
the sharding, the packing and the single all-gather of osqp_jl_amd/batch.py --
the same code that runs over RCCL with the device solver."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

COUNT, SEED = 8, 4
OPTS = dict(verbose=False, eps_abs=1e-5, eps_rel=1e-5, adaptive_rho_interval=50)


def oracle_solver(first, count, seed):
    import osqp_jl_amd as oq
    from osqp_jl_amd import batch

    lib = oq.load_library(oq.ORACLE_LIB_PATH)
    out = torch.empty((count, batch.MPC_N + batch.MPC_M + batch.INFO_COLS), dtype=torch.float64)
    for k in range(count):
        m = oq.Model(lib)
        oq.setup_generated(m, 2, 100, first + k, seed, **OPTS)
        r = oq.solve(m)
        out[k, :100] = torch.from_numpy(r.x)
        out[k, 100:300] = torch.from_numpy(r.y)
        out[k, 300:] = torch.tensor([r.info.iter, r.info.status_val, r.info.pri_res, r.info.dua_res])
    return out


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from osqp_jl_amd import batch

    def gather(full, rank, per):  # in place, like the library's communicator: chunk `rank` is filled in on entry
        dist.all_gather_into_tensor(full, full[rank * per:(rank + 1) * per].clone())

    x, y, info = batch.solve_mpc_sharded(oracle_solver, COUNT, SEED, rank=rank, world=world, gather=gather)
    q.put((rank, x.numpy().copy(), y.numpy().copy(), info.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_batch_equals_single_process():
    import subprocess

    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    from osqp_jl_amd import batch

    x1, y1, i1 = batch.solve_mpc_sharded(oracle_solver, COUNT, SEED)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, x, y, info in got:  # every rank holds the whole batch, identical to the single-process result
        assert np.array_equal(x, x1.numpy()) and np.array_equal(y, y1.numpy()) and np.array_equal(info[:, :2], i1.numpy()[:, :2])
    assert np.all(i1.numpy()[:, 1] == 1)


def test_shard_range():
    from osqp_jl_amd import batch

    assert [batch.shard_range(4096, r, 8) for r in (0, 1, 7)] == [(0, 512), (512, 512), (3584, 512)]
    with pytest.raises(ValueError):
        batch.shard_range(10, 0, 4)
