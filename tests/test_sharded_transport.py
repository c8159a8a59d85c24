"""The host-staged transport of the row-sharded path (osqp_jl_amd/sharded.py) on CPU: world_size 2 over `gloo`.
The communicator object is created through the C ABI (no device work happens at creation) and its all-gather
callback is driven directly, exactly as the library drives it with its pinned staging buffer."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import osqp_jl_amd as oq
    from osqp_jl_amd import sharded

    comm = sharded.HostComm(lib=oq.load_library())
    assert (comm.rank, comm.world) == (rank, world) and comm.handle
    out = []
    for count in (1, 6, 1000):  # scalar slots, a PCG slot range, a vector chunk
        buf = np.full(world * count, -1.0)
        buf[rank * count:(rank + 1) * count] = 100.0 * rank + np.arange(count)
        rc = comm._cb(None, buf.ctypes.data_as(C.POINTER(C.c_double)), count)
        out.append((rc, buf.copy()))
    comm.close()
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_host_allgather_two_ranks():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in (0, 1):
        for (rc, buf), count in zip(got[rank], (1, 6, 1000)):
            expect = np.concatenate([100.0 * r + np.arange(count) for r in range(2)])
            assert rc == 0 and np.array_equal(buf, expect)


def test_block_range_matches_library_cut():
    from osqp_jl_amd import sharded

    assert [sharded.block_range(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert sharded.block_range(1_000_000, 8, 7) == (875_000, 1_000_000)
    assert sharded.block_range(5, 4, 3) == (5, 5)  # an empty block: the library refuses such a partition at setup
