"""The RCCL transport of libosqp_amd.so (csrc/comm.hip: `RcclComm`) with TWO ranks on a box without GPUs.

RCCL refuses two ranks on one device and the round's GPU boxes have one, so until a multi-GPU node runs the bench the
transport's own logic -- the 128-byte unique id created on rank 0 and handed to the others, ncclCommInitRank with the
rank / world bookkeeping, the in-place ncclAllGather call whose send pointer lies inside the receive buffer, the
statistics -- would never execute with more than one rank.  Here it does: tests/stub_rccl.c is a host-side stand-in
for librccl (same five symbols + ncclCommCount, shared memory between the processes) that the library loads through
the very dlsym table it uses for the real one (`librccl_path`), with OSQP_AMD_RCCL_STUB=1 telling the C ABI that the
"device" buffers are host memory.  What crosses ranks is checked against the expected gather; the unique id travels
over a gloo group exactly as osqp_jl_amd.sharded.RcclComm sends it in production."""
import ctypes as C
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _build_stub():
    out = os.path.join(tempfile.gettempdir(), "libstub_rccl_%d.so" % os.getpid())
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-o", out, os.path.join(ROOT, "tests", "stub_rccl.c")])
    return out


def _worker(rank, world, port, stub, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OSQP_AMD_RCCL_STUB"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import osqp_jl_amd as oq
    from osqp_jl_amd import sharded

    lib = oq.load_library()
    comm = sharded.RcclComm(lib=lib, librccl_path=stub)  # id on rank 0 -> broadcast -> ncclCommInitRank on every rank
    out = {"info": comm.info(), "gathers": []}
    for count in (1, 6, 1000):  # scalar slots, a PCG slot range, a vector chunk
        buf = np.full(world * count, -1.0)
        buf[rank * count:(rank + 1) * count] = 100.0 * rank + np.arange(count)
        rc = lib.osqp_amd_comm_all_gather(comm.handle, buf.ctypes.data_as(C.c_void_p), count)
        out["gathers"].append((rc, buf.copy()))
    comm.close()
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_transport_two_ranks_through_the_stub():
    stub = _build_stub()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    procs = [ctx.Process(target=_worker, args=(r, world, port, stub, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    os.unlink(stub)
    for rank in range(world):
        assert got[rank]["info"] == (rank, world, world)  # rank, size, and the size the transport itself reports
        for (rc, buf), count in zip(got[rank]["gathers"], (1, 6, 1000)):
            expect = np.concatenate([100.0 * r + np.arange(count) for r in range(world)])
            assert rc == 0 and np.array_equal(buf, expect)
