"""Row-sharded solve of ONE large QP over several GPUs (include/osqp_amd.h).

One process per GPU.  Every rank builds the same problem through the ordinary `setup` / `setup_generated` with
`comm=...`; the library keeps this rank's row block of A, A' and P and exchanges the input vector of each sparse
product (n or m doubles) plus a few scalars through the communicator -- nothing else crosses ranks.  The
reference has no multi-device path; the user-facing calls stay the reference's (`solve`, `update_q`,
`update_bounds`, `warm_start`), every rank issues them in the same order and gets the full solution back.

Two transports:
  RcclComm -- ncclAllGather on the engine's own HIP stream (RCCL over xGMI).  The ncclUniqueId is created on rank 0
              by the library and passed around with torch.distributed; torch is not on the data path.
  HostComm -- the library stages through pinned host memory and calls back into Python, which runs the all-gather
              on a torch.distributed group of CPU tensors (gloo).  Any number of ranks can share one GPU this way,
              which is how the sharded path is tested on a single-GPU box.
"""
import ctypes as C
import os

import numpy as np

from . import types as T
from .interface import OSQPError


def _dist():
    import torch.distributed as dist
    return dist


def torch_librccl_path():
    """The librccl PyTorch ships (the copy an nccl process group of this process already holds), or None."""
    try:
        import torch
    except ImportError:
        return None
    p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return p if os.path.exists(p) else None


class _Comm:
    def __init__(self, lib, rank, world):
        self.lib = lib if lib is not None else T.load_library()
        self.rank, self.world = int(rank), int(world)
        self.handle = C.c_void_p()

    def close(self):
        if self.handle:
            self.lib.osqp_amd_comm_destroy(self.handle)
            self.handle = C.c_void_p()

    def info(self):
        """(rank, world, ranks the transport itself reports: RCCL's ncclCommCount; -1 if it cannot say)."""
        r, w, t = T.c_int(), T.c_int(), T.c_int()
        self._check(self.lib.osqp_amd_comm_info(self.handle, C.byref(r), C.byref(w), C.byref(t)), "osqp_amd_comm_info")
        return int(r.value), int(w.value), int(t.value)

    def _check(self, rc, what):
        if rc != 0:
            msg = self.lib.osqp_amd_last_error()
            raise OSQPError("%s failed: %s" % (what, msg.decode() if msg else rc))


class HostComm(_Comm):
    """Host-staged all-gather over a torch.distributed group of CPU tensors (gloo)."""

    def __init__(self, group=None, lib=None):
        dist = _dist()
        super().__init__(lib, dist.get_rank(group), dist.get_world_size(group))
        self.group = group
        self._cb = T.ALLGATHER_FN(self._allgather)  # keep the trampoline alive
        self._check(self.lib.osqp_amd_comm_create_host(C.byref(self.handle), self.rank, self.world, self._cb, None),
                    "osqp_amd_comm_create_host")

    def _allgather(self, ctx, buf, count):
        try:
            import torch
            dist = _dist()
            full = torch.from_numpy(np.ctypeslib.as_array(buf, shape=(self.world * count,)))
            mine = full[self.rank * count:(self.rank + 1) * count].clone()
            dist.all_gather_into_tensor(full, mine, group=self.group)
            return 0
        except Exception:  # an exception cannot cross the C frame: report failure instead
            import traceback
            traceback.print_exc()
            return 1


class RcclComm(_Comm):
    """ncclAllGather on the engine's stream.  `group`: any torch.distributed group, used once for the 128-byte id."""

    def __init__(self, group=None, lib=None, librccl_path=None):
        dist = _dist()
        super().__init__(lib, dist.get_rank(group), dist.get_world_size(group))
        path = librccl_path or torch_librccl_path()
        cpath = path.encode() if path else None
        uid = [None]
        if self.rank == 0:
            raw = C.create_string_buffer(128)
            self._check(self.lib.osqp_amd_comm_unique_id(raw, cpath), "osqp_amd_comm_unique_id")
            uid[0] = raw.raw
        dist.broadcast_object_list(uid, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        raw = C.create_string_buffer(uid[0], 128)
        self._check(self.lib.osqp_amd_comm_create_rccl(C.byref(self.handle), self.rank, self.world, raw, cpath),
                    "osqp_amd_comm_create_rccl")


def block_range(total, world, rank):
    """Rows [first, last) of a length-`total` dimension owned by `rank` (ceil-sized blocks, as the library cuts them)."""
    chunk = (total + world - 1) // world
    first = min(rank * chunk, total)
    return first, min(first + chunk, total)
