"""Batched small-QP path and its multi-GPU sharding.

`count` independent QPs that share one sparsity pattern are solved one per
workgroup on the device (csrc/batch.hip).  Across GPUs the instance range is cut
into contiguous equal blocks, one per rank (one process per GPU); there is no
communication during the solve.  The only collective is the final gather of
the packed per-rank results [x | y | info]: an in-place all-gather on the
library's own communicator (`sharded.RcclComm`: ncclAllGather over xGMI issued by
libosqp_amd.so itself; `sharded.HostComm`: pinned-host staging + a caller callback)
-- `MpcBatch` below, which needs no torch at all: its packed result array is a `DeviceArray` allocated through the library.
"""
import ctypes as C

import numpy as np

from . import types as T
from .interface import _as_f64, _fptr, _iptr, make_settings, OSQPError
from .constants import status_map

MPC_N, MPC_M = 100, 200
INFO_COLS = 4  # iter, status_val, pri_res, dua_res


def solve_batch(lib, P, A, Px_all, Ax_all, q_all, l_all, u_all, device=0, **settings):
    """Host-pointer entry point (osqp_amd_batch_solve).  P (upper triangle) and A are
    scipy CSC matrices giving the shared pattern; the *_all arrays are [count x .]."""
    import scipy.sparse as sp

    P = sp.csc_matrix(sp.triu(P)); P.sort_indices()
    A = sp.csc_matrix(A); A.sort_indices()
    n, m = A.shape[1], A.shape[0]
    Px_all, Ax_all = _as_f64(Px_all), _as_f64(Ax_all)
    q_all, l_all, u_all = _as_f64(q_all), _as_f64(l_all), _as_f64(u_all)
    count = q_all.shape[0]
    assert Px_all.shape == (count, P.nnz) and Ax_all.shape == (count, A.nnz)
    assert l_all.shape == (count, m) and u_all.shape == (count, m)
    Pp, Pi = np.ascontiguousarray(P.indptr, dtype=np.int64), np.ascontiguousarray(P.indices, dtype=np.int64)
    Ap, Ai = np.ascontiguousarray(A.indptr, dtype=np.int64), np.ascontiguousarray(A.indices, dtype=np.int64)
    stgs = make_settings(lib, settings)
    x = np.empty((count, n)); y = np.empty((count, m))
    infos = (T.CInfo * count)()
    rc = lib.osqp_amd_batch_solve(count, n, m, _iptr(Pp), _iptr(Pi), _fptr(Px_all), _iptr(Ap), _iptr(Ai), _fptr(Ax_all),
                                  _fptr(q_all), _fptr(l_all), _fptr(u_all), C.byref(stgs), _fptr(x), _fptr(y), infos, device)
    if rc != 0:
        raise OSQPError("Error in batched solve: " + lib.osqp_amd_last_error().decode())
    info = np.array([[i.iter, i.status_val, i.pri_res, i.dua_res, i.obj_val, i.rho_updates] for i in infos])
    return x, y, info


def shard_range(count, rank, world):
    """Contiguous equal blocks: instance i -> rank floor(i / (count / world))."""
    if count % world != 0:
        raise ValueError("instance count must be divisible by the number of ranks")
    per = count // world
    return rank * per, per


def device_mpc_solver(lib, device, **settings):
    """Returns solve(first, count, seed) -> packed torch tensor [count x (n + m + 4)] on `device`,
    generated and solved in HBM by osqp_amd_batch_solve_generated."""
    import torch

    stgs = make_settings(lib, settings)

    def solve(first, count, seed):
        packed = torch.empty((count, MPC_N + MPC_M + INFO_COLS), dtype=torch.float64, device=f"cuda:{device}")
        x = torch.empty((count, MPC_N), dtype=torch.float64, device=f"cuda:{device}")
        y = torch.empty((count, MPC_M), dtype=torch.float64, device=f"cuda:{device}")
        info = torch.empty((count, INFO_COLS), dtype=torch.float64, device=f"cuda:{device}")
        rc = lib.osqp_amd_batch_solve_generated(first, count, seed, C.byref(stgs), x.data_ptr(), y.data_ptr(), info.data_ptr(), device)
        if rc != 0:
            raise OSQPError("Error in batched solve: " + lib.osqp_amd_last_error().decode())
        packed[:, :MPC_N] = x
        packed[:, MPC_N:MPC_N + MPC_M] = y
        packed[:, MPC_N + MPC_M:] = info
        return packed

    return solve


def split_packed(full):
    """(x, y, info) views of a packed [count x (n + m + 4)] array."""
    return full[:, :MPC_N], full[:, MPC_N:MPC_N + MPC_M], full[:, MPC_N + MPC_M:]


def solve_mpc_sharded(solver, count, seed, rank=0, world=1, gather=None):
    """The host logic of the sharded batch with the two device steps passed in: each rank solves its block with
    `solver(first, count, seed)` -> packed [per x 304]; `gather(full, rank, per)` fills the other ranks' rows of
    `full` in place (one collective).  Returns (x, y, info) views of the whole batch.  `MpcBatch` is the product form
    of the same steps (both inside libosqp_amd.so); the CPU tests drive this one with the oracle and gloo."""
    first, per = shard_range(count, rank, world)
    mine = solver(first, per, seed)
    if world > 1:
        full = mine.new_empty((count, mine.shape[1]))
        full[first:first + per] = mine
        gather(full, rank, per)
    else:
        full = mine
    return split_packed(full)


class DeviceArray:
    """A [rows x cols] fp64 array in HBM allocated through the library (osqp_amd_device_alloc): what `MpcBatch.solve` writes
    into.  `numpy()` downloads it; `clone()` copies it on the device."""

    def __init__(self, lib, rows, cols, device=0):
        self.lib, self.shape, self.device = lib, (int(rows), int(cols)), int(device)
        self.nbytes = 8 * self.shape[0] * self.shape[1]
        self.ptr = lib.osqp_amd_device_alloc(self.nbytes, self.device)
        if not self.ptr:
            raise OSQPError("device allocation failed: " + lib.osqp_amd_last_error().decode())

    def data_ptr(self):
        return self.ptr

    def numpy(self):
        out = np.empty(self.shape)
        if self.lib.osqp_amd_device_copy(out.ctypes.data_as(C.c_void_p), self.ptr, self.nbytes, 0, self.device) != 0:
            raise OSQPError("device copy failed: " + self.lib.osqp_amd_last_error().decode())
        return out

    def clone(self):
        other = DeviceArray(self.lib, self.shape[0], self.shape[1], self.device)
        if self.lib.osqp_amd_device_copy(other.ptr, self.ptr, self.nbytes, 2, self.device) != 0:
            raise OSQPError("device copy failed: " + self.lib.osqp_amd_last_error().decode())
        return other

    def free(self):
        if self.ptr:
            self.lib.osqp_amd_device_free(self.ptr, self.device)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class MpcBatch:
    """`total` MPC instances cut over the ranks of `comm` (None: one rank), resident in HBM; `solve()` = rows K11 + K12
    in one library call (osqp_amd_batch_mpc_solve): this rank's block, one workgroup per instance, written in place into
    the packed device array, then one in-place all-gather on the library's communicator."""

    def __init__(self, lib, total, seed=1, device=0, comm=None, **settings):
        self.lib, self.total, self.device, self.comm = lib, int(total), int(device), comm
        self.world = comm.world if comm is not None else 1
        self.rank = comm.rank if comm is not None else 0
        self.first, self.per = shard_range(self.total, self.rank, self.world)
        stgs = make_settings(lib, settings)
        self.handle = C.c_void_p()
        rc = lib.osqp_amd_batch_mpc_create(C.byref(self.handle), self.total, seed, C.byref(stgs),
                                           comm.handle if comm is not None else None, self.device)
        if rc != 0:
            raise OSQPError("Error in batched setup: " + lib.osqp_amd_last_error().decode())

    def alloc(self):
        return DeviceArray(self.lib, self.total, MPC_N + MPC_M + INFO_COLS, self.device)

    def solve(self, out=None):
        packed = self.alloc() if out is None else out
        rc = self.lib.osqp_amd_batch_mpc_solve(self.handle, packed.data_ptr())
        if rc != 0:
            raise OSQPError("Error in batched solve: " + self.lib.osqp_amd_last_error().decode())
        return packed

    def close(self):
        if self.handle:
            self.lib.osqp_amd_batch_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def status_names(info):
    return [status_map[int(v)] for v in info[:, 1]]
