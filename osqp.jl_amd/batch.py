"""Batched small-QP path and its multi-GPU sharding.

`count` independent QPs that share one sparsity pattern are solved one per
workgroup on the device (csrc/batch.hip).  Across GPUs the instance range is cut
into contiguous equal blocks, one per rank (one process per GPU); there is no
communication during the solve.  The only collective is the final gather of
the packed per-rank results [x | y | info] (RCCL `all_gather_into_tensor` when
the process group is `nccl`; the same code runs on `gloo` in the CPU tests).
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


def solve_mpc_sharded(solver, count, seed, rank=0, world=1, dist=None):
    """Each rank solves its block with `solver(first, count, seed)`; one all-gather of the
    packed block returns the whole batch on every rank.  Returns (x, y, info) views."""
    first, per = shard_range(count, rank, world)
    mine = solver(first, per, seed)
    if world > 1:
        import torch

        full = torch.empty((count, mine.shape[1]), dtype=mine.dtype, device=mine.device)
        dist.all_gather_into_tensor(full, mine.contiguous())
    else:
        full = mine
    return full[:, :MPC_N], full[:, MPC_N:MPC_N + MPC_M], full[:, MPC_N + MPC_M:]


def status_names(info):
    return [status_map[int(v)] for v in info[:, 1]]
