#!/usr/bin/env python3
"""Rate of the batched small-QP path per shape: 4096 instances of one random pattern per shape through osqp_amd_batch_solve
(host arrays in, so the wall time includes the PCIe copies -- the KERNEL time is what rocprofv3 reports for the same run:
tools/gpu_r5_batch2.sh), iterations and factorisations per instance from the info rows, and the multiply-add model
   flops = iters * (2 n^2 + 4 nnz(A) + 2 nnz(P full)) + factorisations * (2 n^3 + 2 sum_rows len^2)
(dense product with the inverse, the two sparse products with A, the product with P; symmetric Gauss-Jordan inversion and
assembly of P + sigma I + A' rho A) that turns a kernel time into a per-flop rate comparable across shapes.
usage: python tools/batch_shapes.py [count]"""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import osqp_jl_amd as oq  # noqa: E402
from osqp_jl_amd import batch  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lib = oq.load_library()
OPTS = dict(verbose=False, eps_abs=1e-4, eps_rel=1e-4, adaptive_rho_interval=50, max_iter=4000)


def family(n, m, seed, tridiag=False, pat_A=None):
    from test_batch_gpu import _family
    return _family(n, m, 8, seed, tridiagonal_P=tridiag, pat_A=pat_A, equalities=False)


def tile(args, count):
    pat_P, pat_A, Px, Ax, q, l, u = args
    reps = (count + Px.shape[0] - 1) // Px.shape[0]
    t = lambda a: np.ascontiguousarray(np.tile(a, (reps, 1))[:count])
    return pat_P, pat_A, t(Px), t(Ax), t(q), t(l), t(u)


shapes = [("n=32 m=50", 32, 50, False), ("n=64 m=100", 64, 100, False), ("n=96 m=180", 96, 180, False),
          ("n=100 m=200 tridiagonal P", 100, 200, True), ("n=128 m=200", 128, 200, False)]
out = []
for name, n, m, tri in shapes:
    args = tile(family(n, m, 1000 * n + m, tridiag=tri)[0], count)
    batch.solve_batch(lib, *args, **OPTS)  # warm
    t0 = time.perf_counter()
    x, y, info = batch.solve_batch(lib, *args, **OPTS)
    wall = time.perf_counter() - t0
    kernel = lib.osqp_amd_batch_last_kernel()
    iters = float(np.sum(info[:, 0]))
    facts = float(np.sum(info[:, 5] + 1)) if info.shape[1] > 5 else float(count)
    A = args[1]
    rows = np.diff(sp.csr_matrix(A).indptr)
    nnzF = 2 * args[0].nnz - n
    flops = iters * (2.0 * n * n + 4.0 * A.nnz + 2.0 * nnzF) + facts * (2.0 * n ** 3 + 2.0 * float(np.sum(rows.astype(float) ** 2)))
    rec = dict(shape=name, n=n, m=m, nnz_A=int(A.nnz), kernel=int(kernel), instances=count, solved=int(np.sum(info[:, 1] == 1)),
               mean_iters=iters / count, mean_factorisations=facts / count, model_gflop=flops / 1e9, wall_ms_incl_pcie=1e3 * wall)
    out.append(rec)
    print(json.dumps(rec))
