#!/usr/bin/env python3
"""osqp_update_A / osqp_update_P with every value on a compact (sliced-ELL only) workspace: where the time goes.
    python tools/update_probe.py [n] [per_row]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import osqp_jl_amd as oq
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 300
lib = oq.load_library()
m = oq.Model(lib)
oq.setup_generated(m, 0, n, k, 1, linsys_solver="pcg", **bench.SETTINGS)
st = oq.stats(m); nnzA, nnzPt = int(st[1]), int(st[3])
r = oq.solve(m); print("solve 1:", r.info.status, r.info.iter, "%.3f s" % r.info.solve_time, "compact", st[18], flush=True)
rng = np.random.default_rng(0)
Ax = rng.standard_normal(nnzA)
for rep in range(2):
    t0 = time.time(); oq.update(m, Ax=Ax); t1 = time.time()
    r = oq.solve(m); print("update_A (%.1e values): wall %.3f s, info.update_time %.3f s; solve: %s %d it %.3f s" % (nnzA, t1 - t0, r.info.update_time, r.info.status, r.info.iter, r.info.solve_time), flush=True)
idx = np.arange(0, nnzA, 1000)
t0 = time.time(); oq.update(m, Ax=Ax[idx] * 0.5, Ax_idx=idx); t1 = time.time()
r = oq.solve(m); print("update_A by index (%d values): wall %.3f s, info.update_time %.3f s; solve: %s %d it" % (len(idx), t1 - t0, r.info.update_time, r.info.status, r.info.iter), flush=True)
