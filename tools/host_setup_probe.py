#!/usr/bin/env python3
"""osqp_setup from the caller's host arrays (the reference's path) against the device generator, same problem: where the time goes.
    python tools/host_setup_probe.py [n] [per_row]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import osqp_jl_amd as oq
import bench
from test_gpu_parity import _data_to_scipy

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 300
lib = oq.load_library()
orc = oq.load_library(os.path.join(ROOT, "oracle", "_build", "libosqp_oracle.so"))
t0 = time.time(); d = orc.oracle_generate(0, n, k, 1); P, q, A, l, u = _data_to_scipy(d.contents); orc.oracle_data_free(d)
print("host data: n = %d, nnz(A) = %d, nnz(triu P) = %d, generated in %.1f s" % (n, A.nnz, P.nnz, time.time() - t0), flush=True)
for rep in range(2):
    m = oq.Model(lib); t0 = time.time(); oq.setup(m, P=P, q=q, A=A, l=l, u=u, linsys_solver="pcg", **bench.SETTINGS); t1 = time.time()
    r = oq.solve(m); print("host arrays : setup wall %.3f s (info.setup_time %.3f), solve %s %d it %.3f s" % (t1 - t0, r.info.setup_time, r.info.status, r.info.iter, r.info.solve_time), flush=True); oq.clean(m)
    m = oq.Model(lib); t0 = time.time(); oq.setup_generated(m, 0, n, k, 1, linsys_solver="pcg", **bench.SETTINGS); t1 = time.time()
    r = oq.solve(m); print("generated   : setup wall %.3f s (info.setup_time %.3f), solve %s %d it %.3f s" % (t1 - t0, r.info.setup_time, r.info.status, r.info.iter, r.info.solve_time), flush=True); oq.clean(m)
