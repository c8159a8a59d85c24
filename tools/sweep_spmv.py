#!/usr/bin/env python3
"""Time the dominant SpMV kernel (HIP events, osqp_amd_time_kernel) under the env-var tunables.
Usage: python tools/sweep_spmv.py rand-1e6|rand-1e5 VAR=VAL ... (one configuration per process)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    os.environ[k] = v
import bench, osqp_jl_amd as oq
kind, n, per_row, linsys = bench.WORKLOADS[sys.argv[1]]
lib = oq.load_library()
m = oq.Model(lib)
oq.setup_generated(m, kind, n, per_row, 1, linsys_solver=linsys, **bench.SETTINGS)
st = oq.stats(m)
out = {"cfg": sys.argv[2:]}
for which, name in ((0, "A"), (1, "At"), (2, "P")):
    ms = float(lib.osqp_amd_time_kernel(m.workspace, which, 10))
    out[name + "_ms"] = round(ms, 4)
out["A_GBs"] = round(st[10] / out["A_ms"] / 1e6, 1)
print(json.dumps(out))
