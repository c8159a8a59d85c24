"""The largest single-GPU instance of the random family (n = m = 2e6, 1000 per row: 2e9 non-zeros in A, the most 32-bit
non-zero positions allow per matrix): setup, solve to eps = 1e-4, resident / peak device memory, SpMV time.
    python tools/largest_instance.py > profiles/r04_largest_instance.json"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import osqp_jl_amd as oq  # noqa: E402

n, k = 2_000_000, 1000
lib = oq.load_library()
m = oq.Model(lib)
t = time.time()
oq.setup_generated(m, 0, n, k, 1, linsys_solver="pcg", **bench.SETTINGS)
ts = time.time() - t
t = time.time()
r = oq.solve(m)
tt = time.time() - t
st = oq.stats(m)
ms = [float(lib.osqp_amd_time_kernel(m.workspace, w, 5)) for w in (0, 1, 2)]
out = {"n": n, "per_row": k, "nnz_A": float(n) * k, "setup_s": round(ts, 2), "status": r.info.status, "iter": int(r.info.iter),
       "solve_s": round(tt, 2), "cg_total": float(st[6]), "device_gb": round(st[9] / 1e9, 1), "peak_gb": round(st[20] / 1e9, 1),
       "compact": float(st[18]), "spmv_ms_A_At_P": [round(x, 3) for x in ms]}
print(json.dumps(out))
