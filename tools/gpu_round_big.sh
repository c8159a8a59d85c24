cd $GRAFT_REPO_ROOT
timeout 1200 python - <<'PY'
import time, json, sys
sys.path.insert(0, '.')
import bench, osqp_jl_amd as oq
lib = oq.load_library()
m = oq.Model(lib)
t0 = time.perf_counter()
oq.setup_generated(m, 0, 2000000, 1000, 1, linsys_solver="pcg", **bench.SETTINGS)
ts = time.perf_counter() - t0
st = oq.stats(m)
t0 = time.perf_counter(); r = oq.solve(m); tt = time.perf_counter() - t0
st2 = oq.stats(m)
ms = [float(lib.osqp_amd_time_kernel(m.workspace, w, 5)) for w in (0, 1, 2)]
print(json.dumps({"n": 2000000, "per_row": 1000, "nnz_A": st[1], "setup_s": round(ts, 2), "status": r.info.status, "iter": int(r.info.iter),
                  "solve_s": round(tt, 3), "cg_total": st2[6], "device_gb": round(st2[9] / 1e9, 1), "peak_gb": round(st2[20] / 1e9, 1),
                  "compact": st2[18], "spmv_ms_A_At_P": [round(x, 3) for x in ms], "spmv_A_GBs": round(st[10] / ms[0] / 1e6, 1)}))
PY
