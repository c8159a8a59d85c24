#!/usr/bin/env python3
"""Does the automatic back-end rule pick the faster back-end?  Random QPs (generator kind 0) over a size ladder:
ADMM it/s of auto / forced direct / PCG on the GPU and of the CPU oracle's LDL'."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import osqp_jl_amd as oq
prod, ora = oq.load_library(), oq.load_library(oq.ORACLE_LIB_PATH)
OPTS = dict(verbose=False, eps_abs=1e-4, eps_rel=1e-4, adaptive_rho_interval=50, check_termination=25, max_iter=4000)
for n, k in ((500, 5), (2000, 10), (5000, 10), (20000, 20)):
    row = {"n": n, "per_row": k}
    for label, lib, ls in (("auto", prod, "qdldl"), ("direct", prod, "direct"), ("pcg", prod, "pcg"), ("cpu", ora, "qdldl")):
        if label in ("direct", "cpu") and n > 5000:
            continue
        m = oq.Model(lib)
        t0 = time.perf_counter(); oq.setup_generated(m, 0, n, k, 3, linsys_solver=ls, **OPTS); ts = time.perf_counter() - t0
        t0 = time.perf_counter(); r = oq.solve(m); tt = time.perf_counter() - t0
        st = oq.stats(m)
        row[label] = {"backend": int(st[0]), "iter": int(r.info.iter), "it_per_s": round(r.info.iter / tt, 1), "setup_s": round(ts, 3), "levels": int(st[5]), "nnzL": int(st[4])}
        oq.clean(m)
    print(json.dumps(row), flush=True)
