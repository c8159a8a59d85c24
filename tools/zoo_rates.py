#!/usr/bin/env python3
"""ADMM iterations/s of the HIP engine (direct and PCG back-ends) and of the CPU oracle on the QP zoo at larger sizes."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import osqp_jl_amd as oq
import qp_zoo

SIZES = {
    "portfolio": dict(n=20000, k=200), "svm": dict(n=200, m=20000), "huber": dict(n=200, m=20000),
    "lasso_data": dict(n=500, m=10000), "equality_qp": dict(n=6000), "control": dict(nx=12, nu=6, T=800),
}
OPTS = dict(verbose=False, eps_abs=1e-4, eps_rel=1e-4, max_iter=4000, adaptive_rho_interval=50, check_termination=25, polish=False)
prod, ora = oq.load_library(), oq.load_library(oq.ORACLE_LIB_PATH)
labels = os.environ.get("ZOO_LABELS", "gpu_direct,gpu_pcg,cpu_ldl").split(",")
only = sys.argv[1:] or sorted(SIZES)
for name in only:
    prob = qp_zoo.ZOO[name](**SIZES[name])
    n, m = prob["P"].shape[0], prob["A"].shape[0]
    row = {"problem": name, "n": n, "m": m, "nnzA": int(prob["A"].nnz), "nnzP": int(prob["P"].nnz)}
    for label, lib, ls in (("gpu_direct", prod, "direct"), ("gpu_pcg", prod, "pcg"), ("cpu_ldl", ora, "qdldl")):
        if label not in labels:
            continue
        try:
            t0 = time.perf_counter()
            mdl = oq.Model(lib); oq.setup(mdl, linsys_solver=ls, **prob, **OPTS)
            ts = time.perf_counter() - t0
            t0 = time.perf_counter(); r = oq.solve(mdl); tt = time.perf_counter() - t0
            st = oq.stats(mdl)
            row[label] = {"status": r.info.status, "iter": int(r.info.iter), "setup_s": round(ts, 3), "solve_s": round(tt, 4),
                          "it_per_s": round(r.info.iter / tt, 1), "nnzL": st[4], "levels": st[5], "cg": st[6]}
            oq.clean(mdl)
        except Exception as e:
            row[label] = {"error": str(e)[:100]}
    print(json.dumps(row), flush=True)
