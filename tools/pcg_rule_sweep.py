#!/usr/bin/env python3
"""Time-to-eps of the PCG path over several seeds (the 25-iteration check granularity makes a single seed noisy).
usage: pcg_rule_sweep.py WORKLOAD SEEDS   (env OSQP_AMD_PCG_LAMBDA / OSQP_AMD_PCG_EXTRAP select the rule)"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, osqp_jl_amd as oq
kind, n, per_row, linsys = bench.WORKLOADS[sys.argv[1]]
seeds = range(1, int(sys.argv[2]) + 1)
lib = oq.load_library()
settings = dict(bench.SETTINGS)
if os.environ.get("RHO_INTERVAL"):
    settings["adaptive_rho_interval"] = int(os.environ["RHO_INTERVAL"])
tot_t = tot_it = tot_cg = 0
rows = []
for seed in seeds:
    m = oq.Model(lib)
    oq.setup_generated(m, kind, n, per_row, seed, linsys_solver=linsys, **settings)
    t0 = time.perf_counter(); r = oq.solve(m); dt = time.perf_counter() - t0
    st = oq.stats(m)
    rows.append((seed, r.info.status, int(r.info.iter), round(dt, 4), int(st[6])))
    tot_t += dt; tot_it += r.info.iter; tot_cg += st[6]
    oq.clean(m)
print(json.dumps({"workload": sys.argv[1], "lambda": os.environ.get("OSQP_AMD_PCG_LAMBDA"), "extrap": os.environ.get("OSQP_AMD_PCG_EXTRAP"), "rho_interval": settings["adaptive_rho_interval"],
                  "sum_time_s": round(tot_t, 4), "sum_iters": int(tot_it), "sum_cg": int(tot_cg), "runs": rows}))
