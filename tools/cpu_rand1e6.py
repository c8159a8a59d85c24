#!/usr/bin/env python3
"""One-off: the CPU oracle (oracle/, 1 thread, PCG back-end) on the headline workload itself -- rand-1e6
(n = m = 1e6, nnz(A) = 1e9, nnz(triu P) ~ 5e8), same generator, seed and settings as bench.py.  Takes minutes
(generation + Ruiz scaling over 1.5e9 entries, then ~1 minute per ADMM iteration), which is why bench.py's
bounded cpu leg cannot run it live and reads the committed record instead (profiles/r02_cpu_rand1e6.json).

    python tools/cpu_rand1e6.py [--n 1000000] [--per-row 1000] [--iters 3] [--out gpurun_out/r02_cpu_rand1e6.json]
"""
import argparse
import json
import os
import resource
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def meminfo():
    out = {}
    for line in open("/proc/meminfo"):
        k, v = line.split(":")
        if k in ("MemTotal", "MemAvailable"):
            out[k] = round(int(v.split()[0]) / 1048576.0, 1)  # GiB
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--per-row", type=int, default=1000)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--warm", type=int, default=1)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r02_cpu_rand1e6.json"))
    args = ap.parse_args()

    import osqp_jl_amd as oq
    import bench

    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    ora = oq.load_library(oq.ORACLE_LIB_PATH)
    rec = {"workload": f"rand n=m={args.n}, {args.per_row} per row, seed 1", "host_cores": os.cpu_count(), "threads": 1,
           "host_mem_gib": meminfo(), "kind": "port (oracle/, PCG back-end; libosqp is not in the image)"}
    try:
        rec["cpu_model"] = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    print(json.dumps(rec), flush=True)
    # the oracle keeps 64-bit indices: 16 B per stored entry, generator output + the workspace's own scaled copy
    need_gib = 2.0 * 16.0 * (args.n * args.per_row * 1.5) / 2**30 + 2.0
    rec["needed_gib_estimate"] = round(need_gib, 1)
    if rec["host_mem_gib"].get("MemAvailable", 0.0) < need_gib:
        rec.update({"value": None, "reason": "not run: %.0f GiB needed, %.0f GiB available" % (need_gib, rec["host_mem_gib"].get("MemAvailable", 0.0))})
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        json.dump(rec, open(args.out, "w"), indent=1)
        print(json.dumps(rec), flush=True)
        return
    m = oq.Model(ora)
    t0 = time.perf_counter()
    oq.setup_generated(m, 0, args.n, args.per_row, 1, linsys_solver="pcg", **bench.SETTINGS)
    rec["setup_s"] = round(time.perf_counter() - t0, 2)
    rec["peak_rss_gib_after_setup"] = round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1048576.0, 1)
    print(json.dumps(rec), flush=True)
    ws = m.workspace
    if args.warm:
        t0 = time.perf_counter()
        ora.osqp_amd_iterate(ws, args.warm)
        rec["warm_s"] = round(time.perf_counter() - t0, 2)
    st0 = oq.stats(m)
    t0 = time.perf_counter()
    ora.osqp_amd_iterate(ws, args.iters)  # one call: the CG start vectors carry over between the iterations, as in a solve
    spent = time.perf_counter() - t0
    st1 = oq.stats(m)
    rec.update({"iters": args.iters, "seconds": round(spent, 3), "value": round(args.iters / spent, 6), "unit": "iterations/s",
                "cg_iters_per_admm_iter": round((st1[6] - st0[6]) / args.iters, 3), "nnz_A": int(st1[1]), "nnz_P_triu": int(st1[3]),
                "peak_rss_gib": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1048576.0, 1),
                "note": "timed: one osqp_amd_iterate call = the ADMM iterations + the one residual evaluation it ends with (3 sparse products)"})
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(rec, open(args.out, "w"), indent=1)
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
