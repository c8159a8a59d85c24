"""What the single-reduction CG recurrence (Chronopoulos & Gear) would cost in CG iterations on the bench matrices: the CPU
oracle with its usual recurrence and with the experiment switch OSQP_ORACLE_PCG_SINGLE_REDUCTION=1 (oracle/pcg.c), same
problems, same tolerance rule.   python tools/cg_recurrence_counts.py [n per_row seeds...] > profiles/r04_cg_recurrence_counts.md"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, json
sys.path.insert(0, sys.argv[1])
import bench, osqp_jl_amd as oq
n, k, seed = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ora = oq.load_library(oq.ORACLE_LIB_PATH)
m = oq.Model(ora)
oq.setup_generated(m, 0, n, k, seed, linsys_solver="pcg", **bench.SETTINGS)
r = oq.solve(m)
st = oq.stats(m)
print(json.dumps({"iter": int(r.info.iter), "status": r.info.status, "cg": int(st[6]), "pri": r.info.pri_res, "dua": r.info.dua_res, "obj": r.info.obj_val}))
"""


def run(n, k, seed, single):
    env = dict(os.environ)
    env["OSQP_ORACLE_PCG_SINGLE_REDUCTION"] = "1" if single else "0"
    out = subprocess.run([sys.executable, "-c", CHILD, ROOT, str(n), str(k), str(seed)], env=env, capture_output=True, text=True, check=True)
    return json.loads(out.stdout.strip().splitlines()[-1])


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    seeds = [int(a) for a in sys.argv[3:]] or [1, 2, 3, 4]
    print("| n | per row | seed | ADMM it (two reductions / one) | CG iterations to eps (two / one) | objective difference |")
    print("|---|---|---|---|---|---|")
    tot = [0, 0]
    for sd in seeds:
        a, b = run(n, k, sd, False), run(n, k, sd, True)
        tot[0] += a["cg"]; tot[1] += b["cg"]
        print("| %d | %d | %d | %d / %d | %d / %d | %.2e |" % (n, k, sd, a["iter"], b["iter"], a["cg"], b["cg"], abs(a["obj"] - b["obj"])))
    print("\nsum of CG iterations: %d with the two-reduction recurrence, %d with the single-reduction one" % tuple(tot))
