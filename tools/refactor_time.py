"""Wall time of one numeric refactorisation (osqp_update_rho: rho vector, KKT scatter, numeric LDL', inverted blocks) on
the long-horizon control problem, multifrontal (default) against level by level (OSQP_AMD_MF=0).
usage: python tools/refactor_time.py [T ...]      (control problem of horizon T)
       python tools/refactor_time.py grid g ...   (g x g grid QP, tests/qp_zoo.py grid2d; the second run of a pair is OSQP_AMD_MF_BIG=0:
                                                    no fronts out of global memory, i.e. the level-by-level factorisation of round 5)"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np
    import osqp_jl_amd as oq
    import qp_zoo
    T = int(sys.argv[2])
    prob = qp_zoo.grid2d(T) if os.environ.get("REFACTOR_GRID") == "1" else qp_zoo.control(nx=12, nu=6, T=T)
    m = oq.Model(oq.load_library())
    t0 = time.perf_counter()
    oq.setup(m, linsys_solver="direct", verbose=False, adaptive_rho=False, **prob)
    setup = time.perf_counter() - t0
    st = oq.stats(m)
    ts = []
    for k in range(24):
        t0 = time.perf_counter()
        oq.update_settings(m, rho=0.1 + 0.01 * (k % 5))
        ts.append(time.perf_counter() - t0)
    ts = sorted(ts[4:])
    t0 = time.perf_counter()
    r = oq.solve(m)
    solve = time.perf_counter() - t0
    print("T=%d N=%d nnzL=%d levels=%d sn_levels=%d mf=%d setup %.3f s  refactor median %.3f ms min %.3f ms  (%s, %d it, %.0f it/s)" % (
        T, prob["P"].shape[0] + prob["A"].shape[0], st[4], st[5], st[19], st[22], setup, 1e3 * ts[len(ts) // 2], 1e3 * ts[0], r.info.status, r.info.iter, r.info.iter / solve))
    sys.exit(0)

if len(sys.argv) > 1 and sys.argv[1] == "grid":
    for g in [int(a) for a in sys.argv[2:]] or [700]:
        for big in ("1", "0"):
            env = dict(os.environ, REFACTOR_GRID="1", OSQP_AMD_MF_BIG=big)
            subprocess.call([sys.executable, os.path.abspath(__file__), "--child", str(g)], env=env)
    sys.exit(0)
for T in [int(a) for a in sys.argv[1:]] or [800, 8000]:
    for mf in ("1", "0"):
        env = dict(os.environ, OSQP_AMD_MF=mf)
        subprocess.call([sys.executable, os.path.abspath(__file__), "--child", str(T)], env=env)
