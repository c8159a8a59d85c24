"""A 3-D grid QP (tests/qp_zoo.py grid3d: separators of ~g^2 nodes) through the direct back-end at a size it was never run at:
setup stages, the form of the factor (stats), iteration rate, refactorisation time, agreement with the CPU oracle's solution.
usage: python tools/grid3d_probe.py [g=40] [--oracle]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import osqp_jl_amd as oq  # noqa: E402
import qp_zoo  # noqa: E402

g = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 40
prob = qp_zoo.grid3d(g)
n, m = prob["P"].shape[0], prob["A"].shape[0]
opts = dict(verbose=False, eps_abs=1e-4, eps_rel=1e-4, max_iter=4000, adaptive_rho_interval=50)
mdl = oq.Model(oq.load_library())
t0 = time.perf_counter()
oq.setup(mdl, linsys_solver="direct", **opts, **prob)
setup = time.perf_counter() - t0
st = oq.stats(mdl)
t0 = time.perf_counter()
r = oq.solve(mdl)
solve = time.perf_counter() - t0
oq.update_settings(mdl, rho=0.1)
t0 = time.perf_counter()
r2 = oq.solve(mdl)
solve2 = time.perf_counter() - t0
st2 = oq.stats(mdl)
print("second solve (warm): %d iterations in %.4f s = %.0f it/s; tree restarts so far %d; bytes of a solve %.3g" % (r2.info.iter, solve2, r2.info.iter / max(solve2, 1e-9), st2[21], st2[11]))
ts = []
for k in range(12):
    t0 = time.perf_counter()
    oq.update_settings(mdl, rho=0.1 + 0.01 * (k % 5))
    ts.append(time.perf_counter() - t0)
ts = sorted(ts[2:])
print("grid3d g=%d n=%d m=%d N=%d: setup %.2f s, nnz(L) %.3g, pivot levels %d, supernode levels %d, multifrontal %d, lean %d, dense top %d; "
      "%s in %d iterations, %.4f s = %.0f it/s incl. refactorisations (%d); refactorisation median %.2f ms" % (
          g, n, m, n + m, setup, st[4], st[5], st[19], st[22], st[23], st[25], r.info.status, r.info.iter, solve, r.info.iter / solve,
          r.info.rho_updates, 1e3 * ts[len(ts) // 2]))
if "--oracle" in sys.argv:
    mo = oq.Model(oq.load_library(oq.ORACLE_LIB_PATH))
    t0 = time.perf_counter()
    oq.setup(mo, linsys_solver="qdldl", **opts, **prob)
    ro = oq.solve(mo)
    print("oracle: %s in %d iterations, %.2f s setup + solve; max |dx| %.2e (|x| %.2e), max |dy| %.2e" % (
        ro.info.status, ro.info.iter, time.perf_counter() - t0, np.max(np.abs(ro.x - r.x)), np.max(np.abs(ro.x)), np.max(np.abs(ro.y - r.y))))
