"""Soak of the one-launch tree where it is sized beyond what is resident (2-D / 3-D structures): many solves and refactorisations
in one process, the count of time-out restarts (osqp_amd_get_stats slot 21) at the end -- it has to stay 0.
usage: python tools/tree_soak.py [grid2d 700] [solves=200]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import osqp_jl_amd as oq  # noqa: E402
import qp_zoo  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "grid2d"
size = int(sys.argv[2]) if len(sys.argv) > 2 else 700
solves = int(sys.argv[3]) if len(sys.argv) > 3 else 200
prob = qp_zoo.grid3d(size) if kind == "grid3d" else qp_zoo.grid2d(size)
m = oq.Model(oq.load_library())
oq.setup(m, linsys_solver="direct", verbose=False, eps_abs=1e-4, eps_rel=1e-4, adaptive_rho_interval=25, max_iter=4000, **prob)
rng = np.random.default_rng(0)
t0 = time.perf_counter()
iters = 0
for k in range(solves):
    oq.update(m, q=prob["q"] * (1.0 + 0.1 * rng.standard_normal()))
    if k % 7 == 3:
        oq.update_settings(m, rho=0.05 + 0.2 * rng.random())
    r = oq.solve(m)
    assert r.info.status == "Solved", r.info.status
    iters += r.info.iter
st = oq.stats(m)
print("%s %d: %d solves, %d iterations, %d factorisations in %.2f s; tree restarts %d" % (kind, size, solves, iters, st[8], time.perf_counter() - t0, st[21]))
