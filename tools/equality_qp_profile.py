"""equality_qp (tests/qp_zoo.py, n = 6000: dense P, nnz(L) = 1.85e7) through the direct back-end: one setup, one solve -- the command behind profiles/r04_kernel_stats_equality_qp.md (run under rocprofv3 --kernel-trace --stats)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import osqp_jl_amd as oq  # noqa: E402
import qp_zoo  # noqa: E402

prob = qp_zoo.equality_qp(n=6000)
m = oq.Model(oq.load_library())
t = time.time()
oq.setup(m, linsys_solver="direct", verbose=False, eps_abs=1e-4, eps_rel=1e-4, max_iter=4000, adaptive_rho_interval=50,
         check_termination=25, polish=False, **prob)
ts = time.time() - t
t = time.time()
r = oq.solve(m)
tt = time.time() - t
st = oq.stats(m)
print("setup %.3f s, solve %.4f s, %d iterations, %s, nnz(L) %d, levels %d" % (ts, tt, r.info.iter, r.info.status, st[4], st[5]))
oq.clean(m)
