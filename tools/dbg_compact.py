import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
os.environ["OSQP_AMD_PANEL"] = "2"
import numpy as np, scipy.sparse as sp
import osqp_jl_amd as oq
from test_gpu_parity import _data_to_scipy
lib = oq.load_library(); ora = oq.load_library(oq.ORACLE_LIB_PATH)
n, k = 40000, 96
d = ora.oracle_generate(0, n, k, 21); P, q, A, l, u = _data_to_scipy(d.contents); ora.oracle_data_free(d)
Pu = sp.triu(P, format="csc"); Pfull = Pu + sp.triu(Pu, 1).T
rng = np.random.default_rng(5); xv, yv = rng.standard_normal(n), rng.standard_normal(n)
f = oq.interface._fptr
for lim in ("-1", "0"):
    os.environ["OSQP_AMD_COMPACT_NNZ"] = lim
    m0 = oq.Model(lib); oq.setup_generated(m0, 0, n, k, 21, scaling=0, verbose=False, linsys_solver="pcg")
    print("compact", oq.stats(m0)[18])
    for rep in range(2):
        for op, mat, vec in ((0, A, xv), (1, A.T, yv), (2, Pfull, xv)):
            out = np.zeros(n); assert lib.osqp_amd_apply(m0.workspace, op, f(vec), f(out)) == 0
            ref = mat @ vec; bad = np.nonzero(np.abs(out - ref) > 1e-9)[0]
            print(" op", op, "bad rows", len(bad), bad[:10], (out - ref)[bad[:4]])
