"""The two ways the engine builds A = transpose(A') and the full symmetric P on the device (csrc/kernels.hip,
csr_from_coo): counting sort with row counters and a per-row sort, and -- for large matrices -- the stable radix sort
that needs neither device-scope atomics nor the per-row sort when the rows arrive column-ascending.  Forced on for every
size (OSQP_AMD_RADIX_MIN=1) it has to give bit for bit what the counting path gives (OSQP_AMD_RADIX_MIN=-1): on generated
problems (sorted input: no fallback), on a caller's upper triangle of P with unsorted row indices inside the columns (handed to the C ABI as they are: the check
finds unsorted rows of the full matrix and the per-row sort runs on top; unsorted columns of A are refused), with empty rows and columns, and through value updates by index
(the nnz-index maps come from the same permutation).  The switch is read when the library loads: child processes."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, os, hashlib
sys.path.insert(0, sys.argv[1])
import numpy as np, scipy.sparse as sp
import osqp_jl_amd as oq
from osqp_jl_amd import types as T
lib = oq.load_library()
class RawCcsc:  # the mirror's ManagedCcsc sorts the row indices of every column (Julia's SparseMatrixCSC is always sorted); a C caller need not
    def __init__(self, M):
        M = sp.csc_matrix(M)
        self.m, self.n = M.shape
        self.x = np.ascontiguousarray(M.data, dtype=np.float64)
        self.i = np.ascontiguousarray(M.indices, dtype=np.int64)
        self.p = np.ascontiguousarray(M.indptr, dtype=np.int64)
        self.ccsc = T.Ccsc(len(self.x), self.m, self.n, oq.interface._iptr(self.p), oq.interface._iptr(self.i), oq.interface._fptr(self.x), -1)
out = []
def digest(r):
    return hashlib.sha256(np.ascontiguousarray(r.x).tobytes() + np.ascontiguousarray(r.y).tobytes()).hexdigest()[:16] + ":%d:%s" % (r.info.iter, r.info.status)
opts = dict(verbose=False, eps_abs=1e-5, eps_rel=1e-5, adaptive_rho_interval=25, max_iter=400)
# 1. generated problems, both back-ends
for (n, k, solver) in ((20000, 48, "pcg"), (3000, 12, "qdldl")):
    m = oq.Model(lib); oq.setup_generated(m, 0, n, k, 5, linsys_solver=solver, **opts); out.append(digest(oq.solve(m))); oq.clean(m)
# 2. a caller's arrays: unsorted row indices inside the columns, empty rows of A, an empty column, duplicates absent
rng = np.random.default_rng(2)
n, mm = 4000, 6000
A = sp.random(mm, n, density=0.004, random_state=rng, data_rvs=rng.standard_normal, format="csc")
A = A.tolil(); A[100:140, :] = 0; A[:, 7] = 0; A = A.tocsc(); A.eliminate_zeros()
S = sp.random(n, n, density=0.002, random_state=rng, data_rvs=rng.standard_normal, format="csc")
P = sp.triu(sp.diags(1.0 + rng.random(n)) + 0.01 * (S + S.T), format="csc")
def shuffle_columns(M):
    M = M.copy()
    for j in range(M.shape[1]):
        a, b = M.indptr[j], M.indptr[j + 1]
        perm = rng.permutation(b - a)
        M.indices[a:b] = M.indices[a:b][perm]; M.data[a:b] = M.data[a:b][perm]
    M.has_sorted_indices = False
    return M
Au = shuffle_columns(A)
Ps = shuffle_columns(P)  # rows of the full symmetric P then arrive out of order: the check has to send them through the per-row sort
q = rng.standard_normal(n); l = -rng.random(mm); u = rng.random(mm)
ref = None
for solver in ("sorted", "pcg"):
    if solver == "pcg": oq.interface.ManagedCcsc = RawCcsc   # from here on the arrays reach the C ABI unsorted
    m = oq.Model(lib); oq.setup(m, P=(P if solver == "sorted" else Ps), q=q, A=A, l=l, u=u, linsys_solver="pcg", **opts)
    r0 = oq.solve(m)
    out.append(digest(r0))
    if ref is None: ref = r0.x.copy()
    else: assert np.max(np.abs(r0.x - ref)) <= 1e-3 * max(1.0, np.max(np.abs(ref))), "unsorted columns changed the solution"
    if solver == "sorted":
        out.append("skip:0:Solved"); oq.clean(m); continue
    idx = np.arange(0, A.nnz, 7)
    oq.update(m, Ax=A.data[idx] * 0.5, Ax_idx=idx)
    pidx = np.arange(0, Ps.nnz, 3)
    oq.update(m, Px=Ps.data[pidx] * 0.9, Px_idx=pidx)
    out.append(digest(oq.solve(m)))
    oq.clean(m)
# 3. unsorted columns of A (round 4): accepted as libosqp accepts them -- set up from a sorted host copy -- never mangled
m = oq.Model(lib)
try:
    oq.setup(m, P=P, q=q, A=Au, l=l, u=u, linsys_solver="pcg", **opts)
    ru = oq.solve(m)
    assert np.max(np.abs(ru.x - ref)) <= 1e-3 * max(1.0, np.max(np.abs(ref))), "unsorted columns of A changed the solution"
    out.append("unsorted-A:0:accepted")
except oq.OSQPError:
    out.append("unsorted-A:0:refused")
print("\n".join(out))
"""


def _run(extra_env):
    env = dict(os.environ)
    env.pop("OSQP_AMD_RADIX_MIN", None)
    env.update(extra_env)
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ":" in ln]
    assert len(lines) == 7, r.stdout
    return lines


def test_radix_path_equals_counting_path():
    counting = _run({"OSQP_AMD_RADIX_MIN": "-1"})
    radix = _run({"OSQP_AMD_RADIX_MIN": "1"})
    assert counting[0].endswith("Solved") and counting[2].endswith("Solved"), counting
    assert radix == counting
    assert counting[6].endswith("accepted")
