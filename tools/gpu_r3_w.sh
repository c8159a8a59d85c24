#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3w; mkdir -p $O
cat > /tmp/vmm_case.py <<'PY'
import sys, os
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import numpy as np, scipy.sparse as sp, ctypes as C
import osqp_jl_amd as oq
from conftest import *  # noqa
import importlib
tg = importlib.import_module("test_gpu_parity")
lib = oq.load_library()
from osqp_jl_amd import types as T
orc = oq.load_library(os.path.join(sys.argv[1], "oracle", "_build", "libosqp_oracle.so"))
n, k = 40000, 96
d = orc.oracle_generate(0, n, k, 21)
P, q, A, l, u = tg._data_to_scipy(d.contents)
orc.oracle_data_free(d)
m = oq.Model(lib)
oq.setup_generated(m, 0, n, k, 21, scaling=0, verbose=False, linsys_solver="pcg")
rng = np.random.default_rng(5)
xv, yv = rng.standard_normal(n), rng.standard_normal(n)
Pfull = P + sp.triu(P, 1).T
def check(tag, mats):
    for op, mat, vec in mats:
        out = np.zeros(n)
        rc = lib.osqp_amd_apply(m.workspace, op, oq.interface._fptr(vec), oq.interface._fptr(out))
        ref = mat @ vec
        print(tag, "op", op, "rc", rc, "err %.3e of %.3e" % (np.max(np.abs(out - ref)), np.max(np.abs(ref))), flush=True)
check("setup ", ((0, A, xv), (1, A.T, yv), (2, Pfull, xv)))
A2 = A.copy(); A2.data = A2.data * 1.5
Pu = sp.triu(P, format="csc"); Pu2 = Pu.copy(); Pu2.data = Pu2.data * 0.5
oq.update(m, Px=Pu2.data, Ax=A2.data)
P2full = Pu2 + sp.triu(Pu2, 1).T
check("update", ((0, A2, xv), (1, A2.T, yv), (2, P2full, xv)))
PY
for cfg in "OSQP_AMD_VMM=0" "OSQP_AMD_VMM_MIN_MB=1" "OSQP_AMD_VMM_MIN_MB=1 OSQP_AMD_POISON=1" "OSQP_AMD_VMM_MIN_MB=64" "OSQP_AMD_VMM_MIN_MB=1 OSQP_AMD_DEBUG=1"; do
  for pg in "2 1" "3 1"; do set -- $pg
  echo "== $cfg  panel=$1 group=$2" | tee -a $O/vmm_case.txt
  env $cfg OSQP_AMD_PANEL=$1 OSQP_AMD_PANEL_GROUP=$2 timeout 120 python /tmp/vmm_case.py $GRAFT_REPO_ROOT 2>&1 | grep -v amdgpu.ids | tail -8 | tee -a $O/vmm_case.txt
  done
done
