#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3as; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
cat > /tmp/setup_only.py <<'PY'
import sys, time
sys.path.insert(0, sys.argv[1])
import osqp_jl_amd as oq, bench
lib = oq.load_library()
for k in range(2):
    m = oq.Model(lib); t0=time.time(); oq.setup_generated(m, 0, 1000000, 1000, 1, linsys_solver="pcg", **bench.SETTINGS); t1=time.time()
    r = oq.solve(m); print("setup wall %.3f s; solve: %s iter %d pri %.12e dua %.12e" % (t1-t0, r.info.status, r.info.iter, r.info.pri_res, r.info.dua_res), flush=True); oq.clean(m)
PY
for g in 1 0; do echo "== OSQP_AMD_SELL_FILL=$g"; OSQP_AMD_SELL_FILL=$g OSQP_AMD_SETUP_TRACE=1 timeout 300 python /tmp/setup_only.py $GRAFT_REPO_ROOT 2>&1 | grep -E "slices of|Ruiz|setup wall" | tail -5 | cut -c1-150; done | tee $O/setup_trace.txt
