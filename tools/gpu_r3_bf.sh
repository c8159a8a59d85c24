#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3bf; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
OSQP_AMD_SKIP_RAND1E6=1 OSQP_AMD_POISON=1 timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_full_size_gpu.py > $O/pytest_poison.log 2>&1; echo "poison rc=$?"; tail -3 $O/pytest_poison.log
cat > /tmp/setup_only.py <<'PY'
import sys, time
sys.path.insert(0, sys.argv[1])
import osqp_jl_amd as oq, bench
lib = oq.load_library()
for k in range(2):
    m = oq.Model(lib); t0=time.time(); oq.setup_generated(m, 0, 1000000, 1000, 1, linsys_solver="pcg", **bench.SETTINGS); t1=time.time()
    r = oq.solve(m); st = oq.stats(m); print("setup wall %.3f s; peak %.2f GB resident %.2f; solve: %s iter %d pri %.12e dua %.12e" % (t1-t0, st[20]/1e9, st[9]/1e9, r.info.status, r.info.iter, r.info.pri_res, r.info.dua_res), flush=True); oq.clean(m)
PY
OSQP_AMD_SETUP_TRACE=1 timeout 300 python /tmp/setup_only.py $GRAFT_REPO_ROOT 2>&1 | grep -E "setup" | tee $O/setup_trace.txt | tail -12 | cut -c1-170
bash tools/gpu_round_big.sh 2>/dev/null | tail -1 | tee $O/largest_instance.json
