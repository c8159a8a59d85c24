#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3ah; mkdir -p $O
timeout 900 python -m pytest tests/test_radix_transpose_gpu.py -q -x 2>&1 | tail -15
OSQP_AMD_RADIX_MIN=1 timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_full_size_gpu.py > $O/pytest_radix_everywhere.log 2>&1; echo "radix everywhere rc=$?"; tail -3 $O/pytest_radix_everywhere.log
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
cd /tmp; rm -rf /tmp/prof_s
cat > /tmp/setup_only.py <<'PY'
import sys, time
sys.path.insert(0, sys.argv[1])
import osqp_jl_amd as oq, bench
lib = oq.load_library()
m = oq.Model(lib); oq.setup_generated(m, 0, 1000000, 1000, 1, linsys_solver="pcg", **bench.SETTINGS); oq.clean(m)
PY
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o s -- python /tmp/setup_only.py $GRAFT_REPO_ROOT > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_s -name "*_results.db" | head -1) > $O/kernel_stats_setup.md; head -24 $O/kernel_stats_setup.md | cut -c1-150
bash tools/gpu_round_big.sh 2>/dev/null | tail -1 | tee $O/largest_instance.json
