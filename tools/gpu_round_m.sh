cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_fuzz_gpu.py -k supernodal -x -q 2>&1 | tail -5
timeout 600 python -m pytest tests/test_problem_zoo.py -m gpu -x -q 2>&1 | tail -5
ZOO_LABELS=gpu_direct timeout 300 python tools/zoo_rates.py control 2>/dev/null
OSQP_AMD_SNODE_TREE=0 ZOO_LABELS=gpu_direct timeout 300 python tools/zoo_rates.py control 2>/dev/null
cd /tmp && ZOO_LABELS=gpu_direct timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_m -o ctl -- python $GRAFT_REPO_ROOT/tools/zoo_rates.py control > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python tools/rocpd_summary.py $(find /tmp/prof_m -name '*_results.db' | head -1) > gpurun_out/m_control_prof.txt 2>&1; head -14 gpurun_out/m_control_prof.txt
