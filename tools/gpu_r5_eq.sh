cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_eq; mkdir -p $O
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0 ZOO_LABELS=gpu_direct
timeout 1500 python -m pytest tests/test_problem_zoo.py tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -3
OSQP_AMD_SETUP_TRACE=1 OSQP_AMD_SYMBOLIC_TRACE=1 timeout 600 python tools/zoo_rates.py equality_qp 2> $O/setup_trace_equality_qp.txt | cut -c1-300
grep -v amdgpu.ids $O/setup_trace_equality_qp.txt | cut -c1-100 | head -60
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python $GRAFT_REPO_ROOT/tools/zoo_rates.py equality_qp > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find $O/prof -name '*_results.db' | head -1) > $O/kernel_stats_equality_qp.md
rm -rf $O/prof
head -14 $O/kernel_stats_equality_qp.md | cut -c1-150
