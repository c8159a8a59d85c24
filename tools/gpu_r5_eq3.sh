cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_eq; mkdir -p $O
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0 ZOO_LABELS=gpu_direct
timeout 1500 python -m pytest tests/test_problem_zoo.py tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -3
for sd in 1 0; do OSQP_AMD_SCHUR_DENSE=$sd timeout 600 python tools/zoo_rates.py equality_qp portfolio lasso_data svm 2>/dev/null | cut -c1-250; done
cd /tmp
OSQP_AMD_SETUP_TRACE=1 timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python $GRAFT_REPO_ROOT/tools/zoo_rates.py equality_qp > $O/zoo_eq.jsonl 2> $O/setup_trace_equality_qp.txt
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find $O/prof -name '*_results.db' | head -1) > $O/kernel_stats_equality_qp.md
rm -rf $O/prof
grep "numeric\|symbolic analysis\|back-end" $O/setup_trace_equality_qp.txt | cut -c1-90; head -9 $O/kernel_stats_equality_qp.md | cut -c1-150
