cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_eq; mkdir -p $O
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0 ZOO_LABELS=gpu_direct
timeout 1500 python -m pytest tests/test_problem_zoo.py tests/test_gpu_parity.py tests/test_multifrontal_gpu.py tests/test_fuzz_gpu.py -m gpu -q 2>&1 | tail -3
timeout 600 python tools/refactor_time.py 800 8000 2>&1 | grep "T=" | tee $O/refactor_time.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python $GRAFT_REPO_ROOT/tools/zoo_rates.py equality_qp > $O/zoo_eq.jsonl 2>/dev/null
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find $O/prof -name '*_results.db' | head -1) > $O/kernel_stats_equality_qp.md
rm -rf $O/prof
cut -c1-260 $O/zoo_eq.jsonl; head -8 $O/kernel_stats_equality_qp.md | cut -c1-150
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof2 -o p -- python $GRAFT_REPO_ROOT/bench.py --workload control-1e6 --steps 100 --warmup 25 --no-cpu --traffic off > $O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof2 -name '*_results.db' | head -1)
python tools/factor_timeline.py $DB > $O/factor_timeline_control1e6.txt
python tools/rocpd_dispatches.py $DB k_mf_front 20 > $O/mf_dispatches.txt
rm -rf $O/prof2
head -8 $O/factor_timeline_control1e6.txt; cat $O/mf_dispatches.txt | cut -c1-110
