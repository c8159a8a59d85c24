# round 5, first GPU call: the multifrontal factorisation -- parity tests, then control-1e6 with setup trace and factor timeline
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_mf; mkdir -p $O
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0
timeout 900 python -m pytest tests/test_multifrontal_gpu.py tests/test_fuzz_gpu.py tests/test_problem_zoo.py tests/test_sharded_gpu.py -m gpu -x -q > $O/pytest.log 2>&1
tail -15 $O/pytest.log
OSQP_AMD_SETUP_TRACE=1 OSQP_AMD_SYMBOLIC_TRACE=1 timeout 900 python bench.py --workload control-1e6 --steps 20 --warmup 5 --no-cpu --traffic off 2> $O/setup_trace_control1e6.txt > $O/bench_control1e6.json
tail -3 $O/bench_control1e6.json | cut -c1-600
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python $GRAFT_REPO_ROOT/bench.py --workload control-1e6 --steps 100 --warmup 25 --no-cpu --traffic off > $O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof -name '*_results.db' | head -1)
python tools/rocpd_summary.py $DB > $O/kernel_stats_control-1e6.md
python tools/factor_timeline.py $DB > $O/factor_timeline_control1e6.txt
python tools/rocpd_dispatches.py $DB k_mf_front 44 > $O/mf_dispatches.txt; cat $O/mf_dispatches.txt | head -30
rm -rf $O/prof
head -30 $O/kernel_stats_control-1e6.md | cut -c1-160; head -12 $O/factor_timeline_control1e6.txt
