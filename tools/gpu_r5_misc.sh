cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_misc; mkdir -p $O
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0
timeout 1500 python -m pytest tests/test_fuzz_gpu.py tests/test_multifrontal_gpu.py tests/test_problem_zoo.py -m gpu -q 2>&1 | tail -3
OSQP_AMD_BATCH_TRACE=1 timeout 300 python tools/batch_shapes.py 64 2>&1 >/dev/null | sort | uniq -c | grep "of 50" | cut -c1-160
timeout 600 python tools/refactor_time.py 800 8000 2>&1 | grep "mf=1"
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/prof_c -o p -- python $GRAFT_REPO_ROOT/bench.py --workload control-1e6 --steps 20 --warmup 5 --no-cpu --traffic off > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/factor_timeline.py $(find $O/prof_c -name '*_results.db' | head -1) > $O/factor_timeline_control1e6.txt
python $GRAFT_REPO_ROOT/tools/rocpd_dispatches.py $(find $O/prof_c -name '*_results.db' | head -1) k_mf_front 20 > $O/mf_dispatches.txt
rm -rf $O/prof_c
cd $GRAFT_REPO_ROOT
head -4 $O/factor_timeline_control1e6.txt; cut -c1-100 $O/mf_dispatches.txt
