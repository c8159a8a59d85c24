# round 3, call B: gpu suite after the slot-map fix, setup trace, kernel stats of the rand-1e6 bench command
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3b
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3b/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3b/pytest.log
OSQP_AMD_SETUP_TRACE=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --traffic off > gpurun_out/r3b/bench_rand1e6_k20w5.json 2> gpurun_out/r3b/setup_trace_rand1e6.txt; echo "bench rc=$?"
grep "\[setup\]" gpurun_out/r3b/setup_trace_rand1e6.txt
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3b/prof -o rand1e6 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu --traffic off > $GRAFT_REPO_ROOT/gpurun_out/r3b/prof_bench.json 2>/dev/null; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
ls gpurun_out/r3b/prof/* | head
python tools/rocpd_summary.py $(ls gpurun_out/r3b/prof/*/*_results.db | head -1) > gpurun_out/r3b/kernel_stats_rand-1e6.md 2>&1; head -30 gpurun_out/r3b/kernel_stats_rand-1e6.md | cut -c1-180
rm -rf gpurun_out/r3b/prof
