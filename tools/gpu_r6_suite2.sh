# round 6 (second session): the whole GPU suite + kernel view of the grid iteration
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_suite2; mkdir -p $O
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0 OSQP_AMD_BENCH_OTHERS=0
timeout 3000 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.log 2>&1
tail -8 $O/pytest_gpu.log
cd /tmp
for w in grid2d-5e5; do
  rm -rf /tmp/prof_$w
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $w --no-cpu --traffic off --steps 100 --warmup 25 > $O/bench_prof_$w.json 2> $O/prof_$w.err
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/prof_$w -name '*_results.db' | head -1) > $O/kernel_stats_$w.md 2>&1
  head -24 $O/kernel_stats_$w.md | cut -c1-200
done
