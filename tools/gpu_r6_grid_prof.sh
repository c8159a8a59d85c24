# round 6: kernel-level view of the grid workloads (rocprofv3 --kernel-trace --stats)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_grid; mkdir -p $O
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0 OSQP_AMD_BENCH_OTHERS=0
cd /tmp
for w in ${1:-grid2d-5e5}; do
  rm -rf /tmp/prof_$w
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $w --no-cpu --traffic off --steps 100 --warmup 25 > $O/bench_prof_$w.json 2> $O/prof_$w.err
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/prof_$w -name '*_results.db' | head -1) > $O/kernel_stats_$w.md 2>&1
  head -14 $O/kernel_stats_$w.md | cut -c1-200
done
