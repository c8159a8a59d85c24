# kernel profile of control-1e6 (per-kernel times of an iteration) after a solve-kernel change; VARIANT = environment assignments
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_prof; mkdir -p $O
for v in $VARIANTS; do
  env $v rocprofv3 --kernel-trace --stats -d $O/prof_c -o p -- python $GRAFT_REPO_ROOT/bench.py --workload control-1e6 --steps 100 --warmup 25 --no-cpu --traffic off > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find $O/prof_c -name '*_results.db' | head -1) > $O/kernel_stats_control-1e6_$v.md
  rm -rf $O/prof_c
  echo "== $v"; head -16 $O/kernel_stats_control-1e6_$v.md | cut -c20-150
done
