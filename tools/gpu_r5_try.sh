# per-kernel stats of one zoo class through the direct back-end
cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_zooprof; mkdir -p $O
for name in portfolio svm control; do
  ZOO_LABELS=gpu_direct rocprofv3 --kernel-trace --stats -d $O/p_$name -o p -- python $GRAFT_REPO_ROOT/tools/zoo_rates.py $name > $O/rate_$name.json 2>/dev/null
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find $O/p_$name -name '*_results.db' | head -1) > $O/kernel_stats_zoo_$name.md
  rm -rf $O/p_$name
  echo "== $name"; cat $O/rate_$name.json | cut -c1-300; head -14 $O/kernel_stats_zoo_$name.md | cut -c1-150
done
