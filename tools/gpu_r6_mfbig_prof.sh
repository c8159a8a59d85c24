cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_mfbig; mkdir -p $O
export TMPDIR=/tmp REFACTOR_GRID=1 OSQP_AMD_MF_BIG=1
cd /tmp; rm -rf /tmp/prof_rf
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_rf -o p -- python $GRAFT_REPO_ROOT/tools/refactor_time.py --child ${1:-700} > $O/rf.log 2>&1
tail -1 $O/rf.log
DB=$(find /tmp/prof_rf -name '*_results.db' | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB | head -16 | cut -c1-160 | tee $O/refactor_kernel_stats_grid${1:-700}.md
python $GRAFT_REPO_ROOT/tools/factor_timeline.py $DB 2>&1 | tail -12 | tee $O/factor_timeline_grid${1:-700}.txt
python $GRAFT_REPO_ROOT/tools/rocpd_dispatches.py $DB k_mf 400 2>/dev/null | tail -130 | cut -c1-150 > $O/front_dispatches_grid${1:-700}.txt
