# round 6: kernel view of one refactorisation of the grid with the dense top (rocprofv3 --kernel-trace --stats)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_snd; mkdir -p $O
export TMPDIR=/tmp REFACTOR_GRID=1
cd /tmp; rm -rf /tmp/prof_rf
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_rf -o p -- python $GRAFT_REPO_ROOT/tools/refactor_time.py --child ${1:-700} > $O/rf.log 2>&1
tail -1 $O/rf.log
DB=$(find /tmp/prof_rf -name '*_results.db' | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB | head -24 | cut -c1-170 | tee $O/refactor_kernel_stats_grid${1:-700}.md
python $GRAFT_REPO_ROOT/tools/factor_timeline.py $DB 2>&1 | tail -14 | tee $O/factor_timeline_grid${1:-700}.txt
