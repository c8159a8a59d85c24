cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python $GRAFT_REPO_ROOT/tools/batch_shapes.py 4096 > $O/batch_shapes.jsonl 2> $O/batch_shapes.err
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof -name '*_results.db' | head -1)
python tools/rocpd_dispatches.py $DB k_batch 40 > $O/batch_dispatches.txt
rm -rf $O/prof
cat $O/batch_shapes.jsonl; cat $O/batch_dispatches.txt
