# round 5: the four-wavefront batched kernel on generic patterns -- tests, the MPC bench line, then kernel times per shape
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_batch; mkdir -p $O
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0
timeout 1500 python -m pytest tests/test_batch_gpu.py -m gpu -q > $O/pytest.log 2>&1
grep -E "passed|failed|FAILED|Error" $O/pytest.log | head -30
timeout 600 python bench.py --workload mpc-batch --no-cpu --traffic off > $O/bench_mpc_batch.json 2>/dev/null
python - $O/bench_mpc_batch.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "unit", "ms_per_step")}, d.get("roofline", {}).get("frac"))
PY
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python $GRAFT_REPO_ROOT/tools/batch_shapes.py 4096 > $O/batch_shapes.jsonl 2> $O/batch_shapes.err
OSQP_AMD_BATCH_QUAD=0 timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof0 -o p -- python $GRAFT_REPO_ROOT/tools/batch_shapes.py 4096 > $O/batch_shapes_512thread.jsonl 2>> $O/batch_shapes.err
cd $GRAFT_REPO_ROOT
python tools/rocpd_dispatches.py $(find $O/prof -name '*_results.db' | head -1) k_batch 40 > $O/batch_dispatches.txt
python tools/rocpd_dispatches.py $(find $O/prof0 -name '*_results.db' | head -1) k_batch 40 > $O/batch_dispatches_512thread.txt
rm -rf $O/prof $O/prof0
cut -c1-230 $O/batch_shapes.jsonl; cut -c1-130 $O/batch_dispatches.txt; cut -c1-130 $O/batch_dispatches_512thread.txt
