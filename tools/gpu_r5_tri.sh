# round 5: triangular solves by supernodes with coalesced block products -- tests, control-1e6 rate + kernel table; batch shapes with the NH = 50 entry
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_tri; mkdir -p $O
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0
timeout 1500 python -m pytest tests/test_multifrontal_gpu.py tests/test_fuzz_gpu.py tests/test_problem_zoo.py tests/test_batch_gpu.py -m gpu -q > $O/pytest.log 2>&1
grep -E "passed|failed|FAILED|Error" $O/pytest.log | head -20
timeout 900 python bench.py --workload control-1e6 --steps 100 --warmup 25 --no-cpu --traffic off 2> /dev/null > $O/bench_control1e6.json
python - $O/bench_control1e6.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "setup_s", "time_to_eps_s", "iters_to_eps", "iterations_per_s_incl_setup")}, d["roofline"]["frac"])
PY
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python $GRAFT_REPO_ROOT/bench.py --workload control-1e6 --steps 100 --warmup 25 --no-cpu --traffic off > $O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof -name '*_results.db' | head -1)
python tools/rocpd_summary.py $DB > $O/kernel_stats_control-1e6.md
python tools/factor_timeline.py $DB > $O/factor_timeline_control1e6.txt
rm -rf $O/prof
head -24 $O/kernel_stats_control-1e6.md | cut -c1-150; head -8 $O/factor_timeline_control1e6.txt
timeout 600 python tools/batch_shapes.py 4096 2>/dev/null | cut -c1-200 > $O/batch_shapes_wall.jsonl; cat $O/batch_shapes_wall.jsonl
