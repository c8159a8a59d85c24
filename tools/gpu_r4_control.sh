# control-1e6 at the end of round 4: bench line (live PMC, CPU oracle beside it), kernel stats, setup trace, factor timeline
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04_control; mkdir -p $O
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0
timeout 1200 python bench.py --workload control-1e6 > $O/bench_control1e6.json 2>/dev/null
OSQP_AMD_SETUP_TRACE=1 OSQP_AMD_SYMBOLIC_TRACE=1 timeout 900 python bench.py --workload control-1e6 --steps 20 --warmup 5 --no-cpu --traffic off 2> $O/setup_trace_control1e6.txt > /dev/null
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python $GRAFT_REPO_ROOT/bench.py --workload control-1e6 --steps 100 --warmup 25 --no-cpu --traffic off > $O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $O/prof -name '*_results.db' | head -1)
python tools/rocpd_summary.py $DB > $O/kernel_stats_control-1e6.md
python tools/factor_timeline.py $DB > $O/factor_timeline_control1e6.txt
rm -rf $O/prof
python - $O/bench_control1e6.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]; c = d.get("cpu_baseline", {})
print(d["value"], d["ms_per_step"], "frac", r["frac"], "traffic", r.get("traffic"), "setup", d["setup_s"], "to_eps", d["time_to_eps_s"], "cpu", c.get("value"))
PY
head -14 $O/kernel_stats_control-1e6.md | cut -c1-150; head -3 $O/factor_timeline_control1e6.txt
