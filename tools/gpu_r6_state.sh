# round 6 (second session): where the tree stands -- grid / control / batch bench lines, grid setup trace + kernel stats
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_state; mkdir -p $O
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0 OSQP_AMD_BENCH_OTHERS=0
show() { python - $1 <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print(sys.argv[1].split("/")[-1], {k: d.get(k) for k in ("value", "ms_per_step", "time_to_eps_s", "iters_to_eps", "setup_s")}, "frac", r.get("frac"), "step", (r.get("step") or {}).get("frac"))
PY
}
for w in grid2d-5e5 control-1e6; do
  OSQP_AMD_SETUP_TRACE=1 OSQP_AMD_SYMBOLIC_TRACE=1 timeout 900 python bench.py --workload $w --no-cpu --traffic off --steps 100 --warmup 25 > $O/bench_$w.json 2> $O/setup_trace_$w.txt
  show $O/bench_$w.json
  grep -E "fronts\]|lean|symbolic|numeric|ordering|total|dissect|degree" $O/setup_trace_$w.txt | head -40 | cut -c1-220
done
timeout 600 python bench.py --workload mpc-batch --steps 20 --warmup 3 --no-cpu --traffic off > $O/bench_mpc_batch.json 2>/dev/null; show $O/bench_mpc_batch.json
(timeout 600 python tools/refactor_time.py 8000; REFACTOR_ONLY_BIG=1 timeout 600 python tools/refactor_time.py grid 700) 2>&1 | grep -E "T=" | tee $O/refactor_time.txt
cd /tmp
for w in grid2d-5e5 control-1e6; do
  rm -rf /tmp/prof_$w
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $w --no-cpu --traffic off --steps 100 --warmup 25 > /dev/null 2> $O/prof_$w.err
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/prof_$w -name '*_results.db' | head -1) > $O/kernel_stats_$w.md 2>&1
  python $GRAFT_REPO_ROOT/tools/factor_timeline.py $(find /tmp/prof_$w -name '*_results.db' | head -1) > $O/factor_timeline_$w.txt 2>&1
  head -30 $O/kernel_stats_$w.md | cut -c1-200
done
