# round 6: the direct back-end on a 2-D grid QP at full size: setup trace, bench line, kernel stats
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_grid; mkdir -p $O
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0
for w in grid2d-5e5 grid2d-1e6; do
  OSQP_AMD_SETUP_TRACE=1 timeout 900 python bench.py --workload $w --no-cpu --traffic off --steps 100 --warmup 25 > $O/bench_$w.json 2> $O/setup_trace_$w.txt
  python - $O/bench_$w.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "time_to_eps_s", "iters_to_eps", "status", "setup_s", "rho_updates")}, d["config"]["backend"], d["roofline"]["frac"], d["roofline"]["step"]["frac"])
PY
  grep -E "fronts\]|lean|symbolic analysis|numeric|ordering|total" $O/setup_trace_$w.txt | head -40
done
