# control-1e6 rate against the count from which a level takes the wavefront form
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_try; mkdir -p $O
for k in 16384 8192 4096 1500 400; do
  OSQP_AMD_SNODE_WAVE_MIN=$k timeout 600 python bench.py --workload control-1e6 --steps 100 --warmup 25 --no-cpu --traffic off > $O/b_$k.json 2>/dev/null
  python - $O/b_$k.json $k <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get("roofline") or {}
print("wave_min", sys.argv[2], "it/s", d.get("value"), "ms", d.get("ms_per_step"), "frac", r.get("frac"), "setup", d.get("setup_s"), "to_eps", d.get("time_to_eps_s"), "iters", d.get("iters_to_eps"), d.get("status"))
PY
done
