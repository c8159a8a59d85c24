cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_try; mkdir -p $O
for c in 1 2 1 2 1 2; do
  OSQP_AMD_SNODE_FOLD=$c timeout 600 python bench.py --workload control-1e6 --steps 100 --warmup 25 --no-cpu --traffic off > $O/b.json 2>/dev/null
  python - $O/b.json "$c" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get("roofline") or {}
print("fold", sys.argv[2], "it/s", d.get("value"), "ms", d.get("ms_per_step"), "frac", r.get("frac"), "to_eps", d.get("time_to_eps_s"), "iters", d.get("iters_to_eps"), d.get("status"))
PY
done
