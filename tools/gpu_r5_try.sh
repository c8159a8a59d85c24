cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_try; mkdir -p $O
for sr in 0 1 0 1 0 1; do
  OSQP_AMD_PCG_SR=$sr timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu --traffic off > $O/b.json 2>/dev/null
  python - $O/b.json "$sr rand-1e6 k20w5" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("sr", sys.argv[2], "it/s", d.get("value"), "ms", d.get("ms_per_step"), "cg/it", d.get("cg_iters_per_admm_iter"), "to_eps", d.get("time_to_eps_s"), "iters", d.get("iters_to_eps"), "cg_to_eps", d.get("cg_iters_to_eps"), d.get("status"))
PY
done
