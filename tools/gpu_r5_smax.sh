# control-1e6: largest supernode (OSQP_AMD_SNODE_MAX) against rate / setup / refactor: the inverted diagonal blocks are 0.77 GB of the 2.0 GB an iteration streams
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_smax; mkdir -p $O
for s in 16 24 32 48 64; do
  OSQP_AMD_SNODE_MAX=$s timeout 600 python bench.py --workload control-1e6 --steps 100 --warmup 25 --no-cpu --traffic off > $O/b_$s.json 2>/dev/null
  python - $O/b_$s.json $s <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get("roofline") or {}
print("smax", sys.argv[2], "it/s", d.get("value"), "ms", d.get("ms_per_step"), "frac", r.get("frac"), "alg_bytes", r.get("algorithmic_bytes_per_launch"), "setup", d.get("setup_s"), "to_eps", d.get("time_to_eps_s"), "iters", d.get("iters_to_eps"))
PY
done
