# quick A/B of a solve-kernel change: supernode tests, control-1e6 rate by variant
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_try; mkdir -p $O
timeout 1500 python -m pytest tests/test_problem_zoo.py tests/test_multifrontal_gpu.py tests/test_fuzz_gpu.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4
for k in 1 0 1; do
  OSQP_AMD_SNODE_FOLD=$k timeout 600 python bench.py --workload control-1e6 --steps 100 --warmup 25 --no-cpu --traffic off > $O/b_$k.json 2>/dev/null
  python - $O/b_$k.json $k <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get("roofline") or {}
print("fold", sys.argv[2], "it/s", d.get("value"), "ms", d.get("ms_per_step"), "frac", r.get("frac"), "setup", d.get("setup_s"), "to_eps", d.get("time_to_eps_s"), "iters", d.get("iters_to_eps"), d.get("status"))
PY
done
timeout 600 python tools/refactor_time.py 800 2>&1 | grep "T=" | grep "mf=1"
