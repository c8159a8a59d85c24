# quick A/B: supernode tests with the variant, control-1e6 rate / refactorisation by variant
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_try; mkdir -p $O
for k in 1 0 2 1 0 2; do
  OSQP_AMD_SNODE_LEAF=$k timeout 600 python bench.py --workload control-1e6 --steps 100 --warmup 25 --no-cpu --traffic off > $O/b_$k.json 2>/dev/null
  python - $O/b_$k.json $k <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get("roofline") or {}
print("leaf", sys.argv[2], "it/s", d.get("value"), "ms", d.get("ms_per_step"), "frac", r.get("frac"), "alg", r.get("algorithmic_bytes_per_launch"), "setup", d.get("setup_s"), "to_eps", d.get("time_to_eps_s"), "iters", d.get("iters_to_eps"), d.get("status"))
PY
done
for k in 0 1 2; do OSQP_AMD_SNODE_LEAF=$k timeout 600 python tools/refactor_time.py 800 8000 2>&1 | grep "T=" | grep "mf=1" | sed "s/^/leaf $k /"; done
for k in 1 2; do OSQP_AMD_SNODE_LEAF=$k timeout 1500 python -m pytest tests/test_problem_zoo.py tests/test_multifrontal_gpu.py tests/test_fuzz_gpu.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3; done
