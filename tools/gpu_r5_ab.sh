cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_ab; mkdir -p $O
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0
for ord in 1 0; do
OSQP_AMD_SNODE_ORDER=$ord timeout 900 python bench.py --workload control-1e6 --steps 100 --warmup 25 --no-cpu --traffic off 2>/dev/null > $O/bench_control1e6_order$ord.json
python - $O/bench_control1e6_order$ord.json $ord <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("order", sys.argv[2], {k: d.get(k) for k in ("value", "ms_per_step", "setup_s", "time_to_eps_s", "iters_to_eps")}, d["roofline"]["frac"])
PY
done
timeout 900 python -m pytest tests/test_multifrontal_gpu.py tests/test_problem_zoo.py tests/test_fuzz_gpu.py -m gpu -q 2>&1 | tail -2
