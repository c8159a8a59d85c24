cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_tree; mkdir -p $O
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0
timeout 1500 python -m pytest tests/test_problem_zoo.py tests/test_multifrontal_gpu.py tests/test_fuzz_gpu.py -m gpu -q 2>&1 | tail -3
for pe in 1 0; do
OSQP_AMD_SNODE_TREE_PERSIST=$pe timeout 900 python bench.py --workload control-1e6 --steps 100 --warmup 25 --no-cpu --traffic off 2>/dev/null > $O/p_$pe.json
python - $O/p_$pe.json $pe <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("persist", sys.argv[2], d["value"], d["ms_per_step"], d["iters_to_eps"], d["time_to_eps_s"], d["roofline"]["frac"])
PY
done
timeout 900 python -m pytest tests/test_full_size_gpu.py -k "control" -m gpu -q 2>&1 | tail -2
ZOO_LABELS=gpu_direct timeout 600 python tools/zoo_rates.py control 2>/dev/null | cut -c1-200
