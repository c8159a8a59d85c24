cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_tree; mkdir -p $O
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0
for pe in 0 3 2; do
OSQP_AMD_SNODE_TREE_PERSIST=$pe timeout 900 python bench.py --workload control-1e6 --steps 100 --warmup 25 --no-cpu --traffic off 2>/dev/null > $O/q_$pe.json
python - $O/q_$pe.json $pe <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("persist from", sys.argv[2], d["value"], d["ms_per_step"], d["iters_to_eps"], d["time_to_eps_s"], d["roofline"]["frac"])
PY
done
