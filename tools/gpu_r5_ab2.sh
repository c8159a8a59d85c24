cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_ab2; mkdir -p $O
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0
for wm in 16384 8192 4096 1024; do
OSQP_AMD_SNODE_WAVE_MIN=$wm timeout 900 python bench.py --workload control-1e6 --steps 100 --warmup 25 --no-cpu --traffic off 2>/dev/null > $O/b_$wm.json
python - $O/b_$wm.json $wm <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("wave_min", sys.argv[2], d["value"], d["ms_per_step"], d["iters_to_eps"])
PY
done
for tr in 0; do
OSQP_AMD_SNODE_TREE=$tr timeout 900 python bench.py --workload control-1e6 --steps 100 --warmup 25 --no-cpu --traffic off 2>/dev/null > $O/t_$tr.json
python - $O/t_$tr.json $tr <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("tree", sys.argv[2], d["value"], d["ms_per_step"], d["iters_to_eps"])
PY
done
