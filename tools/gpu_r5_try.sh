cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_try; mkdir -p $O
for r in 0 1 2 3 0 1 2 3; do
  OSQP_AMD_BATCH_ROT=$r timeout 600 python bench.py --workload mpc-batch --steps 20 --warmup 3 --no-cpu --traffic off > $O/b.json 2>/dev/null
  python - $O/b.json "$r" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("rot", sys.argv[2], "ms", d.get("ms_per_step"), "value", d.get("value"), d.get("status"))
PY
done
OSQP_AMD_BATCH_ROT=2 timeout 900 python -m pytest tests/test_batch_gpu.py -m gpu -x -q 2>&1 | tail -2
