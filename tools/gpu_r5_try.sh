cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_try; mkdir -p $O
timeout 1500 python -m pytest tests/test_multifrontal_gpu.py tests/test_full_size_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py --workload control-1e6 --steps 100 --warmup 25 --no-cpu --traffic off > $O/b.json 2>/dev/null
python - $O/b.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get("roofline") or {}
print("default it/s", d.get("value"), "ms", d.get("ms_per_step"), "frac", r.get("frac"), "alg", r.get("algorithmic_bytes_per_launch"), "setup", d.get("setup_s"), "to_eps", d.get("time_to_eps_s"), "iters", d.get("iters_to_eps"), d.get("status"))
PY
timeout 600 python tools/zoo_rates.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); g = d['gpu_direct']; print(d['problem'], g['status'], g['iter'], g['it_per_s'], 'setup', g['setup_s'])
"
