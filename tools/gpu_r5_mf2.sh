cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_mf2; mkdir -p $O
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0
timeout 1500 python -m pytest tests/test_problem_zoo.py tests/test_multifrontal_gpu.py tests/test_fuzz_gpu.py -m gpu -q 2>&1 | tail -3
timeout 600 python tools/refactor_time.py 800 8000 2>&1 | grep "T=" | tee $O/refactor_time.txt
timeout 900 python bench.py --workload control-1e6 --steps 20 --warmup 5 --no-cpu --traffic off 2>/dev/null > $O/bench_control1e6.json
python - $O/bench_control1e6.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "setup_s", "time_to_eps_s", "iters_to_eps", "iterations_per_s_incl_setup")}, d["roofline"]["frac"])
PY
