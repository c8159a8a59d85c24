cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_quick; mkdir -p $O
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0
timeout 1500 python -m pytest tests/test_multifrontal_gpu.py tests/test_radix_transpose_gpu.py tests/test_gpu_parity.py tests/test_problem_zoo.py -m gpu -q > $O/pytest.log 2>&1
grep -E "passed|failed|FAILED|Error" $O/pytest.log | head -20
OSQP_AMD_SETUP_TRACE=1 OSQP_AMD_SYMBOLIC_TRACE=1 timeout 900 python bench.py --workload control-1e6 --steps 20 --warmup 5 --no-cpu --traffic off 2> $O/setup_trace_control1e6.txt > $O/bench_control1e6.json
grep -v "fronts\]\|amdgpu.ids" $O/setup_trace_control1e6.txt | tail -26 | cut -c1-100
python - $O/bench_control1e6.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "setup_s", "time_to_eps_s", "iters_to_eps", "iterations_per_s_incl_setup")}, d["roofline"]["frac"])
PY
timeout 600 python tools/zoo_rates.py 2>/dev/null | tail -12 | cut -c1-220
