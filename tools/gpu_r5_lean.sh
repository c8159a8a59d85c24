# round 5: lean setup (device-built factor arrays) -- tests, control-1e6 setup trace, refactor timing on the small control problems
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_lean; mkdir -p $O
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0
timeout 1200 python -m pytest tests/test_multifrontal_gpu.py tests/test_fuzz_gpu.py tests/test_problem_zoo.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1
tail -12 $O/pytest.log
for lean in 1 0; do
OSQP_AMD_LEAN=$lean OSQP_AMD_SETUP_TRACE=1 OSQP_AMD_SYMBOLIC_TRACE=1 timeout 900 python bench.py --workload control-1e6 --steps 20 --warmup 5 --no-cpu --traffic off 2> $O/setup_trace_control1e6_lean$lean.txt > $O/bench_control1e6_lean$lean.json
grep -v "fronts\]" $O/setup_trace_control1e6_lean$lean.txt | tail -32
python - $O/bench_control1e6_lean$lean.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "setup_s", "time_to_eps_s", "iters_to_eps", "iterations_per_s_incl_setup", "device_gb", "status")})
PY
done
timeout 600 python tools/refactor_time.py 800 8000 2>&1 | grep "T=" | tee $O/refactor_time.txt
