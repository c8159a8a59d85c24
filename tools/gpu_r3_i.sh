# round 3, call I: gpu suite (with the MOI conformance subset), setup trace after the reordering, host allowance
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3i
echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  nproc: $(nproc)  affinity: $(python -c 'import os; print(len(os.sched_getaffinity(0)))')"
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3i/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3i/pytest.log
OSQP_AMD_SETUP_TRACE=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --traffic off > gpurun_out/r3i/bench_rand1e6_k20w5.json 2> gpurun_out/r3i/setup_trace_rand1e6.txt; echo "bench rc=$?"
grep "\[setup\]" gpurun_out/r3i/setup_trace_rand1e6.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3i/bench_rand1e6_k20w5.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','time_to_eps_s','iters_to_eps','setup_s','device_gb','device_peak_gb','run_time_s','iterations_per_s_incl_setup')})
PY
