# round 3, call A: the whole gpu suite on the new setup path, a setup trace of rand-1e6, bench lines
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3a
timeout 1500 python -m pytest tests -m gpu -x -q -s > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r3a/pytest.log
grep -E "worst over the batch|rand-1e6 on host" gpurun_out/r3a/pytest.log
OSQP_AMD_SETUP_TRACE=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --traffic off > gpurun_out/r3a/bench_rand1e6_k20w5.json 2> gpurun_out/r3a/setup_trace_rand1e6.txt; echo "bench rc=$?"
cat gpurun_out/r3a/setup_trace_rand1e6.txt | grep "\[setup\]"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3a/bench_rand1e6_k20w5.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','time_to_eps_s','iters_to_eps','setup_s','device_gb','device_peak_gb','cg_iters_per_admm_iter')})
PY
# a second setup in the same process is the warm number
python - <<'PY'
import time, osqp_jl_amd as oq, bench
lib = oq.load_library()
for k in range(2):
    m = oq.Model(lib); t0=time.time(); oq.setup_generated(m, 0, 1000000, 1000, 1, linsys_solver="pcg", **bench.SETTINGS); t=time.time()-t0
    st=oq.stats(m); print("setup #%d: %.3f s wall, setup_time %.3f s, device %.1f GB, peak %.1f GB" % (k, t, m.workspace.contents.info.contents.setup_time, st[9]/1e9, st[20]/1e9)); oq.clean(m)
PY
timeout 300 python bench.py --workload mpc-batch --steps 10 --warmup 2 --no-cpu > gpurun_out/r3a/bench_mpc.json 2>&1; tail -c 600 gpurun_out/r3a/bench_mpc.json
