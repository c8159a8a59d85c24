cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02k
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r02k/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r02k/pytest.log
for cfg in "1 0" "0 0"; do set -- $cfg; OSQP_AMD_PCG_FUSED=$1 OSQP_AMD_PCG_ASYNC=$2 timeout 300 python bench.py --workload rand-1e5 --steps 200 --warmup 25 --no-cpu --traffic off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused=$1 async=$2', d['value'], d['ms_per_step'], d['time_to_eps_s'], d['cg_iters_per_admm_iter'], d['pri_res'])"; done
timeout 300 python bench.py --workload rand-1e5 --steps 20 --warmup 5 --no-cpu --traffic off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver protocol', d['value'], d['ms_per_step'], d['time_to_eps_s'], d['cg_iters_per_admm_iter'])"
