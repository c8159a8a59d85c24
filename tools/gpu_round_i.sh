cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02i
timeout 1200 python -m pytest tests -m gpu -x -q -k "not full_size and not sharded and not fuzz" > gpurun_out/r02i/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02i/pytest.log
for mp in 1 0; do OSQP_AMD_MAPPED_SLOTS=$mp timeout 300 python bench.py --workload rand-1e5 --steps 200 --warmup 25 --no-cpu --traffic off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mapped=$mp', d['value'], d['ms_per_step'], d['time_to_eps_s'], d['cg_iters_per_admm_iter'])"; done
OSQP_AMD_PCG_ASYNC=1 timeout 300 python bench.py --workload rand-1e5 --steps 200 --warmup 25 --no-cpu --traffic off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('async', d['value'], d['ms_per_step'], d['time_to_eps_s'])"
timeout 300 python bench.py --workload rand-1e5 --steps 20 --warmup 5 --no-cpu --traffic off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('driver protocol', d['value'], d['ms_per_step'], d['time_to_eps_s'], d['cg_iters_per_admm_iter'])"
timeout 300 python bench.py --workload lasso-5e5 --steps 100 --warmup 25 --no-cpu --traffic off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lasso', d['value'], d['ms_per_step'], d['roofline']['ms_per_launch'])"
