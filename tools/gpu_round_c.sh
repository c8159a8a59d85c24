set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
timeout 900 python -m pytest tests -m gpu -x -q -k "not full_size and not sharded" > gpurun_out/r02c/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r02c/pytest.log
timeout 300 python bench.py --workload lasso-5e5 --steps 100 --warmup 25 --no-cpu --traffic off > gpurun_out/r02c/bench_lasso.json 2> gpurun_out/r02c/bench_lasso.err; echo "lasso rc=$?"
cat gpurun_out/r02c/bench_lasso.json | cut -c1-2500
OSQP_AMD_FUSE2=0 timeout 300 python bench.py --workload lasso-5e5 --steps 100 --warmup 25 --no-cpu --traffic off 2>/dev/null | cut -c1-400
OSQP_AMD_MD_FIFO=0 timeout 300 python bench.py --workload lasso-5e5 --steps 100 --warmup 25 --no-cpu --traffic off 2>/dev/null | cut -c1-400
