set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest.log 2>&1; echo "pytest rc=$?" 
tail -15 gpurun_out/r02a/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r02a/bench_1e6.json 2> gpurun_out/r02a/bench_1e6.err; echo "bench rc=$?"
cat gpurun_out/r02a/bench_1e6.json | cut -c1-3000
tail -5 gpurun_out/r02a/bench_1e6.err
timeout 300 python bench.py --workload mpc-batch --steps 20 --warmup 3 > gpurun_out/r02a/bench_batch.json 2> gpurun_out/r02a/bench_batch.err; echo "batch rc=$?"
cat gpurun_out/r02a/bench_batch.json | cut -c1-3000
OSQP_AMD_BENCH_ONE_DEVICE=1 OSQP_AMD_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --workload rand-1e5 --steps 50 --warmup 10 > gpurun_out/r02a/bench_2rank.json 2> gpurun_out/r02a/bench_2rank.err; echo "2rank rc=$?"
cat gpurun_out/r02a/bench_2rank.json | cut -c1-4000
tail -3 gpurun_out/r02a/bench_2rank.err
