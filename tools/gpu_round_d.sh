set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02d
timeout 600 python -m pytest tests/test_batch_gpu.py -x -q > gpurun_out/r02d/pytest_batch.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r02d/pytest_batch.log
timeout 300 python bench.py --workload mpc-batch --steps 20 --warmup 3 --no-cpu > gpurun_out/r02d/bench_batch.json 2> gpurun_out/r02d/bench_batch.err; echo "batch rc=$?"
cat gpurun_out/r02d/bench_batch.json | cut -c1-1500
OSQP_AMD_LIB=$GRAFT_REPO_ROOT/osqp.jl_amd/csrc/libosqp_amd_prof.so timeout 120 python bench.py --workload mpc-batch --steps 2 --warmup 1 --no-cpu 2>&1 | grep -v '^{' | tail -3
