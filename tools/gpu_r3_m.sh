cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3m
timeout 900 python -m pytest tests/test_batch_gpu.py tests/test_full_size_gpu.py tests/test_gpu_parity.py -m gpu -x -q -k "batch" > gpurun_out/r3m/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3m/pytest.log
timeout 300 python bench.py --workload mpc-batch --steps 20 --warmup 3 --no-cpu --traffic off 2>/dev/null | cut -c1-200
