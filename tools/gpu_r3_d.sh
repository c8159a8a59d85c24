# round 3, call D: the batched kernel after the hot-loop rewrite
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3d
timeout 900 python -m pytest tests/test_batch_gpu.py tests/test_full_size_gpu.py tests/test_gpu_parity.py -m gpu -x -q -k "batch" > gpurun_out/r3d/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3d/pytest.log
timeout 300 python bench.py --workload mpc-batch --steps 20 --warmup 3 --no-cpu --traffic off 2>/dev/null | cut -c1-330
