cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3l
timeout 900 python -m pytest tests/test_batch_gpu.py tests/test_full_size_gpu.py tests/test_gpu_parity.py -m gpu -x -q -k "batch" > gpurun_out/r3l/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3l/pytest.log
OSQP_AMD_LIB=$GRAFT_REPO_ROOT/osqp.jl_amd/csrc/libosqp_amd_prof.so timeout 120 python bench.py --workload mpc-batch --steps 2 --warmup 1 --no-cpu --traffic off 2>&1 | grep -v '^{' | tail -3
timeout 300 python bench.py --workload mpc-batch --steps 20 --warmup 3 --no-cpu --traffic off 2>/dev/null | cut -c1-200
