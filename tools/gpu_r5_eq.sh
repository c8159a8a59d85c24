cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_eq; mkdir -p $O
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0
timeout 1500 python -m pytest tests/test_multifrontal_gpu.py tests/test_problem_zoo.py tests/test_gpu_parity.py tests/test_fuzz_gpu.py -m gpu -q 2>&1 | tail -3
OSQP_AMD_SETUP_TRACE=1 OSQP_AMD_SYMBOLIC_TRACE=1 timeout 600 ZOO_LABELS=gpu_direct python tools/zoo_rates.py equality_qp 2> $O/setup_trace_equality_qp.txt | cut -c1-300
grep -v amdgpu.ids $O/setup_trace_equality_qp.txt | cut -c1-100 | head -60
