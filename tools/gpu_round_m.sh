cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_fuzz_gpu.py tests/test_problem_zoo.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4
ZOO_LABELS=gpu_direct timeout 300 python tools/zoo_rates.py control portfolio lasso_data svm huber 2>/dev/null | cut -c1-260
