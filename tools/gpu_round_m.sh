cd $GRAFT_REPO_ROOT
ZOO_LABELS=gpu_direct timeout 600 python tools/zoo_rates.py portfolio svm huber lasso_data control 2>/dev/null | cut -c1-230
timeout 300 python bench.py --workload lasso-5e5 --no-cpu --traffic off 2>/dev/null | cut -c1-160
OSQP_FUZZ_BLOCKS=12 timeout 900 python -m pytest tests/test_fuzz_gpu.py tests/test_problem_zoo.py -m gpu -q -x 2>&1 | tail -2
