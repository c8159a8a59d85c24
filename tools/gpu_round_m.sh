cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_fuzz_gpu.py tests/test_problem_zoo.py -m gpu -x -q 2>&1 | tail -3
ZOO_LABELS=gpu_direct timeout 300 python tools/zoo_rates.py control 2>/dev/null | cut -c1-250
