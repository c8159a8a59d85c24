cd $GRAFT_REPO_ROOT
OSQP_FUZZ_BLOCKS=40 timeout 1500 python -m pytest tests/test_fuzz_gpu.py -m gpu -q 2>&1 | grep -E "AssertionError|passed|failed" | head -10
