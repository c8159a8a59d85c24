cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_batch_gpu.py -m gpu -x -q 2>&1 | tail -15
OSQP_AMD_BATCH_MFMA=0 timeout 900 python -m pytest tests/test_batch_gpu.py -m gpu -x -q -k generic 2>&1 | tail -5
