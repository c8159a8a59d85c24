cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_sharded_gpu.py tests/test_batch_gpu.py -m gpu -x -q 2>&1 | tail -15
