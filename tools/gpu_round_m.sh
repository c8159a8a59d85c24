cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_problem_zoo.py -m gpu -x -q -k "timed_out or nested" 2>&1 | tail -12
