cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_grid; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_full_size_gpu.py -k grid2d -m gpu -q -x > $O/pytest_grid.log 2>&1
tail -15 $O/pytest_grid.log
timeout 900 python -m pytest tests/test_problem_zoo.py tests/test_symbolic_host.py -m gpu -q -x -k "grid2d or lean or zoo" > $O/pytest_zoo.log 2>&1
tail -5 $O/pytest_zoo.log
