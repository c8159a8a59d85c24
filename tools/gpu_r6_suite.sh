# round 6: the whole GPU suite, as the driver runs it
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_suite; mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.log 2>&1
tail -8 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
