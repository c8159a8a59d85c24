# the whole GPU suite + the default bench line (what the driver runs at round end)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_suite; mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
tail -15 $O/pytest.log
