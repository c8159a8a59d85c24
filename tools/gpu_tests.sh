cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tests
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/tests/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/tests/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
