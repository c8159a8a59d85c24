# round 6: fronts beyond LDS -- the forced-threshold tests, then the grid at full size
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_mfbig; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_multifrontal_gpu.py -m gpu -q -x > $O/pytest.log 2>&1
tail -15 $O/pytest.log
