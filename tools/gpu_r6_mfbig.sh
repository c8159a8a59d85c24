# round 6: fronts beyond LDS -- the forced-threshold tests, then refactorisation time on the grid at full size
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_mfbig; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_multifrontal_gpu.py -m gpu -q -x > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 900 python tools/refactor_time.py grid 700 1000 2>&1 | grep -v amdgpu.ids | tee $O/refactor_time_grid.txt
