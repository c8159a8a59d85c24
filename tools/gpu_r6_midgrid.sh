# round 6: size of the dense top on mid-size grids (200 x 200 ... 440 x 440): refactorisation time and whole-solve rate
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp REFACTOR_GRID=1
for g in 200 300 440; do
  for k in 600 1200 2400 4300; do
    OSQP_AMD_SN_DENSE_MAX=$k timeout 600 python tools/refactor_time.py --child $g 2>&1 | grep "T=" | sed "s/^/kmax=$k /"
  done
done
