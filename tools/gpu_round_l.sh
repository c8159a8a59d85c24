cd $GRAFT_REPO_ROOT
for sh in 12 13 14; do for t in 32768 65536 131072; do python tools/sweep_spmv.py rand-1e5 OSQP_AMD_PANEL_SHIFT=$sh OSQP_AMD_PANEL_TILE_NNZ=$t 2>/dev/null; done; done
