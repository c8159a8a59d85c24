cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for sh in 11 12 13 14; do for t in 8192 16384 32768 65536; do python tools/sweep_spmv.py rand-1e5 OSQP_AMD_PANEL_SHIFT=$sh OSQP_AMD_PANEL_TILE_NNZ=$t 2>/dev/null; done; done | tee gpurun_out/sweep_rand1e5.log
