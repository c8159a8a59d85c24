cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for t in 36000 38000 40000 42000 45000 50000 65536; do python tools/sweep_spmv.py rand-1e5 OSQP_AMD_PANEL_TILE_NNZ=$t 2>/dev/null; done | tee gpurun_out/sweep_rand1e5_b.log
