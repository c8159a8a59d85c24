cd $GRAFT_REPO_ROOT
for cfg in "X=1" "OSQP_AMD_PANEL_SHIFT=13" "OSQP_AMD_PANEL_SHIFT=13 OSQP_AMD_PANEL_TILE_NNZ=40000" "OSQP_AMD_PANEL_TILE_NNZ=40000" "OSQP_AMD_PANEL_TILE_NNZ=80000"; do echo "$cfg"; env $cfg timeout 600 python bench.py --workload rand-1e5 --steps 20 --warmup 5 --no-cpu --traffic off 2>/dev/null | cut -c1-120; done
