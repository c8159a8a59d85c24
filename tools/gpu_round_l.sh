cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_full_size_gpu.py -m gpu -x -q 2>&1 | tail -3
for cfg in "OSQP_AMD_PANEL_GROUP=8" "OSQP_AMD_PANEL_GROUP=2" "OSQP_AMD_PANEL_TILE_NNZ=49152" "OSQP_AMD_PANEL_TILE_NNZ=32768" "OSQP_AMD_PANEL_TILE_NNZ=32768 OSQP_AMD_PANEL_GROUP=8"; do python tools/sweep_spmv.py rand-1e6 $cfg 2>/dev/null; done
for cfg in "OSQP_AMD_PANEL_TILE_NNZ=40000" "OSQP_AMD_PANEL_TILE_NNZ=32768" "OSQP_AMD_PANEL_TILE_NNZ=20000"; do python tools/sweep_spmv.py rand-1e5 $cfg 2>/dev/null; done
