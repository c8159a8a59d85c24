cd $GRAFT_REPO_ROOT
for B in 24 32; do OSQP_AMD_LIB=$GRAFT_REPO_ROOT/osqp.jl_amd/csrc/libosqp_amd_b$B.so python tools/sweep_spmv.py rand-1e6 BATCH=$B 2>/dev/null; OSQP_AMD_LIB=$GRAFT_REPO_ROOT/osqp.jl_amd/csrc/libosqp_amd_b$B.so python tools/sweep_spmv.py rand-1e5 BATCH=$B 2>/dev/null; done
python tools/sweep_spmv.py rand-1e6 2>/dev/null
