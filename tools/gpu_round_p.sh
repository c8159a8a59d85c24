cd $GRAFT_REPO_ROOT
OSQP_AMD_LIB=$GRAFT_REPO_ROOT/osqp.jl_amd/csrc/libosqp_amd_cv.so timeout 300 python bench.py --workload mpc-batch --steps 20 --warmup 3 --no-cpu --traffic off 2>/dev/null | cut -c1-260
timeout 300 python bench.py --workload mpc-batch --steps 20 --warmup 3 --no-cpu --traffic off 2>/dev/null | cut -c1-260
