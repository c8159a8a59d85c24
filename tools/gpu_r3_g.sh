cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OSQP_AMD_LIB=$GRAFT_REPO_ROOT/osqp.jl_amd/csrc/libosqp_amd_prof.so timeout 120 python bench.py --workload mpc-batch --steps 2 --warmup 1 --no-cpu --traffic off 2>&1 | grep -v '^{' | tail -3
