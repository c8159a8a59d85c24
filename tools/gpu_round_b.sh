set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
for t in 8192 16384 32768 65536; do python tools/sweep_spmv.py rand-1e5 OSQP_AMD_PANEL_TILE_NNZ=$t; done > gpurun_out/r02b/sweep_1e5.txt 2>&1
cat gpurun_out/r02b/sweep_1e5.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02b/prof_1e5 -o p -- python $GRAFT_REPO_ROOT/bench.py --workload rand-1e5 --steps 200 --warmup 25 --no-cpu --traffic off > $GRAFT_REPO_ROOT/gpurun_out/r02b/prof_1e5.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find gpurun_out/r02b/prof_1e5 -name '*_results.db' | head -1) > gpurun_out/r02b/prof_1e5_stats.md
head -30 gpurun_out/r02b/prof_1e5_stats.md
grep '^{' gpurun_out/r02b/prof_1e5.log | cut -c1-600
OSQP_AMD_LIB=$GRAFT_REPO_ROOT/osqp.jl_amd/csrc/libosqp_amd_prof.so python bench.py --workload mpc-batch --steps 2 --warmup 1 --no-cpu 2>&1 | grep -v '^{' | tail -5
