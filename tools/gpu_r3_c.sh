# round 3, call C: kernel stats of the rand-1e6 bench command (setup kernels included), 2-rank one-device check of the N > 1 line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3c
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_r3c -o rand1e6 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu --traffic off > $GRAFT_REPO_ROOT/gpurun_out/r3c/prof_bench.json 2>/dev/null; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_r3c -name "*_results.db" | head -1); echo "db=$DB"
python tools/rocpd_summary.py $DB > gpurun_out/r3c/kernel_stats_rand-1e6.md 2>&1; head -24 gpurun_out/r3c/kernel_stats_rand-1e6.md | cut -c1-170
OSQP_AMD_BENCH_BACKEND=gloo OSQP_AMD_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus 2 --workload rand-1e5 --steps 50 --warmup 10 --no-cpu > gpurun_out/r3c/bench_2ranks_one_device_gloo.json 2>&1; echo "2-rank rc=$?"; tail -c 1500 gpurun_out/r3c/bench_2ranks_one_device_gloo.json
timeout 600 python -m pytest tests/test_sharded_gpu.py -m gpu -x -q 2>&1 | tail -3
