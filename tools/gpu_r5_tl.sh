cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_tl; mkdir -p $O
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0
timeout 900 python -m pytest tests/test_batch_gpu.py tests/test_sharded_gpu.py tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -3
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/prof_c -o p -- python $GRAFT_REPO_ROOT/bench.py --workload control-1e6 --steps 100 --warmup 25 --no-cpu --traffic off > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/factor_timeline.py $(find $O/prof_c -name '*_results.db' | head -1) > $O/factor_timeline_control1e6.txt
rocprofv3 --kernel-trace --stats -d $O/prof_b -o p -- python $GRAFT_REPO_ROOT/tools/batch_shapes.py 4096 > $O/batch_shapes.jsonl 2>/dev/null
python $GRAFT_REPO_ROOT/tools/rocpd_dispatches.py $(find $O/prof_b -name '*_results.db' | head -1) k_batch 40 > $O/batch_dispatches.txt
rm -rf $O/prof_c $O/prof_b
cd $GRAFT_REPO_ROOT
head -12 $O/factor_timeline_control1e6.txt; cut -c1-120 $O/batch_dispatches.txt
