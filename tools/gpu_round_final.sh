# end-of-round evidence: bench lines + rocprofv3 kernel stats (+ PMC passes) for every workload -> gpurun_out/r02_final
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r02_final
mkdir -p $O
export TMPDIR=/tmp
# 1. bench lines (default flags = what the driver runs, then the driver's own K/W)
timeout 900 python bench.py > $O/bench_rand1e6_default.json 2> $O/bench_rand1e6_default.err
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_rand1e6_k20w5.json 2> $O/bench_rand1e6_k20w5.err
timeout 600 python bench.py --workload rand-1e5 > $O/bench_rand1e5.json 2>/dev/null
timeout 600 python bench.py --workload rand-1e5 --steps 20 --warmup 5 --no-cpu > $O/bench_rand1e5_k20w5.json 2>/dev/null
timeout 600 python bench.py --workload lasso-5e5 > $O/bench_lasso5e5.json 2>/dev/null
timeout 600 python bench.py --workload mpc-batch --steps 20 --warmup 3 > $O/bench_mpc_batch.json 2>/dev/null
OSQP_AMD_BENCH_ONE_DEVICE=1 OSQP_AMD_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --workload rand-1e5 --steps 50 --warmup 10 > $O/bench_2ranks_one_device_gloo.json 2>/dev/null
# 2. kernel stats of the same commands
cd /tmp
for w in rand-1e6 rand-1e5 lasso-5e5 mpc-batch; do
  rocprofv3 --kernel-trace --stats -d $O/prof_$w -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 20 --warmup 5 --no-cpu --traffic off > $O/prof_$w.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find $O/prof_$w -name '*_results.db' | head -1) > $O/kernel_stats_$w.md
done
# 3. PMC passes (separate runs) for the direct iteration kernels and the batched kernel
for c in FETCH_SIZE WRITE_SIZE; do
  for w in lasso-5e5 mpc-batch; do
    rocprofv3 --kernel-trace --pmc $c -d $O/pmc_${c}_$w -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 20 --warmup 5 --no-cpu --traffic off > /dev/null 2>&1
    python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $(find $O/pmc_${c}_$w -name '*_results.db' | head -1) k_ >> $O/pmc_$w.txt
  done
done
rm -rf $O/prof_* $O/pmc_FETCH* $O/pmc_WRITE*  # the databases are large; the summaries stay
ls -la $O
# 4. standard QP classes (direct / PCG / CPU oracle) and the control class's kernel stats
cd $GRAFT_REPO_ROOT
timeout 1500 python tools/zoo_rates.py > $O/zoo_rates.jsonl 2>/dev/null
cd /tmp && ZOO_LABELS=gpu_direct timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_control -o ctl -- python $GRAFT_REPO_ROOT/tools/zoo_rates.py control > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find $O/prof_control -name '*_results.db' | head -1) > $O/kernel_stats_control.md
rm -rf $O/prof_control
ls -la $O
