# end-of-round-3 evidence: gpu suite, bench lines + rocprofv3 kernel stats (+ PMC passes) for every workload -> gpurun_out/r03_final
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03_final
mkdir -p $O
export TMPDIR=/tmp
echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  nproc: $(nproc)" > $O/host.txt
# 0. the gpu suite and smoke
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
# 0b. rand-1e6 parity record of the END state (engine iterate vs CPU oracle after W + K iterations, host KKT; ~12 minutes, mostly the oracle)
timeout 2400 python tools/cpu_rand1e6.py --phases gpu,kkt,cpu --out $O/rand1e6_parity.json --cpu-record $O/cpu_rand1e6_from_parity_run.json > $O/rand1e6_parity.log 2>&1; echo "parity rc=$?"; tail -2 $O/rand1e6_parity.log | cut -c1-400
# 1. bench lines (the driver's K/W first, then bench.py's defaults); --cpu-full re-measures the CPU record of rand-1e6 in this run
timeout 1800 python bench.py --steps 20 --warmup 5 --cpu-full > $O/bench_rand1e6_k20w5.json 2> $O/bench_rand1e6_k20w5.err; echo "rand-1e6 k20w5 rc=$?"
timeout 900 python bench.py > $O/bench_rand1e6_default.json 2> $O/bench_rand1e6_default.err
timeout 600 python bench.py --workload rand-1e5 > $O/bench_rand1e5.json 2>/dev/null
timeout 600 python bench.py --workload rand-1e5 --steps 20 --warmup 5 > $O/bench_rand1e5_k20w5.json 2>/dev/null
timeout 600 python bench.py --workload lasso-5e5 > $O/bench_lasso5e5.json 2>/dev/null
timeout 600 python bench.py --workload mpc-batch --steps 20 --warmup 3 > $O/bench_mpc_batch.json 2>/dev/null
OSQP_AMD_BENCH_ONE_DEVICE=1 OSQP_AMD_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --workload rand-1e5 --steps 50 --warmup 10 > $O/bench_2ranks_one_device_gloo.json 2>/dev/null
cp profiles/r03_cpu_rand1e6.json $O/cpu_rand1e6_record.json
for f in bench_rand1e6_k20w5 bench_rand1e6_default bench_rand1e5 bench_rand1e5_k20w5 bench_lasso5e5 bench_mpc_batch; do python - $O/$f.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], d.get("value"), d.get("ms_per_step"), "frac", r.get("frac"), "step", (r.get("step") or {}).get("frac"), "setup", d.get("setup_s"), "to_eps", d.get("time_to_eps_s"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "stale", (d.get("cpu_baseline") or {}).get("stale"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
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
rm -rf $O/prof_* $O/pmc_FETCH* $O/pmc_WRITE*
# 4. setup trace (no profiler)
cd $GRAFT_REPO_ROOT
OSQP_AMD_SETUP_TRACE=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --traffic off 2> $O/setup_trace_rand1e6.txt > /dev/null; grep "\[setup\]" $O/setup_trace_rand1e6.txt | tail -10
ls -la $O
