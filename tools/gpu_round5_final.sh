# end-of-round-5 evidence: gpu suite, bench lines + rocprofv3 kernel stats (+ PMC passes) for every workload -> gpurun_out/r05_final
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_final
mkdir -p $O
export TMPDIR=/tmp
echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  nproc: $(nproc)" > $O/host.txt
# 0. the gpu suite and smoke
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
# 1. bench lines.  The headline at the driver's K/W with the CPU oracle timed inside the run (default behaviour, ~10 minutes);
#    everything else with the committed CPU record / the live small samples
timeout 1800 python bench.py --steps 20 --warmup 5 > $O/bench_rand1e6_k20w5.json 2> $O/bench_rand1e6_k20w5.err; echo "rand-1e6 k20w5 rc=$?"
cp gpurun_out/cpu_full_record.json $O/cpu_rand1e6_record.json 2>/dev/null
export OSQP_AMD_BENCH_CPU_FULL=0
timeout 900 python bench.py > $O/bench_rand1e6_default.json 2> $O/bench_rand1e6_default.err
timeout 600 python bench.py --workload rand-1e5 > $O/bench_rand1e5.json 2>/dev/null
timeout 600 python bench.py --workload rand-1e5 --steps 20 --warmup 5 > $O/bench_rand1e5_k20w5.json 2>/dev/null
timeout 600 python bench.py --workload lasso-5e5 > $O/bench_lasso5e5.json 2>/dev/null
timeout 600 python bench.py --workload mpc-batch --steps 20 --warmup 3 > $O/bench_mpc_batch.json 2>/dev/null
timeout 1200 python bench.py --workload control-1e6 > $O/bench_control1e6.json 2>/dev/null
for f in bench_rand1e6_k20w5 bench_rand1e6_default bench_rand1e5 bench_rand1e5_k20w5 bench_lasso5e5 bench_mpc_batch bench_control1e6; do python - $O/$f.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    c = d.get("cpu_baseline") or {}
    print(sys.argv[1].split("/")[-1], d.get("value"), d.get("ms_per_step"), "frac", r.get("frac"), "traffic", r.get("traffic"), "step", (r.get("step") or {}).get("frac"), "setup", d.get("setup_s"), "to_eps", d.get("time_to_eps_s"), "cpu", c.get("value"), "live", c.get("live"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
# 2. kernel stats of the same commands
cd /tmp
for w in rand-1e6 rand-1e5 lasso-5e5 mpc-batch control-1e6; do
  rocprofv3 --kernel-trace --stats -d $O/prof_$w -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 20 --warmup 5 --no-cpu --traffic off > $O/prof_$w.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find $O/prof_$w -name '*_results.db' | head -1) > $O/kernel_stats_$w.md
done
# 3. PMC passes (separate runs) for the direct iteration kernels and the batched kernel
for c in FETCH_SIZE WRITE_SIZE; do
  for w in lasso-5e5 mpc-batch control-1e6; do
    rocprofv3 --kernel-trace --pmc $c -d $O/pmc_${c}_$w -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 20 --warmup 5 --no-cpu --traffic off > /dev/null 2>&1
    python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $(find $O/pmc_${c}_$w -name '*_results.db' | head -1) k_ >> $O/pmc_$w.txt
  done
done
rm -rf $O/prof_* $O/pmc_FETCH* $O/pmc_WRITE*
cd $GRAFT_REPO_ROOT
# 4. the QP classes through the direct back-end, and the batched kernel's phase stamps (experiment build)
timeout 900 python tools/zoo_rates.py > $O/zoo_rates.jsonl 2>/dev/null
OSQP_AMD_LIB=osqp.jl_amd/csrc/libosqp_amd_prof.so python bench.py --workload mpc-batch --steps 1 --warmup 0 --no-cpu --traffic off 2>&1 | grep "cycles" | head -2 > $O/batch_phase_cycles.txt
OSQP_AMD_BATCH_QUAD=0 OSQP_AMD_LIB=osqp.jl_amd/csrc/libosqp_amd_prof.so python bench.py --workload mpc-batch --steps 1 --warmup 0 --no-cpu --traffic off 2>&1 | grep "cycles" | head -2 >> $O/batch_phase_cycles.txt
OSQP_AMD_BATCH_QUAD=0 timeout 600 python bench.py --workload mpc-batch --steps 20 --warmup 3 --no-cpu --traffic off > $O/bench_mpc_batch_512thread_kernel.json 2>/dev/null
# 5. setup traces (no profiler)
OSQP_AMD_SETUP_TRACE=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --traffic off 2> $O/setup_trace_rand1e6.txt > /dev/null
OSQP_AMD_SETUP_TRACE=1 OSQP_AMD_SYMBOLIC_TRACE=1 timeout 900 python bench.py --workload control-1e6 --steps 20 --warmup 5 --no-cpu --traffic off 2> $O/setup_trace_control1e6.txt > /dev/null
grep "\[setup\]" $O/setup_trace_control1e6.txt | tail -8 | cut -c1-100
# 5b. round 5: refactorisation times (multifrontal vs level by level), factor timeline of control-1e6, batched path per shape, equality_qp
timeout 600 python tools/refactor_time.py 800 8000 2>&1 | grep "T=" > $O/refactor_time.txt; cat $O/refactor_time.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/prof_c -o p -- python $GRAFT_REPO_ROOT/bench.py --workload control-1e6 --steps 100 --warmup 25 --no-cpu --traffic off > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/factor_timeline.py $(find $O/prof_c -name '*_results.db' | head -1) > $O/factor_timeline_control1e6.txt
python $GRAFT_REPO_ROOT/tools/rocpd_dispatches.py $(find $O/prof_c -name '*_results.db' | head -1) k_mf_front 40 > $O/mf_dispatches_control1e6.txt
ZOO_LABELS=gpu_direct OSQP_AMD_SETUP_TRACE=1 OSQP_AMD_SYMBOLIC_TRACE=1 rocprofv3 --kernel-trace --stats -d $O/prof_e -o p -- python $GRAFT_REPO_ROOT/tools/zoo_rates.py equality_qp > $O/zoo_equality_qp.jsonl 2> $O/setup_trace_equality_qp.txt
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find $O/prof_e -name '*_results.db' | head -1) > $O/kernel_stats_equality_qp.md
rocprofv3 --kernel-trace --stats -d $O/prof_b -o p -- python $GRAFT_REPO_ROOT/tools/batch_shapes.py 4096 > $O/batch_shapes.jsonl 2>/dev/null
python $GRAFT_REPO_ROOT/tools/rocpd_dispatches.py $(find $O/prof_b -name '*_results.db' | head -1) k_batch 40 > $O/batch_dispatches.txt
OSQP_AMD_BATCH_QUAD=0 rocprofv3 --kernel-trace --stats -d $O/prof_b0 -o p -- python $GRAFT_REPO_ROOT/tools/batch_shapes.py 4096 > $O/batch_shapes_512thread.jsonl 2>/dev/null
python $GRAFT_REPO_ROOT/tools/rocpd_dispatches.py $(find $O/prof_b0 -name '*_results.db' | head -1) k_batch 40 > $O/batch_dispatches_512thread.txt
rm -rf $O/prof_c $O/prof_e $O/prof_b $O/prof_b0
cd $GRAFT_REPO_ROOT
head -3 $O/factor_timeline_control1e6.txt
# PMC traffic of the control-1e6 iteration (bench.py collects it live with --traffic live: separate rocprofv3 passes inside)
timeout 1500 python bench.py --workload control-1e6 --traffic live --no-cpu > $O/bench_control1e6_pmc.json 2>/dev/null
# 6. rand-1e6 parity record of the END state (engine iterate vs CPU oracle after W + K iterations, host KKT; ~12 minutes, mostly the oracle)
timeout 2400 python tools/cpu_rand1e6.py --phases gpu,kkt,cpu --out $O/rand1e6_parity.json --cpu-record $O/cpu_rand1e6_from_parity_run.json > $O/rand1e6_parity.log 2>&1; echo "parity rc=$?"; tail -2 $O/rand1e6_parity.log | cut -c1-400
ls -la $O
