# end state of round 5 for the pieces that changed after gpu_round5_final.sh: control-1e6 (bench line with live PMC + CPU oracle, kernel stats, timeline), zoo, refactor times, the suite
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_end5; mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
export OSQP_AMD_BENCH_CPU_FULL=0
timeout 1500 python bench.py --workload control-1e6 > $O/bench_control1e6.json 2>/dev/null
timeout 900 python bench.py --workload control-1e6 --steps 20 --warmup 5 --no-cpu --traffic off > $O/bench_control1e6_k20w5.json 2>/dev/null
OSQP_AMD_SETUP_TRACE=1 OSQP_AMD_SYMBOLIC_TRACE=1 timeout 900 python bench.py --workload control-1e6 --steps 20 --warmup 5 --no-cpu --traffic off 2> $O/setup_trace_control1e6.txt > /dev/null
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/prof_c -o p -- python $GRAFT_REPO_ROOT/bench.py --workload control-1e6 --steps 100 --warmup 25 --no-cpu --traffic off > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find $O/prof_c -name '*_results.db' | head -1) > $O/kernel_stats_control-1e6.md
python $GRAFT_REPO_ROOT/tools/factor_timeline.py $(find $O/prof_c -name '*_results.db' | head -1) > $O/factor_timeline_control1e6.txt
python $GRAFT_REPO_ROOT/tools/rocpd_dispatches.py $(find $O/prof_c -name '*_results.db' | head -1) k_mf_front 40 > $O/front_dispatches_control1e6.txt
rm -rf $O/prof_c
# PMC passes of the control-1e6 iteration per kernel (separate runs), equality_qp kernel stats + setup trace, the headline line as a check
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --workload control-1e6 --steps 20 --warmup 5 --no-cpu --traffic off > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $(find $O/pmc_$c -name '*_results.db' | head -1) k_ >> $O/pmc_control-1e6.txt
  rm -rf $O/pmc_$c
done
ZOO_LABELS=gpu_direct OSQP_AMD_SETUP_TRACE=1 OSQP_AMD_SYMBOLIC_TRACE=1 rocprofv3 --kernel-trace --stats -d $O/prof_e -o p -- python $GRAFT_REPO_ROOT/tools/zoo_rates.py equality_qp > $O/zoo_equality_qp.jsonl 2> $O/setup_trace_equality_qp.txt
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find $O/prof_e -name '*_results.db' | head -1) > $O/kernel_stats_equality_qp.md
rm -rf $O/prof_e
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu > $O/bench_rand1e6_k20w5_nocpu.json 2>/dev/null
timeout 900 python tools/zoo_rates.py > $O/zoo_rates.jsonl 2>/dev/null
timeout 600 python tools/refactor_time.py 800 8000 2>&1 | grep "T=" > $O/refactor_time.txt
timeout 600 python bench.py --workload mpc-batch --steps 20 --warmup 3 > $O/bench_mpc_batch.json 2>/dev/null
for f in bench_control1e6 bench_control1e6_k20w5 bench_mpc_batch bench_rand1e6_k20w5_nocpu; do python - $O/$f.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get("roofline") or {}; c = d.get("cpu_baseline") or {}
print(sys.argv[1].split("/")[-1], d.get("value"), d.get("ms_per_step"), "frac", r.get("frac"), "traffic", r.get("traffic"), "step", (r.get("step") or {}).get("frac"), "setup", d.get("setup_s"), "to_eps", d.get("time_to_eps_s"), "incl", d.get("iterations_per_s_incl_setup"), "cpu", c.get("value"))
PY
done
cat $O/refactor_time.txt; head -3 $O/factor_timeline_control1e6.txt
