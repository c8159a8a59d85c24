# end-of-round-6 evidence: gpu suite, bench lines + rocprofv3 kernel stats (+ PMC passes) for every workload -> gpurun_out/r06_final
# usage: bash tools/gpu_round6_final.sh [parts]   parts: any of suite headline others prof pmc batch direct (default: all)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_final
mkdir -p $O
export TMPDIR=/tmp
PARTS=${1:-suite headline others prof pmc batch direct}
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)  nproc: $(nproc)" > $O/host.txt
show() { python - "$@" <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get("roofline") or {}
        c = d.get("cpu_baseline") or {}
        print(f.split("/")[-1], d.get("value"), d.get("ms_per_step"), "frac", r.get("frac"), "traffic", r.get("traffic"), "step", (r.get("step") or {}).get("frac"),
              "setup", d.get("setup_s"), "to_eps", d.get("time_to_eps_s"), "cpu", c.get("value"), "live", c.get("live"))
    except Exception as e:
        print(f, "unreadable", e)
PY
}
if has suite; then
  timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
fi
if has headline; then  # the driver's command: the headline with the CPU oracle timed inside the run and the other workloads beside it (~10 minutes)
  timeout 1800 python bench.py --steps 20 --warmup 5 > $O/bench_rand1e6_k20w5.json 2> $O/bench_rand1e6_k20w5.err; echo "rand-1e6 k20w5 rc=$?"
  cp gpurun_out/cpu_full_record.json $O/cpu_rand1e6_record.json 2>/dev/null
  show $O/bench_rand1e6_k20w5.json
fi
export OSQP_AMD_BENCH_CPU_FULL=0 OSQP_AMD_BENCH_OTHERS=0
if has others; then
  timeout 600 python bench.py --workload rand-1e5 > $O/bench_rand1e5.json 2>/dev/null
  timeout 600 python bench.py --workload rand-1e5 --steps 20 --warmup 5 > $O/bench_rand1e5_k20w5.json 2>/dev/null
  timeout 600 python bench.py --workload lasso-5e5 > $O/bench_lasso5e5.json 2>/dev/null
  timeout 600 python bench.py --workload mpc-batch --steps 20 --warmup 3 > $O/bench_mpc_batch.json 2>/dev/null
  timeout 1200 python bench.py --workload control-1e6 > $O/bench_control1e6.json 2>/dev/null
  OSQP_AMD_SETUP_TRACE=1 timeout 1200 python bench.py --workload grid2d-5e5 > $O/bench_grid2d_5e5.json 2> $O/setup_trace_grid2d_5e5.txt
  OSQP_AMD_SETUP_TRACE=1 timeout 1200 python bench.py --workload grid2d-1e6 > $O/bench_grid2d_1e6.json 2> $O/setup_trace_grid2d_1e6.txt
  show $O/bench_rand1e5.json $O/bench_rand1e5_k20w5.json $O/bench_lasso5e5.json $O/bench_mpc_batch.json $O/bench_control1e6.json $O/bench_grid2d_5e5.json $O/bench_grid2d_1e6.json
fi
cd /tmp
if has prof; then
  for w in rand-1e6 rand-1e5 lasso-5e5 mpc-batch; do
    rocprofv3 --kernel-trace --stats -d $O/prof_$w -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 20 --warmup 5 --no-cpu --traffic off > $O/prof_$w.log 2>&1
    python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find $O/prof_$w -name '*_results.db' | head -1) > $O/kernel_stats_$w.md
  done
  for w in control-1e6 grid2d-5e5 grid2d-1e6; do
    rocprofv3 --kernel-trace --stats -d $O/prof_$w -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 100 --warmup 25 --no-cpu --traffic off > $O/prof_$w.log 2>&1
    DB=$(find $O/prof_$w -name '*_results.db' | head -1)
    python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB > $O/kernel_stats_$w.md
    python $GRAFT_REPO_ROOT/tools/factor_timeline.py $DB > $O/factor_timeline_$w.txt 2>&1
  done
  rm -rf $O/prof_*
fi
if has pmc; then  # PMC passes (separate runs): HBM bytes of the direct iteration kernels; instruction mix of the batched kernel
  for c in FETCH_SIZE WRITE_SIZE; do
    for w in lasso-5e5 mpc-batch control-1e6 grid2d-5e5; do
      rocprofv3 --kernel-trace --pmc $c -d $O/pmc_${c}_$w -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 20 --warmup 5 --no-cpu --traffic off > /dev/null 2>&1
      python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $(find $O/pmc_${c}_$w -name '*_results.db' | head -1) k_ >> $O/pmc_$w.txt
    done
  done
  for c in "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64" "SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
    rocprofv3 --kernel-trace --pmc $c -d $O/pmc_mix -o p -- python $GRAFT_REPO_ROOT/bench.py --workload mpc-batch --steps 20 --warmup 3 --no-cpu --traffic off > /dev/null 2>&1
    python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $(find $O/pmc_mix -name '*_results.db' | head -1) k_batch >> $O/pmc_mix_mpc-batch.txt
    rm -rf $O/pmc_mix
  done
  rm -rf $O/pmc_FETCH* $O/pmc_WRITE*
fi
cd $GRAFT_REPO_ROOT
if has batch; then  # the batched path per shape (kernel time from rocprofv3), phase stamps of the MPC kernel (experiment build), the 512-thread kernel
  cd /tmp
  rocprofv3 --kernel-trace --stats -d $O/prof_b -o p -- python $GRAFT_REPO_ROOT/tools/batch_shapes.py 4096 > $O/batch_shapes.jsonl 2>/dev/null
  python $GRAFT_REPO_ROOT/tools/rocpd_dispatches.py $(find $O/prof_b -name '*_results.db' | head -1) k_batch 40 > $O/batch_dispatches.txt
  rm -rf $O/prof_b
  cd $GRAFT_REPO_ROOT
  [ -f osqp.jl_amd/csrc/libosqp_amd_prof.so ] && OSQP_AMD_LIB=osqp.jl_amd/csrc/libosqp_amd_prof.so python bench.py --workload mpc-batch --steps 1 --warmup 0 --no-cpu --traffic off 2>&1 | grep "cycles" | head -2 > $O/batch_phase_cycles.txt
  OSQP_AMD_BATCH_QUAD=0 timeout 600 python bench.py --workload mpc-batch --steps 20 --warmup 3 --no-cpu --traffic off > $O/bench_mpc_batch_512thread_kernel.json 2>/dev/null
fi
if has direct; then  # the QP classes through the direct back-end; refactorisation times; setup traces
  timeout 900 python tools/zoo_rates.py > $O/zoo_rates.jsonl 2>/dev/null
  (timeout 600 python tools/refactor_time.py 800 8000; timeout 900 python tools/refactor_time.py grid 700 1000) 2>&1 | grep "T=" > $O/refactor_time.txt; cat $O/refactor_time.txt
  OSQP_AMD_SETUP_TRACE=1 OSQP_AMD_SYMBOLIC_TRACE=1 timeout 900 python bench.py --workload control-1e6 --steps 20 --warmup 5 --no-cpu --traffic off 2> $O/setup_trace_control1e6.txt > /dev/null
  timeout 1500 python bench.py --workload control-1e6 --traffic live --no-cpu > $O/bench_control1e6_pmc.json 2>/dev/null
  timeout 1500 python bench.py --workload grid2d-5e5 --traffic live --no-cpu > $O/bench_grid2d_5e5_pmc.json 2>/dev/null
  show $O/bench_control1e6_pmc.json $O/bench_grid2d_5e5_pmc.json
fi
ls -la $O | head -60
