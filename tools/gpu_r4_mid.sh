#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_mid; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "polish" > $O/pytest_polish.log 2>&1; tail -4 $O/pytest_polish.log | cut -c1-400
timeout 900 python -m pytest tests/test_batch_gpu.py tests/test_problem_zoo.py tests/test_full_size_gpu.py -x -q -k "not rand1e6" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
OSQP_AMD_BENCH_TRAFFIC=off timeout 600 python bench.py --workload mpc-batch --no-cpu > $O/bench_mpc_batch.json 2>/dev/null
OSQP_AMD_LIB=osqp.jl_amd/csrc/libosqp_amd_prof.so OSQP_AMD_BENCH_TRAFFIC=off python bench.py --workload mpc-batch --steps 1 --warmup 0 --no-cpu 2>&1 | grep "cycles" | head -2
OSQP_AMD_BENCH_TRAFFIC=off timeout 1500 python bench.py --workload control-1e6 --no-cpu > $O/bench_control1e6.json 2>/dev/null
python - $O/bench_mpc_batch.json $O/bench_control1e6.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d.get("roofline") or {}
        print(f.split("/")[-1], d.get("value"), d.get("ms_per_step"), "frac", r.get("frac"), "setup", d.get("setup_s"), "to_eps", d.get("time_to_eps_s"), d.get("iters_to_eps"), d.get("status"), d.get("solved"))
    except Exception as e:
        print(f, "ERR", e)
PY
cd /tmp; rm -rf /tmp/prof_c
OSQP_AMD_BENCH_TRAFFIC=off timeout 1200 rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o c -- python $GRAFT_REPO_ROOT/bench.py --workload control-1e6 --no-cpu > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_c -name "*_results.db" | head -1) > $O/kernel_stats_control-1e6.md; head -12 $O/kernel_stats_control-1e6.md | cut -c1-170
