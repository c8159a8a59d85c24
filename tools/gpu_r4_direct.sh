#!/bin/bash
# round 4, direct back-end evidence: zoo rates (dense block by block sweeps), lasso-5e5 (short-row pivot kernel), control-1e6
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_direct; mkdir -p $O
timeout 900 python -m pytest tests/test_problem_zoo.py tests/test_fuzz_gpu.py tests/test_gpu_parity.py -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
ZOO_LABELS=gpu_direct,gpu_pcg timeout 900 python tools/zoo_rates.py > $O/zoo_rates.jsonl 2>/dev/null; cut -c1-330 $O/zoo_rates.jsonl
OSQP_AMD_SETUP_TRACE=1 ZOO_LABELS=gpu_direct python tools/zoo_rates.py equality_qp 2>&1 | grep "setup\]" | cut -c1-90 > $O/setup_trace_equality_qp.txt; tail -12 $O/setup_trace_equality_qp.txt
timeout 600 python bench.py --workload lasso-5e5 > $O/bench_lasso5e5.json 2>/dev/null
OSQP_AMD_SETUP_TRACE=1 timeout 1500 python bench.py --workload control-1e6 --cpu-seconds 20 > $O/bench_control1e6.json 2> $O/bench_control1e6.err; grep "setup\]" $O/bench_control1e6.err | cut -c1-90 | tail -12
python - $O/bench_lasso5e5.json $O/bench_control1e6.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); r = d.get("roofline") or {}
        print(f.split("/")[-1], d.get("value"), d.get("ms_per_step"), "frac", r.get("frac"), "traffic", r.get("traffic"), "setup", d.get("setup_s"), "to_eps", d.get("time_to_eps_s"), d.get("iters_to_eps"), d.get("status"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "ERR", e)
PY
cd /tmp; rm -rf /tmp/prof_l /tmp/prof_c
OSQP_AMD_BENCH_TRAFFIC=off timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_l -o l -- python $GRAFT_REPO_ROOT/bench.py --workload lasso-5e5 --no-cpu > /dev/null 2>&1
OSQP_AMD_BENCH_TRAFFIC=off timeout 1200 rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o c -- python $GRAFT_REPO_ROOT/bench.py --workload control-1e6 --no-cpu > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_l -name "*_results.db" | head -1) > $O/kernel_stats_lasso-5e5.md; head -8 $O/kernel_stats_lasso-5e5.md | cut -c1-160
python tools/rocpd_summary.py $(find /tmp/prof_c -name "*_results.db" | head -1) > $O/kernel_stats_control-1e6.md; head -14 $O/kernel_stats_control-1e6.md | cut -c1-160
