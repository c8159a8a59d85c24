#!/bin/bash
# round 4, mpc-batch evidence: tests, bench line with live PMC traffic, rocprofv3 kernel stats, phase cycle stamps of both kernels
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_batch; mkdir -p $O
timeout 900 python -m pytest tests/test_batch_gpu.py tests/test_full_size_gpu.py -k "batch or mpc" -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 600 python bench.py --workload mpc-batch > $O/bench_mpc_batch.json 2> $O/bench_mpc_batch.err
OSQP_AMD_BATCH_QUAD=0 OSQP_AMD_BENCH_TRAFFIC=off timeout 600 python bench.py --workload mpc-batch > $O/bench_mpc_batch_512.json 2>/dev/null
python - $O/bench_mpc_batch.json $O/bench_mpc_batch_512.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1]); r = d.get("roofline") or {}
    print(f.split("/")[-1], d.get("ms_per_step"), d.get("instances_per_s"), "solved", d.get("solved"), "frac", r.get("frac"), "traffic", r.get("traffic"))
PY
cd /tmp; rm -rf /tmp/prof_b
OSQP_AMD_BENCH_TRAFFIC=off timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o b -- python $GRAFT_REPO_ROOT/bench.py --workload mpc-batch > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_b -name "*_results.db" | head -1) > $O/kernel_stats_mpc-batch.md; head -6 $O/kernel_stats_mpc-batch.md | cut -c1-200
for q in 1 0; do OSQP_AMD_BATCH_QUAD=$q OSQP_AMD_LIB=osqp.jl_amd/csrc/libosqp_amd_prof.so OSQP_AMD_BENCH_TRAFFIC=off python bench.py --workload mpc-batch --steps 1 --warmup 0 2>&1 | grep "cycles" | head -2; done > $O/phase_cycles.txt; cat $O/phase_cycles.txt
