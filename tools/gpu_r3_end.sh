#!/bin/bash
# refresh of the end-of-round evidence that later commits touched: gpu suite + smoke, headline line (driver protocol, with the CPU
# record of the earlier full run), rand-1e5 lines, setup trace and setup kernel stats
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r03_end; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_rand1e6_k20w5.json 2>/dev/null
timeout 900 python bench.py > $O/bench_rand1e6_default.json 2>/dev/null
timeout 600 python bench.py --workload rand-1e5 > $O/bench_rand1e5.json 2>/dev/null
timeout 600 python bench.py --workload rand-1e5 --steps 20 --warmup 5 > $O/bench_rand1e5_k20w5.json 2>/dev/null
for f in bench_rand1e6_k20w5 bench_rand1e6_default bench_rand1e5 bench_rand1e5_k20w5; do python - $O/$f.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get("roofline") or {}
print(sys.argv[1].split("/")[-1], d.get("value"), d.get("ms_per_step"), "frac", r.get("frac"), "traffic", r.get("traffic"), "setup", d.get("setup_s"), "peak", d.get("device_peak_gb"), "to_eps", d.get("time_to_eps_s"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "stale", (d.get("cpu_baseline") or {}).get("stale"))
PY
done
cat > /tmp/setup_only.py <<'PY'
import sys, time
sys.path.insert(0, sys.argv[1])
import osqp_jl_amd as oq, bench
lib = oq.load_library()
for k in range(int(sys.argv[2])):
    m = oq.Model(lib); t0=time.time(); oq.setup_generated(m, 0, 1000000, 1000, 1, linsys_solver="pcg", **bench.SETTINGS); t1=time.time()
    st = oq.stats(m); oq.clean(m); print("setup wall %.3f s, clean %.3f s, resident %.2f GB peak %.2f GB" % (t1-t0, time.time()-t1, st[9]/1e9, st[20]/1e9), flush=True)
PY
OSQP_AMD_SETUP_TRACE=1 timeout 300 python /tmp/setup_only.py $GRAFT_REPO_ROOT 2 2>&1 | grep -E "setup" > $O/setup_trace_rand1e6.txt; tail -11 $O/setup_trace_rand1e6.txt | cut -c1-170
cd /tmp; rm -rf /tmp/prof_s
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o s -- python /tmp/setup_only.py $GRAFT_REPO_ROOT 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/prof_s -name "*_results.db" | head -1) > $O/kernel_stats_setup.md; head -14 $O/kernel_stats_setup.md | cut -c1-140
