#!/bin/bash
# round 4: osqp_setup from host arrays at the headline size; the N = 4 / 8 bench lines on one device over gloo; the default
# bench line with the CPU oracle timed in the same run
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_host; mkdir -p $O
free -g | head -2
timeout 300 python tools/host_setup_rand1e6.py --n 20000 --per-row 20 --out $O/host_setup_small.json | cut -c1-400
OSQP_AMD_SETUP_TRACE=1 timeout 1500 python tools/host_setup_rand1e6.py --out $O/host_setup_rand1e6.json > $O/host_setup_rand1e6.log 2>&1; tail -1 $O/host_setup_rand1e6.log | cut -c1-900; grep "setup\]" $O/host_setup_rand1e6.log | cut -c1-100 | head -14
for N in 4 8; do
  OSQP_AMD_BENCH_ONE_DEVICE=1 OSQP_AMD_BENCH_BACKEND=gloo timeout 1200 python bench.py --gpus $N --workload rand-1e5 --no-cpu > $O/bench_${N}ranks_one_device_gloo.json 2> $O/bench_${N}ranks.err
  python - $O/bench_${N}ranks_one_device_gloo.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("ranks", d.get("n_gpus"), "value", d.get("value"), "launch", d.get("launch"), "seen", d.get("collective_ranks_seen"), "batch", (d.get("batch") or {}).get("value"), (d.get("batch") or {}).get("comm_ranks_seen"), "sharded", (d.get("sharded") or {}).get("value"), (d.get("sharded") or {}).get("comm_ranks_seen"), (d.get("sharded") or {}).get("error"))
except Exception as e:
    print("ERR", e, open(sys.argv[1]).read()[-300:])
PY
done
timeout 1700 python bench.py --steps 20 --warmup 5 > $O/bench_rand1e6_k20w5.json 2> $O/bench_rand1e6_k20w5.err
python - $O/bench_rand1e6_k20w5.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get("roofline") or {}; c = d.get("cpu_baseline") or {}
print("rand-1e6", d.get("value"), d.get("ms_per_step"), "frac", r.get("frac"), "traffic", r.get("traffic"), "setup", d.get("setup_s"), "to_eps", d.get("time_to_eps_s"), "cpu", c.get("value"), "live", c.get("live"), c.get("cpu_full_wall_s"), c.get("cpu_full_error"), c.get("cpu_full_skipped"))
PY
