#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3ai; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
cat > /tmp/setup_only.py <<'PY'
import sys, time
sys.path.insert(0, sys.argv[1])
import osqp_jl_amd as oq, bench
lib = oq.load_library()
for k in range(2):
    m = oq.Model(lib); t0=time.time(); oq.setup_generated(m, 0, 1000000, 1000, 1, linsys_solver="pcg", **bench.SETTINGS); t1=time.time()
    st = oq.stats(m); oq.clean(m); print("setup wall %.3f s, clean %.3f s, resident %.2f GB peak %.2f GB" % (t1-t0, time.time()-t1, st[9]/1e9, st[20]/1e9), flush=True)
PY
for g in 1 0; do echo "== OSQP_AMD_SELL_GATHER=$g"; OSQP_AMD_SELL_GATHER=$g OSQP_AMD_SETUP_TRACE=1 timeout 300 python /tmp/setup_only.py $GRAFT_REPO_ROOT 2>&1 | grep -E "setup" | tail -11 | cut -c1-170; done | tee $O/setup_trace.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --traffic off 2>/dev/null | tail -1 > $O/bench_rand1e6_k20w5.json; python - <<PY
import json
d=json.loads(open('$O/bench_rand1e6_k20w5.json').read())
print({k:d[k] for k in ('value','setup_s','device_gb','device_peak_gb','run_time_s','time_to_eps_s','pri_res','dua_res')})
PY
