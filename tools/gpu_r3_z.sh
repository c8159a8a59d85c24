#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3z; mkdir -p $O
echo "== poison, VMM from 1 MiB, address ranges never reused"
OSQP_AMD_SKIP_RAND1E6=1 OSQP_AMD_POISON=1 OSQP_AMD_VMM_MIN_MB=1 timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_full_size_gpu.py > $O/pytest_poison_vmm.log 2>&1; echo "rc=$?"; grep -E "passed|failed|FAILED|Error" $O/pytest_poison_vmm.log | head -20
echo "== default"
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?"; tail -3 $O/pytest.log
cat > /tmp/setup_only.py <<'PY'
import sys, time
sys.path.insert(0, sys.argv[1])
import osqp_jl_amd as oq, bench
lib = oq.load_library()
for k in range(3):
    m = oq.Model(lib); t0=time.time(); oq.setup_generated(m, 0, 1000000, 1000, 1, linsys_solver="pcg", **bench.SETTINGS); t1=time.time()
    st = oq.stats(m); oq.clean(m); print("setup wall %.3f s, clean %.3f s, resident %.2f GB peak %.2f GB" % (t1-t0, time.time()-t1, st[9]/1e9, st[20]/1e9), flush=True)
PY
for r in 1 2; do
  echo "== process $r" | tee -a $O/setup_trace.txt
  OSQP_AMD_SETUP_TRACE=1 timeout 300 python /tmp/setup_only.py $GRAFT_REPO_ROOT 2>&1 | grep -E "setup" | tee -a $O/setup_trace.txt | grep "setup wall"
done
for r in 1 2 3; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --traffic off > $O/bench_rand1e6_k20w5_$r.json 2>/dev/null; python - <<PY
import json
d=json.loads(open('$O/bench_rand1e6_k20w5_$r.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','setup_s','device_gb','device_peak_gb','run_time_s','iterations_per_s_incl_setup','time_to_eps_s')})
PY
done
