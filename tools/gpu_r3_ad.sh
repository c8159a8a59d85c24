#!/bin/bash
# (A/B of the host-loop switches on rand-1e5: profiles/r03_rand1e5_hostloop_ab.txt)
# host CG loop: speculative first iteration, extrapolation sums in the reduce, rhs left behind by the update -- A/B on rand-1e5, suite
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3ad; mkdir -p $O
run() { # name, env...
  name=$1; shift
  for w in "--steps 100 --warmup 25" "--steps 20 --warmup 5"; do
    env "$@" timeout 300 python bench.py --workload rand-1e5 $w --no-cpu --traffic off 2>/dev/null | tail -1 > $O/tmp.json
    python - "$name" "$w" <<PY
import json,sys
d=json.loads(open('$O/tmp.json').read())
print("%-28s %-24s %8.1f it/s  %.4f ms/step  cg/it %.2f  iters_to_eps %s  pri %.6e dua %.6e" % (sys.argv[1], sys.argv[2], d['value'], d['ms_per_step'], d['cg_iters_per_admm_iter'], d['iters_to_eps'], d['pri_res'], d['dua_res']))
PY
  done
}
{
run "all on" A=1
run "all on (again)" A=1
run "no speculation" OSQP_AMD_PCG_SPEC=0
run "no extrap fusion" OSQP_AMD_PCG_FUSE_EXTRAP=0
run "no rhs fusion" OSQP_AMD_PCG_FUSE_RHS=0
run "all off" OSQP_AMD_PCG_SPEC=0 OSQP_AMD_PCG_FUSE_EXTRAP=0 OSQP_AMD_PCG_FUSE_RHS=0
} | tee $O/ab_rand1e5.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --traffic off 2>/dev/null | tail -1 > $O/bench_rand1e6_k20w5.json
python - <<PY
import json
d=json.loads(open('$O/bench_rand1e6_k20w5.json').read())
print({k:d[k] for k in ('value','setup_s','device_peak_gb','time_to_eps_s','iters_to_eps','pri_res','dua_res')})
PY
