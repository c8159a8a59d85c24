# round 6: register-blocked pivot loop of the LDS fronts: tests, A/B of the refactorisation times (control T = 800 / 8000, control-1e6, grids)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0 OSQP_AMD_BENCH_OTHERS=0
timeout 1500 python -m pytest tests/test_multifrontal_gpu.py -m gpu -q -x 2>&1 | tail -3
for v in 1 0; do
  for T in 800 8000 30000; do OSQP_AMD_MF_REGPIV=$v timeout 600 python tools/refactor_time.py --child $T 2>&1 | grep "T=" | sed "s/^/regpiv=$v /"; done
  REFACTOR_GRID=1 OSQP_AMD_MF_REGPIV=$v timeout 600 python tools/refactor_time.py --child 700 2>&1 | grep "T=" | sed "s/^/regpiv=$v grid /"
done
for v in 1 0; do
OSQP_AMD_MF_REGPIV=$v timeout 900 python bench.py --workload control-1e6 --no-cpu --traffic off --steps 100 --warmup 25 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('regpiv=$v control-1e6: %.1f it/s  to eps %.4f s  setup %.2f' % (d['value'], d['time_to_eps_s'], d['setup_s']))"
done
