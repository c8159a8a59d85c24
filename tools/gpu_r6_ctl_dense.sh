cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0 OSQP_AMD_BENCH_OTHERS=0
run() {
  w=$1; shift
  env "$@" OSQP_AMD_SETUP_TRACE=1 timeout 900 python bench.py --workload $w --no-cpu --traffic off --steps 100 --warmup 25 2> /tmp/trace.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w $*: %.1f it/s  %.4f ms/step  to eps %.4f s  iters %d  frac %.3f step %.3f setup %.2f' % (d['value'], d['ms_per_step'], d['time_to_eps_s'], d['iters_to_eps'], d['roofline']['frac'], d['roofline']['step']['frac'], d['setup_s']))"
  grep "dense top" /tmp/trace.txt
}
for k in 600 1500 3000 6000; do run control-1e6 OSQP_AMD_SN_DENSE=2 OSQP_AMD_SN_DENSE_MAX=$k; done
OSQP_AMD_SN_DENSE=2 OSQP_AMD_SN_DENSE_MAX=1500 timeout 600 python tools/refactor_time.py --child 8000 2>&1 | grep "T="
