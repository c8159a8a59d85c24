# round 6: quick check after a change to the dense top / the fronts: the multifrontal tests, grid + control bench lines, refactorisation times
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_quick; mkdir -p $O
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0 OSQP_AMD_BENCH_OTHERS=0
timeout 1500 python -m pytest tests/test_multifrontal_gpu.py -m gpu -q -x ${PYTEST_K:+-k "$PYTEST_K"} 2>&1 | tail -3
run() {
  w=$1; shift
  env "$@" OSQP_AMD_SETUP_TRACE=1 timeout 900 python bench.py --workload $w --no-cpu --traffic off --steps 100 --warmup 25 2> $O/trace.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w $*: %.1f it/s  %.4f ms/step  to eps %.4f s  iters %d  frac %.3f step %.3f setup %.2f' % (d['value'], d['ms_per_step'], d['time_to_eps_s'], d['iters_to_eps'], d['roofline']['frac'], d['roofline']['step']['frac'], d['setup_s']))"
}
for w in ${WORKLOADS:-grid2d-5e5 grid2d-1e6 control-1e6}; do run $w A=1; done
REFACTOR_GRID=1 timeout 600 python tools/refactor_time.py --child 700 2>&1 | grep "T="
REFACTOR_GRID=1 timeout 600 python tools/refactor_time.py --child 1000 2>&1 | grep "T="
timeout 600 python tools/refactor_time.py --child 8000 2>&1 | grep "T="
