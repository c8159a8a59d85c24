# round 6: the dense top over the supernodes -- forced-threshold tests, then the grid at full size (bench line, refactorisation time)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_snd; mkdir -p $O
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0 OSQP_AMD_BENCH_OTHERS=0
timeout 1200 python -m pytest tests/test_multifrontal_gpu.py -m gpu -q -x -k "dense_top or pivot_block" > $O/pytest.log 2>&1
tail -15 $O/pytest.log
run() {
  w=$1; shift
  env "$@" OSQP_AMD_SETUP_TRACE=1 timeout 900 python bench.py --workload $w --no-cpu --traffic off --steps 100 --warmup 25 2> $O/trace.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w $*: %.1f it/s  %.4f ms/step  to eps %.4f s  iters %d  frac %.3f step %.3f setup %.2f' % (d['value'], d['ms_per_step'], d['time_to_eps_s'], d['iters_to_eps'], d['roofline']['frac'], d['roofline']['step']['frac'], d['setup_s']))"
  grep "dense top" $O/trace.txt
}
for k in ${KS:-0 1500 3072 4300}; do
  run grid2d-5e5 OSQP_AMD_SN_DENSE_MAX=$k
  REFACTOR_GRID=1 OSQP_AMD_SN_DENSE_MAX=$k timeout 600 python tools/refactor_time.py --child 700 2>&1 | grep "T="
done | tee $O/sweep.txt
