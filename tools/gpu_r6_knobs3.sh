# round 6: without the front vectors (the rows gather everything) under a dense top: dense-top size and tree width
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_knobs2; mkdir -p $O
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0 OSQP_AMD_BENCH_OTHERS=0 OSQP_AMD_SNODE_TOP=0
run() {
  w=$1; shift
  env "$@" OSQP_AMD_SETUP_TRACE=1 timeout 900 python bench.py --workload $w --no-cpu --traffic off --steps 100 --warmup 25 2> $O/trace.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w $*: %.1f it/s  %.4f ms/step  to eps %.4f s  iters %d' % (d['value'], d['ms_per_step'], d['time_to_eps_s'], d['iters_to_eps']))"
}
for w in grid2d-5e5 grid2d-1e6; do
  run $w OSQP_AMD_SNODE_TREE_512=1
  run $w OSQP_AMD_SN_DENSE_MAX=3072
  run $w OSQP_AMD_SN_DENSE_MAX=6000
  run $w OSQP_AMD_SN_DENSE_MAX=6000 OSQP_AMD_SNODE_TREE_512=1
  run $w OSQP_AMD_SN_DENSE_MAX=8000
done 2>&1 | tee $O/knobs3.txt
