cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_tree; mkdir -p $O
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0 OSQP_AMD_BENCH_OTHERS=0
run() {
  w=$1; shift
  env "$@" timeout 900 python bench.py --workload $w --no-cpu --traffic off --steps 100 --warmup 25 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w $*: %.1f it/s  %.4f ms/step  to eps %.4f s' % (d['value'], d['ms_per_step'], d['time_to_eps_s']))"
}
for w in grid2d-5e5 grid2d-1e6 control-1e6; do
run $w A=1
run $w OSQP_AMD_SNODE_TREE_512=0
run $w OSQP_AMD_SNODE_MAX=32
done
