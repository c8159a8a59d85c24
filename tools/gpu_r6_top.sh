cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_tree; mkdir -p $O
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0 OSQP_AMD_BENCH_OTHERS=0
timeout 1200 python -m pytest tests/test_multifrontal_gpu.py tests/test_full_size_gpu.py -k "multifrontal or fronts or grid2d or control" -m gpu -q -x 2>&1 | tail -4
for top in 1 0; do
for w in grid2d-5e5 grid2d-1e6 control-1e6; do
  OSQP_AMD_SNODE_TOP=$top OSQP_AMD_SETUP_TRACE=1 timeout 900 python bench.py --workload $w --no-cpu --traffic off --steps 100 --warmup 25 2> $O/trace.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('top=$top $w: %.1f it/s  %.4f ms/step  to eps %.4f s  iters %d  frac %.3f step %.3f' % (d['value'], d['ms_per_step'], d['time_to_eps_s'], d['iters_to_eps'], d['roofline']['frac'], d['roofline']['step']['frac']))"
  grep "top part" $O/trace.txt
done
done | tee $O/top_ab.txt
