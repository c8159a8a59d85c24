# round 6: the one-launch tree sized beyond what the occupancy query reports (OSQP_AMD_SNODE_TREE_PER_CU): rates, restarts
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0 OSQP_AMD_BENCH_OTHERS=0 OSQP_AMD_SETUP_TRACE=1
for w in grid2d-5e5 grid2d-1e6 control-1e6; do
for pc in 0 2 4; do env $( [ $pc != 0 ] && echo OSQP_AMD_SNODE_TREE_PER_CU=$pc ) timeout 600 python bench.py --workload $w --no-cpu --traffic off --steps 100 --warmup 25 2> /tmp/tr.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w per_cu=$pc: %.1f it/s  to eps %.4f s' % (d['value'], d['time_to_eps_s']))"; grep -E "one-launch tree:|tree starts|restart" /tmp/tr.txt | head -3; done; done
for pc in 0 2; do env $( [ $pc != 0 ] && echo OSQP_AMD_SNODE_TREE_PER_CU=$pc ) timeout 600 python tools/zoo_rates.py control 2>/dev/null | cut -c1-200; done
