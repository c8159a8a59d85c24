cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for b in 65536 52000 46000 42000 40000 38000 34000; do
  echo -n "tile_nnz=$b: "
  OSQP_AMD_PANEL_TILE_NNZ=$b timeout 300 python bench.py --workload rand-1e5 --steps 20 --warmup 5 --no-cpu --traffic off 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['ms_per_launch'], d['time_to_eps_s'])"
done
for b in 65536 40000; do
  echo -n "K=100 W=25 tile_nnz=$b: "
  OSQP_AMD_PANEL_TILE_NNZ=$b timeout 300 python bench.py --workload rand-1e5 --no-cpu --traffic off 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['ms_per_launch'], d['time_to_eps_s'])"
done
