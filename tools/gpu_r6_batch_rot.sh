# round 6: which SIMD carries the one-wavefront sweeps -- role rotation of the four wavefronts by workgroup id
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_batch_rot; mkdir -p $O
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0
for rot in 0 8 1 3 5 10; do
  for rep in 1 2; do
  OSQP_AMD_BATCH_ROT=$rot timeout 600 python bench.py --workload mpc-batch --no-cpu --traffic off > $O/bench_rot$rot.json 2> /dev/null
  python - $O/bench_rot$rot.json $rot <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("rot", sys.argv[2], d.get("ms_per_step"), d.get("solved"))
PY
  done
done | tee $O/rot_sweep.txt
