# round 6: two-ended first phase of the sweeps in the four-wavefront batched kernel -- tests, A/B of the MPC bench line, cycle stamps
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_batch; mkdir -p $O
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0
timeout 1500 python -m pytest tests/test_batch_gpu.py tests/test_full_size_gpu.py -k "batch or mpc" -m gpu -q -x > $O/pytest.log 2>&1
grep -E "passed|failed|FAILED|Error" $O/pytest.log | head -30
for v in 1 0; do
  OSQP_AMD_BATCH_TWO_ENDED=$v OSQP_AMD_BATCH_TRACE=1 timeout 600 python bench.py --workload mpc-batch --no-cpu --traffic off > $O/bench_mpc_batch_te$v.json 2> $O/bench_te$v.err
  grep "first-phase" $O/bench_te$v.err | head -2
  python - $O/bench_mpc_batch_te$v.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "unit", "ms_per_step", "mean_iters_per_instance", "solved")}, d.get("roofline", {}).get("frac"))
PY
done
for v in 1 0; do
  OSQP_AMD_BATCH_TWO_ENDED=$v OSQP_AMD_LIB=$GRAFT_REPO_ROOT/osqp.jl_amd/csrc/libosqp_amd_prof.so timeout 300 python bench.py --workload mpc-batch --no-cpu --traffic off --steps 1 --warmup 0 2>/dev/null | grep "quad cycles" | head -2 | tee $O/phase_cycles_te$v.txt
done
