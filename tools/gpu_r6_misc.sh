# round 6: the new boundary / multi-rank tests and the bench line with its other_workloads block
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06_misc; mkdir -p $O
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0
timeout 1500 python -m pytest tests/test_c_harness.py tests/test_devmem_gpu.py tests/test_bench_multi_rank_gpu.py -m gpu -q -x > $O/pytest.log 2>&1
tail -15 $O/pytest.log
timeout 900 python bench.py --no-cpu --traffic off --steps 20 --warmup 5 > $O/bench_default_nocpu.json 2> $O/bench.err
python - $O/bench_default_nocpu.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "time_to_eps_s")}, d.get("device"))
for k, v in (d.get("other_workloads") or {}).items():
    print(k, json.dumps(v)[:300])
PY
tail -3 $O/bench.err
