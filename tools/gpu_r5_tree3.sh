cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_tree; mkdir -p $O
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0
timeout 1500 python -m pytest tests/test_problem_zoo.py tests/test_multifrontal_gpu.py tests/test_fuzz_gpu.py tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_full_size_gpu.py -k "control" -m gpu -q 2>&1 | tail -2
timeout 900 python bench.py --workload control-1e6 --steps 100 --warmup 25 --no-cpu --traffic off 2>/dev/null > $O/final.json
python - $O/final.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("control-1e6", d["value"], d["ms_per_step"], d["iters_to_eps"], d["time_to_eps_s"], d["roofline"]["frac"], d["setup_s"])
PY
for pe in default 0; do
if [ $pe = 0 ]; then export OSQP_AMD_SNODE_TREE_PERSIST=0; fi
ZOO_LABELS=gpu_direct timeout 600 python tools/zoo_rates.py control 2>/dev/null | cut -c1-200
python - <<'PY'
import sys, time
sys.path.insert(0, "tests")
import osqp_jl_amd as oq, qp_zoo
prob = qp_zoo.control(nx=12, nu=6, T=8000)
m = oq.Model(oq.load_library())
oq.setup(m, linsys_solver="direct", verbose=False, eps_abs=1e-4, eps_rel=1e-4, adaptive_rho_interval=50, **prob)
t0 = time.perf_counter(); r = oq.solve(m); tt = time.perf_counter() - t0
print("T=8000", r.info.status, r.info.iter, "it/s %.0f" % (r.info.iter / tt))
PY
done
