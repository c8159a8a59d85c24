cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05_try; mkdir -p $O
timeout 1500 python -m pytest tests/test_problem_zoo.py tests/test_multifrontal_gpu.py tests/test_fuzz_gpu.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
for c in 1 0 1 0; do
  OSQP_AMD_DIRECT_LEAVE_RHS=$c timeout 600 python bench.py --workload control-1e6 --steps 100 --warmup 25 --no-cpu --traffic off > $O/b.json 2>/dev/null
  python - $O/b.json "$c" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get("roofline") or {}
print("leave_rhs", sys.argv[2], "it/s", d.get("value"), "ms", d.get("ms_per_step"), "frac", r.get("frac"), "to_eps", d.get("time_to_eps_s"), "iters", d.get("iters_to_eps"), d.get("status"))
PY
done
for c in 1 0; do OSQP_AMD_DIRECT_LEAVE_RHS=$c python - <<'PY'
import sys, os, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, osqp_jl_amd as oq, qp_zoo
lib = oq.load_library()
for T in (800, 8000, 30000):
    prob = qp_zoo.control(nx=12, nu=6, T=T)
    m = oq.Model(lib)
    oq.setup(m, linsys_solver="direct", verbose=False, eps_abs=1e-4, eps_rel=1e-4, adaptive_rho_interval=50, check_termination=25, max_iter=400, **prob)
    r = oq.solve(m)
    best = 1e9
    for rep in range(3):
        oq.warm_start(m, x=np.zeros(prob["P"].shape[0]), y=np.zeros(prob["A"].shape[0]))
        r = oq.solve(m)
        best = min(best, r.info.solve_time)
    print("leave_rhs", os.environ["OSQP_AMD_DIRECT_LEAVE_RHS"], "T", T, r.info.status, r.info.iter, "it/s %.0f" % (r.info.iter / best))
    oq.clean(m)
PY
done
