set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02f
timeout 1500 python -m pytest tests -m gpu -x -q -k "not full_size" > gpurun_out/r02f/pytest.log 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/r02f/pytest.log
for a in 1 0; do OSQP_AMD_PCG_ASYNC=$a timeout 300 python bench.py --workload rand-1e5 --steps 200 --warmup 25 --no-cpu --traffic off 2>/dev/null | cut -c1-900; done
OSQP_AMD_GRAPH=0 timeout 300 python bench.py --workload rand-1e5 --steps 200 --warmup 25 --no-cpu --traffic off 2>/dev/null | cut -c1-300
timeout 300 python bench.py --workload rand-1e5 --steps 20 --warmup 5 --no-cpu --traffic off 2>/dev/null | cut -c1-300
