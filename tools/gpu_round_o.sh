cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_size_gpu.py -m gpu -x -q -k "pcg or rand or arithmetic or property or compact" 2>&1 | tail -4
for pair in 1 0; do
echo "pair=$pair"
OSQP_AMD_SPMV_PAIR=$pair timeout 600 python bench.py --workload rand-1e5 --steps 20 --warmup 5 --no-cpu --traffic off 2>/dev/null | cut -c1-200
OSQP_AMD_SPMV_PAIR=$pair timeout 600 python bench.py --workload rand-1e5 --steps 200 --warmup 25 --no-cpu --traffic off 2>/dev/null | cut -c1-200
done
