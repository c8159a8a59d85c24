cd $GRAFT_REPO_ROOT
python tools/sweep_spmv.py rand-1e5 2>/dev/null
python tools/sweep_spmv.py rand-1e6 2>/dev/null
timeout 600 python bench.py --workload rand-1e5 --steps 20 --warmup 5 --no-cpu --traffic off 2>/dev/null | cut -c1-200
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "spmv or arithmetic or panel" 2>&1 | tail -2
