cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5; do ZOO_LABELS=gpu_direct timeout 300 python tools/zoo_rates.py control 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['gpu_direct']['status'], d['gpu_direct']['iter'], d['gpu_direct']['it_per_s'])"; done
timeout 600 python -m pytest tests/test_problem_zoo.py tests/test_fuzz_gpu.py -m gpu -q -k "nested or supernodal or timed_out" 2>&1 | tail -1
