cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_problem_zoo.py -m gpu -x -q 2>&1 | tail -3
for c in 1 0 1 0; do OSQP_AMD_DENSE_SYM=$c timeout 600 python tools/zoo_rates.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); g = d['gpu_direct']
    if d['problem'] in ('equality_qp',): print('sym $c', d['problem'], g['status'], g['iter'], g['it_per_s'], 'setup', g['setup_s'])
"; done
