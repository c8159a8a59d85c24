cd $GRAFT_REPO_ROOT
for c in -1 5e7; do OSQP_AMD_COMPACT_NNZ=$c timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --traffic off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', d['value'], d['ms_per_step'], d['device_gb'], d['roofline']['ms_per_launch'], d['time_to_eps_s'])"; done
for c in -1 5e7; do OSQP_AMD_COMPACT_NNZ=$c python tools/sweep_spmv.py rand-1e6; done
