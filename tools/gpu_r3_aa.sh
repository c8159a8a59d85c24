#!/bin/bash
# allocator test, largest instance, rand-1e5 line with the new allocator
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3aa; mkdir -p $O
timeout 900 python -m pytest tests/test_devmem_gpu.py -q -x 2>&1 | tail -5
bash tools/gpu_round_big.sh 2>/dev/null | tail -1 | tee $O/largest_instance.json
timeout 300 python bench.py --workload rand-1e5 --no-cpu --traffic off 2>/dev/null | tail -1 > $O/bench_rand1e5.json
python - <<PY
import json
d=json.loads(open('$O/bench_rand1e5.json').read())
print({k:d[k] for k in ('value','setup_s','device_gb','device_peak_gb','time_to_eps_s')})
PY
