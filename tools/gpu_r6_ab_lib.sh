# A/B of two builds of the library on the direct workloads: bash tools/gpu_r6_ab_lib.sh libA.so libB.so ...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp OSQP_AMD_BENCH_CPU_FULL=0 OSQP_AMD_BENCH_OTHERS=0
for lib in "$@"; do
for w in grid2d-5e5 grid2d-1e6 control-1e6; do
OSQP_AMD_LIB=osqp.jl_amd/csrc/$lib timeout 600 python bench.py --workload $w --no-cpu --traffic off --steps 100 --warmup 25 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib $w: %.1f it/s  to eps %.4f s' % (d['value'], d['time_to_eps_s']))"
done
OSQP_AMD_LIB=osqp.jl_amd/csrc/$lib timeout 600 python tools/grid3d_probe.py 50 2>&1 | grep "^second" | sed "s/^/$lib /"
done
