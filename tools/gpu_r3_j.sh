cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3j
cat > /tmp/setup_only.py <<'PY'
import sys, time
sys.path.insert(0, sys.argv[1])
import osqp_jl_amd as oq, bench
lib = oq.load_library()
m = oq.Model(lib); t0=time.time(); oq.setup_generated(m, 0, 1000000, 1000, 1, linsys_solver="pcg", **bench.SETTINGS); print("setup", time.time()-t0)
oq.clean(m)
PY
cd /tmp && OSQP_AMD_SETUP_TRACE=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_r3j -o s -- python /tmp/setup_only.py $GRAFT_REPO_ROOT 2>&1 | grep -E "setup" 
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_r3j -name "*_results.db" | head -1)
python tools/rocpd_summary.py $DB > gpurun_out/r3j/kernel_stats_setup.md 2>&1; head -24 gpurun_out/r3j/kernel_stats_setup.md | cut -c1-150
