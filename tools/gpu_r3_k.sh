cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3k
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r3k/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3k/pytest.log
cat > /tmp/setup_only.py <<'PY'
import sys, time
sys.path.insert(0, sys.argv[1])
import osqp_jl_amd as oq, bench
lib = oq.load_library()
for k in range(2):
    m = oq.Model(lib); t0=time.time(); oq.setup_generated(m, 0, 1000000, 1000, 1, linsys_solver="pcg", **bench.SETTINGS); print("setup wall", time.time()-t0, "peak GB", oq.stats(m)[20]/1e9)
    oq.clean(m)
PY
cd /tmp && OSQP_AMD_SETUP_TRACE=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_r3k -o s -- python /tmp/setup_only.py $GRAFT_REPO_ROOT 2>&1 | grep -E "setup" 
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_r3k -name "*_results.db" | head -1)
python tools/rocpd_summary.py $DB > gpurun_out/r3k/kernel_stats_setup.md 2>&1; head -16 gpurun_out/r3k/kernel_stats_setup.md | cut -c1-150
