#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3u; mkdir -p $O
cat > /tmp/setup_only.py <<'PY'
import sys, time, ctypes as C
sys.path.insert(0, sys.argv[1])
import osqp_jl_amd as oq, bench
lib = oq.load_library()
for k in range(2):
    m = oq.Model(lib); t0=time.time()
    try:
        oq.setup_generated(m, 0, 1000000, 1000, 1, linsys_solver="pcg", **bench.SETTINGS)
    except Exception as e:
        lib.osqp_amd_last_error.restype = C.c_char_p
        print("setup FAILED:", e, lib.osqp_amd_last_error()); break
    t1=time.time(); st = oq.stats(m); oq.clean(m); print("setup wall %.3f s, clean %.3f s, resident %.2f GB peak %.2f GB" % (t1-t0, time.time()-t1, st[9]/1e9, st[20]/1e9), flush=True)
PY
OSQP_AMD_ALLOC_TRACE=1 OSQP_AMD_SETUP_TRACE=1 timeout 300 python /tmp/setup_only.py $GRAFT_REPO_ROOT 2>&1 | grep -E "setup|alloc" > $O/alloc_trace.txt
tail -30 $O/alloc_trace.txt | cut -c1-200
