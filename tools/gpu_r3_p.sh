#!/bin/bash
# where the setup's wall time goes on this box: hipMalloc / hipFree per stage, torch's HIP runtime against the system one
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3p; mkdir -p $O
cat > /tmp/setup_only.py <<'PY'
import sys, time, shutil
sys.path.insert(0, sys.argv[1])
import osqp_jl_amd as oq, bench
if len(sys.argv) > 2:      # a copy under another name: load_library does not preload torch's runtime for it -> system ROCm 7.2
    shutil.copy(sys.argv[1] + "/osqp.jl_amd/csrc/libosqp_amd.so", "/tmp/libosqp_amd_sys.so")
    lib = oq.load_library("/tmp/libosqp_amd_sys.so")
else:
    lib = oq.load_library()
for k in range(3):
    m = oq.Model(lib); t0=time.time(); oq.setup_generated(m, 0, 1000000, 1000, 1, linsys_solver="pcg", **bench.SETTINGS); t1=time.time()
    oq.clean(m); print("setup wall %.3f s, clean %.3f s" % (t1-t0, time.time()-t1), flush=True)
PY
cat > /tmp/alloc_bench.py <<'PY'
import sys, time, os, ctypes as C
sys.path.insert(0, sys.argv[1])
import osqp_jl_amd as oq
lib = oq.load_library()
import importlib.util
hip = C.CDLL(os.path.join(os.path.dirname(importlib.util.find_spec("torch").origin), "lib", "libamdhip64.so"))
def alloc(gb):
    p = C.c_void_p(); t0=time.time(); rc = lib.osqp_amd_device_alloc(C.byref(p), C.c_longlong(int(gb*1e9)), 0); assert rc == 0; return p, time.time()-t0
def free(p):
    t0=time.time(); lib.osqp_amd_device_free(p, 0); return time.time()-t0
def touch(p, gb):
    t0=time.time(); hip.hipMemset(p, 0, C.c_size_t(int(gb*1e9))); hip.hipDeviceSynchronize(); return time.time()-t0
for rep in range(2):
  for gb in (1, 4, 12):
    a, ta = alloc(gb); tt = touch(a, gb); tt2 = touch(a, gb); b, tb = alloc(gb); fa = free(a); c, tc = alloc(gb); tc1 = touch(c, gb); fb = free(b); fc = free(c)
    print(f"{gb:3d} GB: alloc {ta*1e3:7.1f} ms  first memset {tt*1e3:7.1f}  second memset {tt2*1e3:7.1f}  alloc#2 {tb*1e3:7.1f}  free {fa*1e3:7.1f}  alloc after free {tc*1e3:7.1f} (memset {tc1*1e3:7.1f})  free {fb*1e3:7.1f} {fc*1e3:7.1f}", flush=True)
hold, _ = alloc(30); touch(hold, 30)
ts=[]
for k in range(20):
    t0=time.time(); p,_ = alloc(4); t1=time.time(); free(p); ts.append((round((t1-t0)*1e3,1), round((time.time()-t1)*1e3,1)))
print("20 x (alloc 4 GB, free) while holding 30 GB (alloc ms, free ms):", ts)
ps=[]; ta=[]; tf=[]
for k in range(8):
    t0=time.time(); ps.append(alloc(4)[0]); ta.append(round((time.time()-t0)*1e3,1))
for p in ps:
    t0=time.time(); free(p); tf.append(round((time.time()-t0)*1e3,1))
print("8 x alloc 4 GB:", ta, " 8 x free:", tf)
PY
python /tmp/alloc_bench.py $GRAFT_REPO_ROOT 2>&1 | grep -v amdgpu.ids | tee $O/alloc_bench.txt
for r in 1 2; do
  echo "== torch's HIP runtime, process $r" | tee -a $O/setup_compare.txt
  OSQP_AMD_SETUP_TRACE=1 timeout 300 python /tmp/setup_only.py $GRAFT_REPO_ROOT 2>&1 | grep -E "setup" | tee -a $O/setup_compare.txt
  echo "== system HIP runtime, process $r" | tee -a $O/setup_compare.txt
  OSQP_AMD_SETUP_TRACE=1 timeout 300 python /tmp/setup_only.py $GRAFT_REPO_ROOT sys 2>&1 | grep -E "setup" | tee -a $O/setup_compare.txt
done
