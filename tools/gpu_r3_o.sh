#!/bin/bash
# setup wall time: current library against two earlier commits of this round, on one box, no profiler; hipMalloc / hipFree cost
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3o; mkdir -p $O
cat > /tmp/setup_only.py <<'PY'
import sys, time
sys.path.insert(0, sys.argv[1])
import osqp_jl_amd as oq, bench
lib = oq.load_library()
for k in range(2):
    m = oq.Model(lib); t0=time.time(); oq.setup_generated(m, 0, 1000000, 1000, 1, linsys_solver="pcg", **bench.SETTINGS); print("setup wall", time.time()-t0, flush=True)
    oq.clean(m)
PY
cat > /tmp/alloc_bench.py <<'PY'
import sys, time, ctypes as C
sys.path.insert(0, sys.argv[1])
import osqp_jl_amd as oq
lib = oq.load_library()
hip = C.CDLL("libamdhip64.so")
def alloc(gb):
    p = C.c_void_p(); t0=time.time(); rc = lib.osqp_amd_device_alloc(C.byref(p), C.c_longlong(int(gb*1e9)), 0); return p, time.time()-t0
def free(p):
    t0=time.time(); lib.osqp_amd_device_free(p, 0); return time.time()-t0
def touch(p, gb):
    t0=time.time(); hip.hipMemset(p, 0, C.c_size_t(int(gb*1e9))); hip.hipDeviceSynchronize(); return time.time()-t0
for gb in (1, 4, 12):
    a, ta = alloc(gb); tt = touch(a, gb); tt2 = touch(a, gb); b, tb = alloc(gb); fa = free(a); c, tc = alloc(gb); tc1 = touch(c, gb); fb = free(b); fc = free(c)
    print(f"{gb:3d} GB: alloc {ta*1e3:7.1f} ms  first memset {tt*1e3:7.1f}  second memset {tt2*1e3:7.1f}  alloc#2 {tb*1e3:7.1f}  free {fa*1e3:7.1f}  alloc after free {tc*1e3:7.1f} (memset {tc1*1e3:7.1f})  free {fb*1e3:7.1f} {fc*1e3:7.1f}", flush=True)
# a setup-like sequence: hold 30 GB, then 20 cycles of alloc 4 GB / free 4 GB
hold, _ = alloc(30); touch(hold, 30)
t0=time.time()
for k in range(20):
    p,_ = alloc(4); free(p)
print("20 x (alloc 4 GB, free) while holding 30 GB:", (time.time()-t0)*1e3, "ms")
ps=[]; t0=time.time()
for k in range(8): ps.append(alloc(4)[0])
t1=time.time()
for p in ps: free(p)
print("8 x alloc 4 GB:", (t1-t0)*1e3, "ms; 8 x free:", (time.time()-t1)*1e3, "ms")
PY
python /tmp/alloc_bench.py $GRAFT_REPO_ROOT 2>&1 | grep -v amdgpu.ids | tee $O/alloc_bench.txt
for L in HEAD 148ec9c f95e1a9 HEAD; do
  echo "== library $L" | tee -a $O/setup_compare.txt
  if [ $L = HEAD ]; then unset OSQP_AMD_LIB; else export OSQP_AMD_LIB=$GRAFT_REPO_ROOT/build_cmp/lib_$L.so; fi
  OSQP_AMD_SETUP_TRACE=1 timeout 300 python /tmp/setup_only.py $GRAFT_REPO_ROOT 2>&1 | grep -E "setup" | tee -a $O/setup_compare.txt
done
