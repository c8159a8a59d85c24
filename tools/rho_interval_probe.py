import sys, time; sys.path.insert(0,'/root/repo')
import osqp_jl_amd as oq, bench
lib = oq.load_library()
for wl in ("rand-1e5","rand-1e6"):
    kind,n,k,ls = bench.WORKLOADS[wl]
    for interval in (25,50,100):
        s = dict(bench.SETTINGS); s["adaptive_rho_interval"]=interval
        m = oq.Model(lib); oq.setup_generated(m, kind, n, k, 1, linsys_solver=ls, **s)
        t=time.time(); r=oq.solve(m); t=time.time()-t
        st=oq.stats(m)
        print(wl, interval, r.info.status, r.info.iter, "rho_updates", r.info.rho_updates, "cg", st[6], "time %.3f"%t, flush=True)
        oq.clean(m)
