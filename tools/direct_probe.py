import sys, time; sys.path.insert(0,'/root/repo')
import osqp_jl_amd as oq
lib = oq.load_library()
for (kind,n,k) in ((0,2000,20),(0,5000,10),(1,200000,0)):
    m = oq.Model(lib)
    t=time.time(); oq.setup_generated(m, kind, n, k, 1, linsys_solver="direct", verbose=False, eps_abs=1e-5, eps_rel=1e-5, adaptive_rho_interval=25); ts=time.time()-t
    st = oq.stats(m)
    t=time.time(); r = oq.solve(m); tv=time.time()-t
    t=time.time(); oq.update_settings(m, rho=0.3); tr=time.time()-t
    print(dict(kind=kind,n=n,k=k,setup_s=round(ts,3),nnzL=st[4],levels=st[5],solve_s=round(tv,3),iters=r.info.iter,rho_updates=r.info.rho_updates,refactor_s=round(tr,4), status=r.info.status))
