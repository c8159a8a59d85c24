import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import osqp_jl_amd as oq, qp_zoo
for prob, ls in ((qp_zoo.control(nx=8,nu=4,T=400), "qdldl"), (qp_zoo.grid2d(40), "pcg")):
    m = oq.Model(oq.load_library())
    oq.setup(m, linsys_solver=ls, verbose=True, max_iter=50, **prob)
    oq.clean(m)
