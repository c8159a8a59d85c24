"""A small zoo of the standard QP classes OSQP is used for (portfolio, SVM, Huber fitting, lasso with a data
matrix, equality-constrained QP), generated with numpy's PCG64 from fixed seeds.  The reference's tests hold no
data for these; they widen the parity surface beyond the reference's own cases: the HIP engine is compared with
the CPU oracle and with an independent scipy evaluation of OSQP's stopping criteria."""
import numpy as np
import scipy.sparse as sp


def portfolio(n=1500, k=40, seed=1, gamma=1.0):
    """min x'Dx + y'y - mu'x / gamma  s.t.  y = F'x, 1'x = 1, x >= 0   (variables [x; y]; one dense row)"""
    rng = np.random.default_rng(seed)
    F = sp.random(n, k, density=0.5, random_state=rng, data_rvs=rng.standard_normal, format="csc")
    D = sp.diags(rng.random(n) * np.sqrt(k))
    mu = rng.standard_normal(n)
    P = sp.block_diag([2 * D, 2 * sp.eye(k)], format="csc")
    q = np.concatenate([-mu / gamma, np.zeros(k)])
    A = sp.vstack([
        sp.hstack([F.T, -sp.eye(k)]),
        sp.hstack([sp.csc_matrix(np.ones((1, n))), sp.csc_matrix((1, k))]),
        sp.hstack([sp.eye(n), sp.csc_matrix((n, k))]),
    ], format="csc")
    l = np.concatenate([np.zeros(k), [1.0], np.zeros(n)])
    u = np.concatenate([np.zeros(k), [1.0], np.ones(n)])
    return dict(P=P, q=q, A=A, l=l, u=u)


def svm(n=60, m=1200, seed=2, lam=1.0):
    """min x'x + lam 1't  s.t.  t >= diag(b) A x + 1, t >= 0   (variables [x; t])"""
    rng = np.random.default_rng(seed)
    half = m // 2
    b = np.concatenate([np.ones(half), -np.ones(m - half)])
    Ad = sp.vstack([
        sp.random(half, n, density=0.3, random_state=rng, data_rvs=lambda s: rng.standard_normal(s) / np.sqrt(n) + 1.0 / n),
        sp.random(m - half, n, density=0.3, random_state=rng, data_rvs=lambda s: rng.standard_normal(s) / np.sqrt(n) - 1.0 / n),
    ], format="csc")
    P = sp.block_diag([2 * sp.eye(n), sp.csc_matrix((m, m))], format="csc")
    q = np.concatenate([np.zeros(n), lam * np.ones(m)])
    A = sp.vstack([
        sp.hstack([sp.diags(b) @ Ad, -sp.eye(m)]),
        sp.hstack([sp.csc_matrix((m, n)), sp.eye(m)]),
    ], format="csc")
    l = np.concatenate([-np.inf * np.ones(m), np.zeros(m)])
    u = np.concatenate([-np.ones(m), np.inf * np.ones(m)])
    return dict(P=P, q=q, A=A, l=l, u=u)


def huber(n=50, m=1000, seed=3):
    """min u'u + 2 1'(r + s)  s.t.  A x - b - u = r - s, r, s >= 0   (variables [x; u; r; s])"""
    rng = np.random.default_rng(seed)
    Ad = sp.random(m, n, density=0.3, random_state=rng, data_rvs=rng.standard_normal, format="csc")
    x_true = rng.standard_normal(n) / np.sqrt(n)
    noise = rng.standard_normal(m)
    outliers = rng.random(m) < 0.05
    noise[outliers] *= 10.0
    b = Ad @ x_true + noise
    P = sp.block_diag([sp.csc_matrix((n, n)), 2 * sp.eye(m), sp.csc_matrix((2 * m, 2 * m))], format="csc")
    q = np.concatenate([np.zeros(n + m), 2 * np.ones(2 * m)])
    A = sp.vstack([
        sp.hstack([Ad, -sp.eye(m), -sp.eye(m), sp.eye(m)]),
        sp.hstack([sp.csc_matrix((2 * m, n + m)), sp.eye(2 * m)]),
    ], format="csc")
    l = np.concatenate([b, np.zeros(2 * m)])
    u = np.concatenate([b, np.inf * np.ones(2 * m)])
    return dict(P=P, q=q, A=A, l=l, u=u)


def lasso_data(n=80, m=800, seed=4):
    """min y'y / 2 + lam 1't  s.t.  y = A x - b, -t <= x <= t   (variables [x; y; t])"""
    rng = np.random.default_rng(seed)
    Ad = sp.random(m, n, density=0.3, random_state=rng, data_rvs=rng.standard_normal, format="csc")
    x_true = np.where(rng.random(n) < 0.5, 0.0, rng.standard_normal(n) / np.sqrt(n))
    b = Ad @ x_true + rng.standard_normal(m)
    lam = 0.2 * np.max(np.abs(Ad.T @ b))
    P = sp.block_diag([sp.csc_matrix((n, n)), sp.eye(m), sp.csc_matrix((n, n))], format="csc")
    q = np.concatenate([np.zeros(n + m), lam * np.ones(n)])
    A = sp.vstack([
        sp.hstack([Ad, -sp.eye(m), sp.csc_matrix((m, n))]),
        sp.hstack([sp.eye(n), sp.csc_matrix((n, m)), -sp.eye(n)]),
        sp.hstack([sp.eye(n), sp.csc_matrix((n, m)), sp.eye(n)]),
    ], format="csc")
    l = np.concatenate([b, -np.inf * np.ones(n), np.zeros(n)])
    u = np.concatenate([b, np.zeros(n), np.inf * np.ones(n)])
    return dict(P=P, q=q, A=A, l=l, u=u)


def equality_qp(n=800, seed=5):
    """min x'Px / 2 + q'x  s.t.  A x = b   (m = n / 2)"""
    rng = np.random.default_rng(seed)
    m = n // 2
    M = sp.random(n, n, density=0.02, random_state=rng, data_rvs=rng.standard_normal, format="csc")
    P = (M @ M.T + 1e-2 * sp.eye(n)).tocsc()
    q = rng.standard_normal(n)
    A = sp.random(m, n, density=0.03, random_state=rng, data_rvs=rng.standard_normal, format="csc")
    b = A @ rng.standard_normal(n)
    return dict(P=P, q=q, A=A, l=b, u=b.copy())


def control(nx=8, nu=4, T=30, seed=6):
    """Linear MPC: min sum x_t'Q x_t + u_t'R u_t  s.t.  x_{t+1} = A x_t + B u_t, x_0 given, box bounds
    (variables [x_0 .. x_T; u_0 .. u_{T-1}]; banded KKT system, elimination tree = a long chain)"""
    rng = np.random.default_rng(seed)
    Ad = np.eye(nx) + 0.1 * rng.standard_normal((nx, nx))
    Ad *= 0.95 / max(1.0, np.max(np.abs(np.linalg.eigvals(Ad))))
    Bd = rng.standard_normal((nx, nu))
    Q = sp.diags(rng.random(nx) * 10.0)
    R = 0.1 * sp.eye(nu)
    x0 = rng.standard_normal(nx)
    P = sp.block_diag([sp.kron(sp.eye(T + 1), Q), sp.kron(sp.eye(T), R)], format="csc")
    q = np.zeros((T + 1) * nx + T * nu)
    Ax = sp.kron(sp.eye(T + 1), -sp.eye(nx)) + sp.kron(sp.eye(T + 1, k=-1), sp.csc_matrix(Ad))
    Bu = sp.kron(sp.vstack([sp.csc_matrix((1, T)), sp.eye(T)]), sp.csc_matrix(Bd))
    Aeq = sp.hstack([Ax, Bu])
    leq = np.concatenate([-x0, np.zeros(T * nx)])
    A = sp.vstack([Aeq, sp.eye((T + 1) * nx + T * nu)], format="csc")
    lo = np.concatenate([-5.0 * np.ones((T + 1) * nx), -0.5 * np.ones(T * nu)])
    return dict(P=P, q=q, A=A, l=np.concatenate([leq, lo]), u=np.concatenate([leq, -lo]))


def grid2d(g=24, seed=7, diag=0.1):
    """A QP on a g x g grid (a structure the direct back-end was not tuned on: a 2-D graph, no chain to unroll, separators of
    ~g nodes): P = the 5-point Laplacian + diag * I (a discretised Dirichlet energy), box constraints on every variable
    (A = I, n = m = g^2), a random load q.  The KKT graph is the grid with one pendant node per variable."""
    rng = np.random.default_rng(seed)
    n = g * g
    T = sp.diags([-np.ones(g - 1), 2.0 * np.ones(g), -np.ones(g - 1)], [-1, 0, 1])
    P = (sp.kron(sp.eye(g), T) + sp.kron(T, sp.eye(g)) + diag * sp.eye(n)).tocsc()
    q = rng.standard_normal(n)
    A = sp.eye(n, format="csc")
    b = 0.5 + rng.random(n)
    return dict(P=P, q=q, A=A, l=-b, u=b)


def grid3d(g=10, seed=8, diag=0.1, coupled_rows=True):
    """A QP on a g x g x g grid (separators of ~g^2 nodes: fronts far beyond one workgroup's LDS already at g = 12): P = the 7-point
    Laplacian + diag * I; constraints: a box on every variable and -- `coupled_rows` -- a band on the difference of every pair of
    neighbours along the first axis (rows of A with two entries: the KKT graph is not just the grid with pendant nodes)."""
    rng = np.random.default_rng(seed)
    n = g ** 3
    T = sp.diags([-np.ones(g - 1), 2.0 * np.ones(g), -np.ones(g - 1)], [-1, 0, 1])
    I = sp.eye(g)
    P = (sp.kron(sp.kron(T, I), I) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(I, I), T) + diag * sp.eye(n)).tocsc()
    q = rng.standard_normal(n)
    rows = [sp.eye(n, format="csr")]
    lo, up = [-(0.5 + rng.random(n))], [0.5 + rng.random(n)]
    if coupled_rows:
        D = sp.diags([-np.ones(g - 1), np.ones(g - 1)], [0, 1], shape=(g - 1, g))
        G = sp.kron(sp.kron(D, I), I).tocsr()
        rows.append(G)
        lo.append(-0.3 * np.ones(G.shape[0])); up.append(0.3 * np.ones(G.shape[0]))
    A = sp.vstack(rows).tocsc()
    return dict(P=P, q=q, A=A, l=np.concatenate(lo), u=np.concatenate(up))


ZOO = {"grid2d": grid2d, "grid3d": grid3d, "control": control, "portfolio": portfolio, "svm": svm, "huber": huber, "lasso_data": lasso_data, "equality_qp": equality_qp}


def kkt_check(prob, x, y, eps):
    """OSQP's own stopping criteria re-evaluated in numpy on the unscaled data; returns (pri, eps_pri, dua, eps_dua)."""
    P, q, A, l, u = prob["P"], prob["q"], prob["A"], prob["l"], prob["u"]
    Pfull = P if (sp.tril(P, -1).nnz > 0) else (P + sp.triu(P, 1).T)
    Ax = A @ x
    z = np.clip(Ax, l, u)
    Px, Aty = Pfull @ x, A.T @ y
    pri = np.max(np.abs(Ax - z)) if A.shape[0] else 0.0
    dua = np.max(np.abs(Px + q + Aty))
    eps_pri = eps + eps * max(np.max(np.abs(Ax)), np.max(np.abs(z)))
    eps_dua = eps + eps * max(np.max(np.abs(Px)), np.max(np.abs(Aty)), np.max(np.abs(q)))
    return pri, eps_pri, dua, eps_dua
