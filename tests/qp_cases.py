"""Known-answer and property cases restated from the reference's test suite
(SURVEY.md section 4 and Appendix B), written against the host mirror so that the
same function checks the CPU oracle and -- through the C ABI -- the HIP engine.

Every case takes ``(oq, lib, linsys)``: the package, a loaded ABI library and the
linear-system back-end name ("qdldl" = direct LDL', "pcg" = indirect).
"""
import json
import math
import os

import numpy as np
import scipy.sparse as sp

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KA = json.load(open(os.path.join(GOLDEN, "known_answers.json")))


def _vec(v):
    return np.array([float(e) for e in v])


def _mat(rows):
    return sp.csc_matrix(np.array(rows, dtype=float))


def basic_problem():
    b = KA["basic"]
    return dict(P=_mat(b["P"]), q=_vec(b["q"]), A=_mat(b["A"]), l=_vec(b["l"]), u=_vec(b["u"])), dict(b["options"])


def _setup(oq, lib, linsys, prob, opts):
    m = oq.Model(lib)
    oq.setup(m, linsys_solver=linsys, **prob, **opts)
    return m


def _close(a, b, tol):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) <= tol


# ----------------------------------------------------------------- test/basic.jl
def case_basic_qp(oq, lib, linsys):  # G1 [REF test/basic.jl:28-50]
    prob, opts = basic_problem()
    m = _setup(oq, lib, linsys, prob, opts)
    r = oq.solve(m)
    g = KA["G1"]
    assert r.info.status == "Solved"
    assert _close(r.x, g["x"], g["tol"]) and _close(r.y, g["y"], g["tol"])
    assert abs(r.info.obj_val - g["obj"]) <= g["tol"]
    return r


def case_update_q(oq, lib, linsys):  # G2 [REF test/basic.jl:52-76]
    prob, opts = basic_problem()
    m = _setup(oq, lib, linsys, prob, opts)
    g = KA["G2"]
    oq.update(m, q=_vec(g["q"]))
    r = oq.solve(m)
    assert _close(r.x, g["x"], g["tol"]) and _close(r.y, g["y"], g["tol"])
    assert abs(r.info.obj_val - g["obj"]) <= g["tol"]


def case_update_l(oq, lib, linsys):  # G3 [REF test/basic.jl:78-102]
    prob, opts = basic_problem()
    m = _setup(oq, lib, linsys, prob, opts)
    g = KA["G3"]
    oq.update(m, l=_vec(g["l"]))
    r = oq.solve(m)
    assert _close(r.x, g["x"], g["tol"]) and _close(r.y, g["y"], g["tol"])
    assert abs(r.info.obj_val - g["obj"]) <= g["tol"]


def case_update_u(oq, lib, linsys):  # G4 [REF test/basic.jl:104-132]
    prob, opts = basic_problem()
    m = _setup(oq, lib, linsys, prob, opts)
    g = KA["G4"]
    oq.update(m, u=_vec(g["u"]))
    r = oq.solve(m)
    assert _close(r.x, g["x"], g["tol"]) and _close(r.y, g["y"], g["tol"])
    assert abs(r.info.obj_val - g["obj"]) <= g["tol"]


def case_update_max_iter(oq, lib, linsys):  # G5 [REF test/basic.jl:134-152]
    prob, opts = basic_problem()
    m = _setup(oq, lib, linsys, prob, opts)
    oq.update_settings(m, max_iter=KA["G5"]["max_iter"])
    r = oq.solve(m)
    assert r.info.status == "Max_iter_reached"
    assert r.info.iter == 80


def case_update_check_termination(oq, lib, linsys):  # G6 [REF test/basic.jl:154-172]
    prob, opts = basic_problem()
    m = _setup(oq, lib, linsys, prob, opts)
    oq.update_settings(m, check_termination=False)
    r = oq.solve(m)
    assert r.info.iter == opts["max_iter"]


def case_update_rho(oq, lib, linsys):  # G7 [REF test/basic.jl:174-208]
    prob, opts = basic_problem()
    m = _setup(oq, lib, linsys, prob, opts)
    r_default = oq.solve(m)
    opts2 = dict(opts)
    opts2["rho"] = KA["G7"]["rho_setup"]
    m2 = _setup(oq, lib, linsys, prob, opts2)
    oq.update_settings(m2, rho=KA["G7"]["rho_update"])
    r_new = oq.solve(m2)
    assert r_default.info.iter == r_new.info.iter


def case_time_limit(oq, lib, linsys):  # G8 [REF test/basic.jl:210-240]
    prob, opts = basic_problem()
    m = _setup(oq, lib, linsys, prob, opts)
    r = oq.solve(m)
    assert r.info.status == "Solved"
    g = KA["G8"]
    oq.update_settings(m, eps_abs=g["eps_abs"], eps_rel=g["eps_rel"], time_limit=g["time_limit"],
                       max_iter=g["max_iter"], check_termination=g["check_termination"])
    r = oq.solve(m)
    assert r.info.status == "Time_limit_reached"
    # the Julia layer NaN-fills x,y for this status [REF src/interface.jl:184-197, src/constants.jl:23]
    assert np.all(np.isnan(r.x))


# ----------------------------------------------------------------- test/polishing.jl
def case_polish_basic(oq, lib, linsys):  # G9 [REF test/polishing.jl:16-38]
    prob, _ = basic_problem()
    g = KA["G9"]
    m = _setup(oq, lib, linsys, prob, g["options"])
    r = oq.solve(m)
    assert np.allclose(r.x, g["x"], atol=g["tol"]) and np.allclose(r.y, g["y"], atol=g["tol"])
    assert abs(r.info.obj_val - g["obj"]) <= g["tol"]
    assert r.info.status_polish == 1


def case_polish_unconstrained(oq, lib, linsys):  # [REF test/polishing.jl:40-67]
    rng = np.random.default_rng(1)
    n = 10
    P = sp.diags(rng.random(n) + 0.2).tocsc()
    q = rng.standard_normal(n)
    A = sp.identity(n, format="csc")
    l, u = -100 * np.ones(n), 100 * np.ones(n)
    m = _setup(oq, lib, linsys, dict(P=P, q=q, A=A, l=l, u=u), KA["G9"]["options"])
    r = oq.solve(m)
    invP = np.linalg.inv(P.toarray())
    assert np.allclose(r.x, -invP @ q, atol=1e-3)
    assert np.allclose(r.y, 0.0, atol=1e-3)
    assert abs(r.info.obj_val - (-0.5 * q @ invP @ q)) <= 1e-3
    assert r.info.status_polish == 1


def load_polish_fixture():
    fx = json.load(open(os.path.join(GOLDEN, "random_polish_qp.json")))
    P = sp.csc_matrix((fx["P"]["x"], fx["P"]["i"], fx["P"]["p"]), shape=(fx["P"]["m"], fx["P"]["n"]))
    A = sp.csc_matrix((fx["A"]["x"], fx["A"]["i"], fx["A"]["p"]), shape=(fx["A"]["m"], fx["A"]["n"]))
    return fx, dict(P=P, q=_vec(fx["q"]), A=A, l=_vec(fx["l"]), u=_vec(fx["u"]))


def case_polish_random(oq, lib, linsys):  # G10 [REF test/polishing.jl:69-93]
    fx, prob = load_polish_fixture()
    m = _setup(oq, lib, linsys, prob, KA["G9"]["options"])
    r = oq.solve(m)
    assert np.allclose(r.x, fx["x_test"], atol=1e-3)
    assert np.allclose(r.y, fx["y_test"], atol=1e-3)
    assert abs(r.info.obj_val - fx["obj_test"]) <= 1e-3
    assert r.info.status_polish == 1


# ----------------------------------------------------------------- test/non_convex.jl
def case_non_convex_small_sigma(oq, lib, linsys):  # G11 [REF test/non_convex.jl:3-22]
    prob, _ = basic_problem()
    prob["P"] = _mat(KA["G11"]["P"])
    failed = False
    try:
        _setup(oq, lib, linsys, prob, dict(verbose=False, sigma=KA["G11"]["sigma"]))
    except oq.OSQPError:
        failed = True
    assert failed


def case_non_convex_big_sigma(oq, lib, linsys):  # G12 [REF test/non_convex.jl:24-41]
    prob, _ = basic_problem()
    prob["P"] = _mat(KA["G12"]["P"])
    m = _setup(oq, lib, linsys, prob, dict(verbose=False, sigma=KA["G12"]["sigma"]))
    r = oq.solve(m)
    assert math.isnan(r.info.obj_val)
    assert r.info.status == "Non_convex"


# ----------------------------------------------------------------- infeasibility
def _case_from(g):
    return dict(P=_mat(g["P"]), q=_vec(g["q"]), A=_mat(g["A"]), l=_vec(g["l"]), u=_vec(g["u"]))


def case_dual_infeasible_lp(oq, lib, linsys):  # G13 [REF test/dual_infeasibility.jl:15-28]
    m = _setup(oq, lib, linsys, _case_from(KA["G13"]), KA["dual_inf_options"])
    r = oq.solve(m)
    assert r.info.status == "Dual_infeasible"
    assert np.all(np.isnan(r.x)) and np.all(np.isfinite(r.dual_inf_cert))


def case_dual_infeasible_qp(oq, lib, linsys):  # G14 [REF test/dual_infeasibility.jl:30-43]
    m = _setup(oq, lib, linsys, _case_from(KA["G14"]), KA["dual_inf_options"])
    assert oq.solve(m).info.status == "Dual_infeasible"


def case_primal_dual_infeasible_warm(oq, lib, linsys):  # G15 [REF test/dual_infeasibility.jl:45-62]
    g = KA["G15"]
    m = _setup(oq, lib, linsys, _case_from(g), KA["dual_inf_options"])
    oq.warm_start(m, x=_vec(g["warm_x"]), y=_vec(g["warm_y"]))
    assert oq.solve(m).info.status == "Dual_infeasible"


def case_primal_dual_infeasible_cold(oq, lib, linsys):  # G16 [REF test/primal_infeasibility.jl:41-59]
    m = _setup(oq, lib, linsys, _case_from(KA["G16"]), KA["prim_inf_options"])
    r = oq.solve(m)
    assert r.info.status == "Primal_infeasible"
    assert np.all(np.isnan(r.y)) and np.all(np.isfinite(r.prim_inf_cert))


def case_primal_infeasible_random(oq, lib, linsys):  # [REF test/primal_infeasibility.jl:15-39], own RNG
    rng = np.random.default_rng(1)
    n, mm = 50, 500
    Pm = sp.random(n, n, 0.6, random_state=rng, data_rvs=rng.standard_normal)
    P = (Pm.T @ Pm).tocsc()
    q = rng.standard_normal(n)
    A = sp.random(mm, n, 0.6, random_state=rng, data_rvs=rng.standard_normal).tolil()
    u = 3 + rng.standard_normal(mm)
    l = -3 + rng.standard_normal(mm)
    k = n // 2
    A[k - 1, :] = A[k, :]
    l[k - 1] = u[k] + 10 * rng.random()
    u[k - 1] = l[k - 1] + 0.5
    m = _setup(oq, lib, linsys, dict(P=P, q=q, A=A.tocsc(), l=l, u=u), KA["prim_inf_options"])
    assert oq.solve(m).info.status == "Primal_infeasible"


# ----------------------------------------------------------------- closed forms
def case_unconstrained(oq, lib, linsys):  # [REF test/unconstrained.jl:14-41]
    rng = np.random.default_rng(1)
    n = 30
    P = sp.diags(rng.random(n) + 0.2).tocsc()
    q = rng.standard_normal(n)
    A = sp.csc_matrix((0, n))
    m = _setup(oq, lib, linsys, dict(P=P, q=q, A=A, l=np.zeros(0), u=np.zeros(0)),
               dict(verbose=False, eps_abs=1e-8, eps_rel=1e-8, eps_dual_inf=1e-18))
    r = oq.solve(m)
    invP = np.linalg.inv(P.toarray())
    assert np.allclose(r.x, -invP @ q, atol=1e-5)
    assert len(r.y) == 0
    assert abs(r.info.obj_val - (-0.5 * q @ invP @ q)) <= 1e-5
    assert r.info.status == "Solved"


def case_feasibility(oq, lib, linsys):  # [REF test/feasibility.jl:14-29]
    rng = np.random.default_rng(3)
    n = mm = 30
    A = sp.random(mm, n, 0.8, random_state=rng, data_rvs=rng.standard_normal).tocsc()
    u = rng.standard_normal(mm)
    m = _setup(oq, lib, linsys, dict(P=sp.csc_matrix((n, n)), q=np.zeros(n), A=A, l=u.copy(), u=u),
               dict(verbose=False, eps_abs=1e-6, eps_rel=1e-6, max_iter=5000))
    r = oq.solve(m)
    assert np.linalg.norm(A @ r.x - u) <= 1e-3


def warm_start_problem():
    rng = np.random.default_rng(1)
    n, mm = 100, 200
    Pm = sp.random(n, n, 0.9, random_state=rng, data_rvs=rng.standard_normal)
    P = (Pm.T @ Pm).tocsc()
    q = rng.standard_normal(n)
    A = sp.random(mm, n, 0.9, random_state=rng, data_rvs=rng.standard_normal).tocsc()
    u = rng.random(mm) * 2
    l = -rng.random(mm) * 2
    return dict(P=P, q=q, A=A, l=l, u=u)


def case_warm_start(oq, lib, linsys):  # [REF test/warm_start.jl:16-48]
    prob = warm_start_problem()
    n, mm = 100, 200
    m = _setup(oq, lib, linsys, prob, dict(verbose=False, eps_abs=1e-8, eps_rel=1e-8, polish=False,
                                             adaptive_rho=False, check_termination=1))
    r = oq.solve(m)
    assert r.info.status == "Solved"
    x_opt, y_opt, tot_iter = r.x.copy(), r.y.copy(), r.info.iter
    oq.warm_start(m, x=np.zeros(n), y=np.zeros(mm))
    r = oq.solve(m)
    assert r.info.iter == tot_iter
    oq.warm_start(m, x=x_opt, y=y_opt)
    r = oq.solve(m)
    assert r.info.iter <= 10


def case_moi_lp(oq, lib, linsys):  # G18 [REF test/MOI_wrapper.jl:280-332]
    g = KA["G18"]
    m = _setup(oq, lib, linsys, _case_from(g), g["options"])
    r = oq.solve(m)
    assert r.info.status == "Solved"
    assert np.allclose(r.x, g["x"], atol=g["tol"], rtol=g["tol"])
    assert abs(r.info.obj_val - g["obj"]) <= g["tol"]
    # MOI's dual is minus OSQP's y [REF src/MOI_wrapper.jl:488, 762, 884]
    assert np.allclose(-r.y, g["moi_duals"], atol=g["tol"], rtol=g["tol"])


def case_equality_lsq(oq, lib, linsys):  # [REF test/MOI_wrapper.jl:694-790], own RNG
    rng = np.random.default_rng(1234)
    n, mm = 8, 2
    for _ in range(11):
        Am, b, Cm, d = rng.random((n, n)), rng.random(n), rng.random((mm, n)), rng.random(mm)
        Cp = np.linalg.pinv(Cm)
        Q = np.eye(n) - Cp @ Cm
        expected = Q @ (np.linalg.pinv(Am @ Q) @ (b - Am @ Cp @ d)) + Cp @ d
        P = sp.csc_matrix(np.triu(2 * Am.T @ Am))  # 1/2 x'Px = x'A'Ax
        q = -2 * Am.T @ b
        m = _setup(oq, lib, linsys, dict(P=P, q=q, A=sp.csc_matrix(Cm), l=d.copy(), u=d.copy()),
                   dict(verbose=False, eps_abs=1e-8, eps_rel=1e-16, max_iter=10000, adaptive_rho_interval=25))
        r = oq.solve(m)
        assert r.info.status == "Solved"
        assert np.allclose(r.x, expected, atol=1e-4)
        assert abs(r.info.obj_val + b @ b - np.linalg.norm(Am @ expected - b) ** 2) <= 1e-4


# ----------------------------------------------------------------- update_P / update_A
def case_update_matrices(oq, lib, linsys):
    """update_P / update_A / update_P_A with and without index vectors must equal a
    fresh setup on the modified data (what [REF test/update_matrices.jl] intends;
    its expectations are dead code on Julia >= 1.1)."""
    rng = np.random.default_rng(7)
    n, mm = 5, 8
    p = 0.7
    Pt = sp.random(n, n, p, random_state=rng, data_rvs=rng.standard_normal)
    P = (Pt @ Pt.T + sp.identity(n)).tocsc()
    Ptn = Pt.copy()
    Ptn.data = Ptn.data + 0.1 * rng.standard_normal(Ptn.nnz)
    P_new = (Ptn @ Ptn.T + sp.identity(n)).tocsc()  # same pattern, new values
    q = rng.standard_normal(n)
    A = sp.random(mm, n, p, random_state=rng, data_rvs=rng.standard_normal).tocsc()
    A_new = A.copy()
    A_new.data = A_new.data + rng.standard_normal(A_new.nnz)
    l, u = np.zeros(mm), 30 + rng.standard_normal(mm)
    opts = dict(verbose=False, eps_abs=1e-8, eps_rel=1e-8, polish=False, check_termination=1,
                adaptive_rho_interval=25, max_iter=20000)
    Pu, Pu_new = sp.triu(P, format="csc"), sp.triu(P_new, format="csc")
    assert np.array_equal(Pu.indices, Pu_new.indices)

    def fresh(Pm, Am):
        return oq.solve(_setup(oq, lib, linsys, dict(P=Pm, q=q, A=Am, l=l, u=u), opts))

    ref_P, ref_A, ref_PA = fresh(P_new, A), fresh(P, A_new), fresh(P_new, A_new)
    for use_idx in (False, True):
        pidx = np.arange(Pu.nnz) if use_idx else None
        aidx = np.arange(A.nnz) if use_idx else None
        m = _setup(oq, lib, linsys, dict(P=P, q=q, A=A, l=l, u=u), opts)
        oq.update(m, Px=Pu_new.data, Px_idx=pidx)
        r = oq.solve(m)
        assert np.allclose(r.x, ref_P.x, atol=1e-5) and np.allclose(r.y, ref_P.y, atol=1e-5)
        m = _setup(oq, lib, linsys, dict(P=P, q=q, A=A, l=l, u=u), opts)
        oq.update(m, Ax=A_new.data, Ax_idx=aidx)
        r = oq.solve(m)
        assert np.allclose(r.x, ref_A.x, atol=1e-5) and np.allclose(r.y, ref_A.y, atol=1e-5)
        m = _setup(oq, lib, linsys, dict(P=P, q=q, A=A, l=l, u=u), opts)
        oq.update(m, Px=Pu_new.data, Px_idx=pidx, Ax=A_new.data, Ax_idx=aidx)
        r = oq.solve(m)
        assert np.allclose(r.x, ref_PA.x, atol=1e-5) and np.allclose(r.y, ref_PA.y, atol=1e-5)
    # partial index update: only the first half of A's nnz
    half = np.arange(A.nnz // 2)
    A_half = A.copy()
    A_half.data[half] = A_new.data[half]
    ref_half = fresh(P, A_half)
    m = _setup(oq, lib, linsys, dict(P=P, q=q, A=A, l=l, u=u), opts)
    oq.update(m, Ax=A_new.data[half], Ax_idx=half)
    r = oq.solve(m)
    assert np.allclose(r.x, ref_half.x, atol=1e-5)


# ----------------------------------------------------------------- interface
def case_interface(oq, lib, linsys):  # [REF test/interface.jl:4-18]
    jl = sp.identity(5, format="csc")
    mc = oq.ManagedCcsc(jl)
    jl2 = oq.ccsc_to_scipy(mc.ccsc)
    assert (jl != jl2).nnz == 0
    model = oq.Model(lib)
    try:
        oq.solve(model)
        raise AssertionError("solve on an empty model must throw")
    except oq.OSQPError:
        pass
    oq.clean(model)  # osqp_cleanup(NULL) must succeed [REF src/interface.jl:24-25, 223-229]
    assert oq.version(lib) == "0.6.2"


def case_bounds_validation(oq, lib, linsys):
    """l > u is rejected at setup and by the update entry points."""
    prob, opts = basic_problem()
    bad = dict(prob)
    bad["l"] = np.array([1.0, -np.inf, -np.inf, -np.inf, -np.inf])
    bad["u"] = np.array([0.0, 0.0, -15, 100, 80])
    try:
        _setup(oq, lib, linsys, bad, opts)
        raise AssertionError("setup must reject l > u")
    except oq.OSQPError:
        pass
    m = _setup(oq, lib, linsys, prob, opts)
    try:
        oq.update_bounds(m, np.ones(5), np.zeros(5))
        raise AssertionError("update_bounds must reject l > u")
    except oq.OSQPError:
        pass
    try:
        oq.update_settings(m, alpha=2.5)
        raise AssertionError("alpha out of range must be rejected")
    except oq.OSQPError:
        pass
    try:
        oq.update_settings(m, scaled_termination=1)  # not updatable in the reference either
        raise AssertionError
    except oq.OSQPError:
        pass


ALL_CASES = [v for k, v in sorted(globals().items()) if k.startswith("case_")]
# cases whose expectation relies on a setup-time factorisation (inertia check) or on polish
DIRECT_ONLY = {"case_non_convex_small_sigma"}


# ----------------------------------------------------------------- modification caches (row N2)
def case_problem_modification_cache(oq, lib, linsys):  # [REF test/MOI_wrapper.jl:95-205], own RNG
    from osqp_jl_amd import modcaches as mc

    rng = np.random.default_rng(1234)
    n, mm = 15, 10
    q = rng.standard_normal(n)
    X = sp.random(n, n, 0.1, random_state=rng, data_rvs=rng.standard_normal)
    P = (X.T @ X + 2.220446049250313e-16 * sp.identity(n)).tocsc()
    l, u = -rng.random(mm), rng.random(mm)
    A = sp.random(mm, n, 0.6, random_state=rng, data_rvs=rng.standard_normal).tocsc()
    A.sort_indices()
    opts = dict(verbose=False, eps_abs=1e-8, eps_rel=1e-16, max_iter=20000, adaptive_rho_interval=25)
    modcache = mc.ProblemModificationCache(P, q, A, l, u)
    model = _setup(oq, lib, linsys, dict(P=P, q=q, A=A, l=l, u=u), opts)
    base = oq.solve(model)
    assert base.info.status == "Solved"
    # modify q
    assert not modcache.q.dirty
    modcache.q[2] = 5.0
    assert modcache.q.dirty and modcache.q.data[2] == 5.0
    modcache.processupdates(model)
    assert not modcache.q.dirty
    r_upd = oq.solve(model)
    assert not np.allclose(base.x, r_upd.x, atol=1e-1)
    r_set = oq.solve(_setup(oq, lib, linsys, dict(P=P, q=modcache.q.data, A=A, l=l, u=u), opts))
    assert np.allclose(r_upd.x, r_set.x, atol=1e-7)
    # modify one entry of A
    Acoo = A.tocoo()
    k = int(rng.integers(0, A.nnz))
    row, col, val = int(Acoo.row[k]), int(Acoo.col[k]), float(rng.standard_normal())
    modcache.A[row, col] = val
    modcache.processupdates(model)
    assert not modcache.A.modifications
    r_upd2 = oq.solve(model)
    Amod = A.tolil(); Amod[row, col] = val; Amod = Amod.tocsc()
    r_set2 = oq.solve(_setup(oq, lib, linsys, dict(P=P, q=modcache.q.data, A=Amod, l=l, u=u), opts))
    assert np.allclose(r_upd2.x, r_set2.x, atol=1e-7)
    # colon indexing on P: zero everything, then put ones on the diagonal (the pattern has a full diagonal)
    modcache.P[:] = 0.0
    for i in range(n):
        modcache.P[i, i] = 1.0
    modcache.processupdates(model)
    r_upd3 = oq.solve(model)
    r_set3 = oq.solve(_setup(oq, lib, linsys, dict(P=sp.identity(n, format="csc"), q=modcache.q.data, A=Amod, l=l, u=u), opts))
    assert np.allclose(r_upd3.x, r_set3.x, atol=1e-7)
    # both bounds dirty: flushed together
    modcache.l[:] = l - 1.0
    modcache.u[:] = u + 1.0
    modcache.processupdates(model)
    assert not modcache.l.dirty and not modcache.u.dirty
    r_upd4 = oq.solve(model)
    r_set4 = oq.solve(_setup(oq, lib, linsys, dict(P=sp.identity(n, format="csc"), q=modcache.q.data, A=Amod, l=l - 1.0, u=u + 1.0), opts))
    assert np.allclose(r_upd4.x, r_set4.x, atol=1e-7)
    # changing the sparsity pattern is refused
    nz = set(zip(Acoo.row.tolist(), Acoo.col.tolist()))
    for i in range(mm):
        for j in range(n):
            if (i, j) not in nz:
                try:
                    modcache.A[i, j] = 1.0
                    raise AssertionError("pattern change must be refused")
                except ValueError:
                    pass
    try:
        modcache.A[:] = 1
        raise AssertionError
    except ValueError:
        pass
    # warm-start cache + optimize(): the second optimize starts from the first solution
    ws = mc.WarmStartCache(n, mm)
    model5 = _setup(oq, lib, linsys, dict(P=P, q=q, A=A, l=l, u=u), opts)
    mod5 = mc.ProblemModificationCache(P, q, A, l, u)
    r1 = mc.optimize(model5, mod5, ws)
    ws.x[:] = r1.x; ws.y[:] = r1.y
    r2 = mc.optimize(model5, mod5, ws)
    assert r2.info.iter <= r1.info.iter and np.allclose(r1.x, r2.x, atol=1e-6)


ALL_CASES = [v for k, v in sorted(globals().items()) if k.startswith("case_")]


def case_ragged_structure(oq, lib, linsys):
    """Edge cases of the sparse structure: empty rows and columns in A, an empty column in P (no diagonal
    entry), a row of A repeated, unsorted row indices inside the columns handed to the C ABI, infinite
    bounds on one side.  Checked against an independent KKT evaluation on the host."""
    rng = np.random.default_rng(11)
    n, mm = 12, 9
    P = np.zeros((n, n))
    for j in (0, 2, 3, 7, 8, 11):
        P[j, j] = 1.0 + rng.random()
    P[0, 3] = P[3, 0] = 0.3
    P[2, 8] = P[8, 2] = -0.2
    A = np.zeros((mm, n))
    for i in (0, 1, 2, 4, 5, 7):
        cols = rng.choice(n - 2, size=3, replace=False)  # columns 10 and 11 stay empty
        A[i, cols] = rng.standard_normal(3)
    A[8, :] = A[0, :]          # repeated row; rows 3 and 6 stay empty
    l = -1.0 - rng.random(mm); u = 1.0 + rng.random(mm)
    l[1] = -np.inf; u[4] = np.inf
    q = rng.standard_normal(n)
    q[[1, 4, 5, 6, 9, 10]] = 0.0   # variables without curvature need a bounded direction: keep them cost-free
    Ps, As = sp.csc_matrix(P), sp.csc_matrix(A)
    opts = dict(verbose=False, eps_abs=1e-7, eps_rel=1e-7, max_iter=20000, adaptive_rho_interval=25)
    m = _setup(oq, lib, linsys, dict(P=Ps, q=q, A=As, l=l, u=u), opts)
    r = oq.solve(m)
    assert r.info.status == "Solved"
    Ax = A @ r.x
    assert np.all(Ax >= l - 1e-5) and np.all(Ax <= u + 1e-5)
    assert np.max(np.abs(P @ r.x + q + A.T @ r.y)) <= 1e-5
    # complementarity: y_i > 0 only at the upper bound, < 0 only at the lower bound
    for i in range(mm):
        if r.y[i] > 1e-5:
            assert abs(Ax[i] - u[i]) <= 1e-4
        if r.y[i] < -1e-5:
            assert abs(Ax[i] - l[i]) <= 1e-4


ALL_CASES = [v for k, v in sorted(globals().items()) if k.startswith("case_")]
