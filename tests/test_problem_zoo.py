"""Standard QP classes (tests/qp_zoo.py): the CPU oracle against an independent evaluation of the stopping criteria
(CPU), and the HIP engine against the oracle and the same evaluation (GPU)."""
import ctypes as C

import numpy as np
import scipy.sparse as sp
import pytest

import osqp_jl_amd as oq
import qp_zoo

EPS = 1e-5
OPTS = dict(verbose=False, eps_abs=EPS, eps_rel=EPS, max_iter=20000, adaptive_rho_interval=25, polish=False)


def _solve(lib, prob, linsys):
    m = oq.Model(lib)
    oq.setup(m, linsys_solver=linsys, **prob, **OPTS)
    r = oq.solve(m)
    oq.clean(m)
    return r


def _check(prob, r):
    assert r.info.status == "Solved"
    pri, eps_pri, dua, eps_dua = qp_zoo.kkt_check(prob, r.x, r.y, EPS)
    assert pri <= 1.5 * eps_pri and dua <= 1.5 * eps_dua, (pri, eps_pri, dua, eps_dua)


@pytest.mark.parametrize("name", sorted(qp_zoo.ZOO))
def test_oracle_on_zoo(oracle_lib, name):
    prob = qp_zoo.ZOO[name]()
    r = _solve(oracle_lib, prob, "qdldl")
    _check(prob, r)


@pytest.mark.gpu
@pytest.mark.parametrize("linsys", ["qdldl", "pcg"])
@pytest.mark.parametrize("name", sorted(qp_zoo.ZOO))
def test_engine_on_zoo(product_lib, oracle_lib, name, linsys):
    prob = qp_zoo.ZOO[name]()
    ro = _solve(oracle_lib, prob, "qdldl")
    rp = _solve(product_lib, prob, linsys)
    _check(prob, rp)
    assert ro.info.status == "Solved"
    assert abs(ro.info.obj_val - rp.info.obj_val) <= 2e-4 * max(1.0, abs(ro.info.obj_val))
    if linsys == "qdldl":  # exact KKT solves on both sides: same trajectory
        assert ro.info.iter == rp.info.iter
        assert np.max(np.abs(ro.x - rp.x)) <= 1e-7 * max(1.0, np.max(np.abs(ro.x)))


@pytest.mark.gpu
def test_dense_top_block_at_high_accuracy(product_lib, oracle_lib):
    """The dense top block of the direct back-end (Schur complement inverted explicitly, csrc/direct.hip) must not
    cost accuracy: eps = 1e-10 with polish, same iteration count and solution as the oracle's LDL'."""
    prob = qp_zoo.portfolio(n=2000, k=80)
    out = []
    for lib, ls in ((oracle_lib, "qdldl"), (product_lib, "direct")):
        m = oq.Model(lib)
        oq.setup(m, linsys_solver=ls, verbose=False, eps_abs=1e-10, eps_rel=1e-10, max_iter=20000, adaptive_rho_interval=25,
                 polish=True, **prob)
        out.append(oq.solve(m))
        if lib is product_lib:
            assert oq.stats(m)[5] > 80  # one level per pivot of the block: the dense path is what ran
        oq.clean(m)
    ro, rp = out
    assert ro.info.status == rp.info.status == "Solved"
    assert ro.info.iter == rp.info.iter
    assert ro.info.status_polish == rp.info.status_polish == 1
    assert np.max(np.abs(ro.x - rp.x)) <= 1e-10 and abs(ro.info.obj_val - rp.info.obj_val) <= 1e-10


@pytest.mark.gpu
def test_nested_dissection_ordering_on_a_long_horizon(product_lib, oracle_lib):
    """A banded multi-stage problem: the engine's second ordering (nested dissection, csrc/symbolic.hip) replaces the
    min-degree chain of thousands of levels; the trajectory must still be the oracle's (exact solves on both sides)."""
    prob = qp_zoo.control(nx=8, nu=4, T=400)
    out = []
    for lib, ls in ((oracle_lib, "qdldl"), (product_lib, "direct")):
        m = oq.Model(lib)
        oq.setup(m, linsys_solver=ls, verbose=False, eps_abs=1e-6, eps_rel=1e-6, max_iter=4000, adaptive_rho_interval=25, **prob)
        out.append(oq.solve(m))
        if lib is product_lib:
            levels = oq.stats(m)[5]
            assert 50 < levels < 1000  # min-degree leaves > 4000 levels here
            assert 0 < oq.stats(m)[19] <= 20  # and the solves run by supernodes: a launch per level of the separator tree
        oq.clean(m)
    ro, rp = out
    assert ro.info.status == rp.info.status == "Solved"
    assert ro.info.iter == rp.info.iter
    assert np.max(np.abs(ro.x - rp.x)) <= 1e-7 * max(1.0, np.max(np.abs(ro.x)))
    assert np.max(np.abs(ro.y - rp.y)) <= 1e-7 * max(1.0, np.max(np.abs(ro.y)))


@pytest.mark.gpu
@pytest.mark.parametrize("smax", ["64", "3", "1"])
def test_single_pivot_supernodes_of_a_wide_level(product_lib, oracle_lib, monkeypatch, smax):
    """Round 5 (control-1e6: 388 258 of the 496 738 leaves are supernodes of ONE pivot): a wide level numbers them first;
    the forward sweep skips them at level 0 (nothing outside the block, a 1 x 1 unit block) and the backward sweep gives
    them a lane each (csrc/direct.hip k_sn_single_bwd) instead of a quarter wavefront.  Forced onto a small problem with
    three partitions (largest supernode 64 / 3 / 1 -- the last: every supernode of every level takes that path backward):
    the same KKT solves as with the path switched off, and the oracle's trajectory."""
    prob = qp_zoo.control(nx=8, nu=4, T=400)
    opts = dict(verbose=False, eps_abs=1e-6, eps_rel=1e-6, max_iter=4000, adaptive_rho_interval=25)
    mo = oq.Model(oracle_lib)
    oq.setup(mo, linsys_solver="qdldl", **opts, **prob)
    ro = oq.solve(mo)
    monkeypatch.setenv("OSQP_AMD_SNODE", "2")
    monkeypatch.setenv("OSQP_AMD_SNODE_MAX", smax)
    monkeypatch.setenv("OSQP_AMD_SNODE_WAVE_MIN", "1")
    n, mm = prob["P"].shape[0], prob["A"].shape[0]
    rhs = np.random.default_rng(3).standard_normal(n + mm)
    sols = []
    for single in ("0", "1"):
        monkeypatch.setenv("OSQP_AMD_SNODE_SINGLE", single)
        m = oq.Model(product_lib)
        oq.setup(m, linsys_solver="direct", **opts, **prob)
        assert oq.stats(m)[19] > 2
        out = np.empty_like(rhs)
        assert m.lib.osqp_amd_apply(m.workspace, 3, rhs.ctypes.data_as(C.POINTER(C.c_double)), out.ctypes.data_as(C.POINTER(C.c_double))) == 0
        rp = oq.solve(m)
        assert rp.info.status == ro.info.status == "Solved" and rp.info.iter == ro.info.iter
        assert np.max(np.abs(ro.x - rp.x)) <= 1e-7 * max(1.0, np.max(np.abs(ro.x)))
        sols.append(out)
        oq.clean(m)
    assert np.all(np.isfinite(sols[1]))
    assert np.max(np.abs(sols[0] - sols[1])) <= 1e-10 * max(1.0, np.max(np.abs(sols[0])))
    oq.clean(mo)


@pytest.mark.gpu
@pytest.mark.parametrize("tree", ["1", "0"])
def test_wavefront_forms_of_the_supernode_levels(product_lib, oracle_lib, monkeypatch, tree):
    """Round 4 (control-1e6): levels of many supernodes give each a wavefront, the small ones (<= 16 pivots, numbered first
    inside their level) a quarter wavefront (csrc/direct.hip k_sn_level_w); blocks are packed triangles.  Forced onto a small
    problem (OSQP_AMD_SNODE_WAVE_MIN=1: every level takes that form), with and without the one-launch top of the tree: the
    oracle's trajectory with either form (wide levels take a notch fewer lanes per row, so the order of a row's sum may
    differ from the workgroup-per-supernode form's)."""
    prob = qp_zoo.control(nx=8, nu=4, T=400)
    opts = dict(verbose=False, eps_abs=1e-6, eps_rel=1e-6, max_iter=4000, adaptive_rho_interval=25)
    mo = oq.Model(oracle_lib)
    oq.setup(mo, linsys_solver="qdldl", **opts, **prob)
    ro = oq.solve(mo)
    monkeypatch.setenv("OSQP_AMD_SNODE_TREE", tree)
    res = []
    for wave_min in ("1000000000", "1"):
        monkeypatch.setenv("OSQP_AMD_SNODE_WAVE_MIN", wave_min)
        m = oq.Model(product_lib)
        oq.setup(m, linsys_solver="direct", **opts, **prob)
        assert oq.stats(m)[19] > 2
        res.append(oq.solve(m))
        oq.clean(m)
    for rp in res:
        assert rp.info.status == ro.info.status == "Solved" and rp.info.iter == ro.info.iter
        assert np.max(np.abs(ro.x - rp.x)) <= 1e-7 * max(1.0, np.max(np.abs(ro.x)))
    oq.clean(mo)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["control", "portfolio"])
def test_update_that_leaves_the_next_right_hand_side_is_bit_identical(product_lib, monkeypatch, name):
    """Round 5: inside a chunk graph the update of iteration k writes the right-hand side of iteration k + 1 into the factor's
    vector (csrc/direct.hip k_direct_update_rhs) and that iteration skips its own right-hand-side launch.  Same arithmetic on
    the same values: the solve with the fusion off (OSQP_AMD_DIRECT_LEAVE_RHS=0) bit for bit, through a rho update and a second
    solve on the workspace (the chunk graph is captured again after the refactorisation)."""
    prob = qp_zoo.control(nx=8, nu=4, T=400) if name == "control" else qp_zoo.portfolio(n=300, k=10)
    opts = dict(verbose=False, eps_abs=1e-6, eps_rel=1e-6, max_iter=4000, adaptive_rho_interval=25)
    res = {}
    for leave in ("0", "1"):
        monkeypatch.setenv("OSQP_AMD_DIRECT_LEAVE_RHS", leave)
        m = oq.Model(product_lib)
        oq.setup(m, linsys_solver="direct", **opts, **prob)
        r1 = oq.solve(m)
        oq.update_q(m, 1.1 * prob["q"])
        r2 = oq.solve(m)
        res[leave] = (r1.info.iter, r1.x.copy(), r1.y.copy(), r2.info.iter, r2.x.copy(), r2.y.copy())
        assert r1.info.status == r2.info.status == "Solved"
        oq.clean(m)
    for a, b in zip(res["0"], res["1"]):
        assert np.array_equal(a, b)


@pytest.mark.gpu
def test_persistent_form_of_the_tree_kernels(product_lib, oracle_lib, monkeypatch):
    """Round 5: k_sn_tree with as many workgroups as the device holds taking the supernodes of the launch in level order from
    a ticket counter (opt-in since the level kernels take their entries flat: OSQP_AMD_SNODE_TREE_PERSIST=k, the levels from k
    on).  A horizon long enough that the levels above level 0 do not all fit the device (the plain form starts higher up):
    the trajectory of the plain form -- same iteration count, same solution -- twice on one workspace (the counters are back
    at rest after a solve), and the oracle's solution (its count may differ by one check interval at this tolerance: the
    supernodal and the oracle's factor round differently)."""
    prob = qp_zoo.control(nx=8, nu=4, T=6000)
    opts = dict(verbose=False, eps_abs=1e-5, eps_rel=1e-5, max_iter=4000, adaptive_rho_interval=25)
    mo = oq.Model(oracle_lib)
    oq.setup(mo, linsys_solver="qdldl", **opts, **prob)
    ro = oq.solve(mo)
    monkeypatch.setenv("OSQP_AMD_SNODE", "2")
    monkeypatch.setenv("OSQP_AMD_SNODE_MAX", "8")  # many supernodes: thousands above level 0
    res = {}
    for persist in ("0", "1", "2"):
        monkeypatch.setenv("OSQP_AMD_SNODE_TREE_PERSIST", persist)
        m = oq.Model(product_lib)
        oq.setup(m, linsys_solver="direct", **opts, **prob)
        assert oq.stats(m)[19] > 2
        for _ in range(2):
            rp = oq.solve(m)
            assert rp.info.status == ro.info.status == "Solved" and abs(rp.info.iter - ro.info.iter) <= 25
            assert np.max(np.abs(ro.x - rp.x)) <= 1e-4 * max(1.0, np.max(np.abs(ro.x)))
            res.setdefault(persist, []).append((rp.info.iter, rp.x.copy()))
            oq.warm_start(m, x=np.zeros_like(ro.x), y=np.zeros_like(ro.y))
        oq.clean(m)
    for persist in ("1", "2"):
        for k in range(2):
            assert res[persist][k][0] == res["0"][k][0]
            assert np.max(np.abs(res[persist][k][1] - res["0"][k][1])) <= 1e-9 * max(1.0, np.max(np.abs(res["0"][k][1])))
    oq.clean(mo)


@pytest.mark.gpu
def test_a_timed_out_wait_in_the_tree_kernels_restarts_the_solve_on_the_level_path(product_lib, oracle_lib, monkeypatch):
    """k_sn_tree waits inside a kernel (csrc/direct.hip); a wait that times out raises a flag in mapped host memory.
    The host side of that: the factor goes back to one launch per level and the solve in progress starts again from a
    cold start (the iterations run since the last test cannot be trusted) -- the caller gets the oracle's answer, not an
    error and not a damaged iterate; the next solve on the same workspace is the oracle's again.  (The flag is injected at
    the first health check: OSQP_AMD_SNODE_FAULT_TEST.)"""
    prob = qp_zoo.control(nx=8, nu=4, T=400)
    opts = dict(verbose=False, eps_abs=1e-6, eps_rel=1e-6, max_iter=4000, adaptive_rho_interval=25)
    mo = oq.Model(oracle_lib)
    oq.setup(mo, linsys_solver="qdldl", **opts, **prob)
    ro = oq.solve(mo)
    monkeypatch.setenv("OSQP_AMD_SNODE_FAULT_TEST", "1")
    m = oq.Model(product_lib)
    oq.setup(m, linsys_solver="direct", **opts, **prob)
    monkeypatch.delenv("OSQP_AMD_SNODE_FAULT_TEST")
    assert oq.stats(m)[19] > 2
    for k in range(2):  # the solve that meets the fault and restarts, then a fresh one on the same workspace
        rp = oq.solve(m)
        assert rp.info.status == ro.info.status == "Solved"
        assert np.max(np.abs(ro.x - rp.x)) <= 1e-4 * max(1.0, np.max(np.abs(ro.x)))
        assert abs(ro.info.obj_val - rp.info.obj_val) <= 1e-5 * max(1.0, abs(ro.info.obj_val))
        if k == 0:
            assert rp.info.iter == ro.info.iter  # the restart is the oracle's cold-start solve: rho and its interval are back at their entry values
        assert oq.stats(m)[19] > 2   # still a supernodal factor, now one launch per level
        assert oq.stats(m)[21] == 1  # one restart, none after it
    assert oq.stats(m)[21] == 1
    oq.clean(m); oq.clean(mo)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["portfolio", "svm", "huber", "lasso_data", "equality_qp", "control"])
def test_polish_on_zoo(product_lib, oracle_lib, name):
    """Polish on the standard classes: same outcome and objective as the oracle's polish."""
    prob = qp_zoo.ZOO[name]()
    out = []
    for lib in (oracle_lib, product_lib):
        m = oq.Model(lib)
        oq.setup(m, linsys_solver="qdldl", **prob, **dict(OPTS, polish=True))
        out.append(oq.solve(m))
        oq.clean(m)
    ro, rp = out
    assert ro.info.status == rp.info.status == "Solved"
    assert ro.info.status_polish == rp.info.status_polish
    assert abs(ro.info.obj_val - rp.info.obj_val) <= 1e-6 * max(1.0, abs(ro.info.obj_val))
    if ro.info.status_polish == 1:
        pri, eps_pri, dua, eps_dua = qp_zoo.kkt_check(prob, rp.x, rp.y, 1e-7)
        assert pri <= 1e-6 and dua <= 1e-5, (pri, dua)


@pytest.mark.gpu
@pytest.mark.parametrize("n", [600, 1100])
def test_inertia_inside_a_large_dense_block(product_lib, n):
    """[REF test/non_convex.jl:6-21] at the size where the dense top block is inverted by block sweeps on the matrix cores
    (csrc/direct.hip k_gj_*, from 512 pivots): an indefinite dense P -- one negative direction buried in the middle of the
    block -- must fail `osqp_setup` (a pivot of the block's own sweeps has the wrong sign), the same P shifted to be positive
    definite must set up, report the block and solve to the closed-form answer of its equality-constrained problem."""
    rng = np.random.default_rng(n)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    ev = 0.5 + rng.random(n)
    ev[n // 2] = -0.3
    Pind = (Q * ev) @ Q.T
    Pind = 0.5 * (Pind + Pind.T)
    m = 20
    A = sp.csc_matrix(rng.standard_normal((m, n)))
    b = rng.standard_normal(m)
    q = rng.standard_normal(n)
    opts = dict(verbose=False, eps_abs=1e-7, eps_rel=1e-7, max_iter=4000, adaptive_rho_interval=25, sigma=1e-6, linsys_solver="direct")
    mdl = oq.Model(product_lib)
    with pytest.raises(oq.OSQPError):
        oq.setup(mdl, P=sp.csc_matrix(Pind), q=q, A=A, l=b, u=b, **opts)
    ev[n // 2] = 0.3
    Ppd = (Q * ev) @ Q.T
    Ppd = 0.5 * (Ppd + Ppd.T)
    mdl = oq.Model(product_lib)
    oq.setup(mdl, P=sp.csc_matrix(Ppd), q=q, A=A, l=b, u=b, **opts)
    assert oq.stats(mdl)[5] > 500  # one level per pivot of the block: the dense path is what ran
    r = oq.solve(mdl)
    assert r.info.status == "Solved"
    K = np.block([[Ppd, A.toarray().T], [A.toarray(), np.zeros((m, m))]])
    sol = np.linalg.solve(K, np.concatenate([-q, b]))
    assert np.max(np.abs(r.x - sol[:n])) <= 1e-5 * max(1.0, np.max(np.abs(sol[:n])))
    oq.clean(mdl)
