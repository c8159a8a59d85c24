"""Seeded random small QPs (ragged structure, empty rows and columns, infinite / equal / one-sided bounds, zero P,
infeasible and unbounded instances): with the direct back-end on both sides the HIP engine must follow the CPU
oracle's trajectory -- same status, same iteration count, same iterates to rounding."""
import numpy as np
import pytest
import scipy.sparse as sp

import osqp_jl_amd as oq

pytestmark = pytest.mark.gpu


def compare(ro, rp, eps):
    """How a product result relates to the oracle's on the same problem (direct back-end on both sides):
    "exact"      same status and iteration count, iterates equal to 1e-7 (relative to the largest entry): the trajectory is the same;
    "close"      same status, termination at most one check apart, iterates equal to the accuracy asked for (100 eps):
                 rounding (the two libraries eliminate in different orders) amplified by an ill-conditioned instance;
    "borderline" one side stopped on an infeasibility test or ran out of iterations where the other did not;
    "different"  anything else."""
    def rel(a, b):
        return float(np.max(np.abs(a - b))) / max(1.0, float(np.max(np.abs(a)))) if len(a) else 0.0
    so, sp = ro.info.status, rp.info.status
    if so != sp:
        soft = {"Max_iter_reached", "Primal_infeasible", "Dual_infeasible", "Solved_inaccurate", "Primal_infeasible_inaccurate",
                "Dual_infeasible_inaccurate"}
        return "borderline" if (so in soft or sp in soft) else "different"
    if so in ("Solved", "Max_iter_reached", "Solved_inaccurate"):
        d = max(rel(ro.x, rp.x), rel(ro.y, rp.y))
    elif so.startswith("Primal_infeasible"):
        d = rel(ro.prim_inf_cert, rp.prim_inf_cert)
    elif so.startswith("Dual_infeasible"):
        d = rel(ro.dual_inf_cert, rp.dual_inf_cert)
    else:
        d = 0.0
    if ro.info.iter == rp.info.iter and d <= 1e-7:
        return "exact"
    if abs(ro.info.iter - rp.info.iter) <= 25 and (d <= 100 * eps or so == "Max_iter_reached"):
        return "close"
    return "different"


def random_problem(rng):
    n = int(rng.integers(1, 25))
    m = int(rng.integers(0, 35))
    dens = rng.choice([0.05, 0.2, 0.6])
    if rng.random() < 0.2:
        P = sp.csc_matrix((n, n))
    else:
        M = sp.random(n, n, density=dens, random_state=rng, data_rvs=rng.standard_normal)
        P = (M @ M.T + (0.0 if rng.random() < 0.3 else 0.1) * sp.eye(n)).tocsc()
    q = rng.standard_normal(n) * rng.choice([0.0, 1.0, 10.0])
    A = sp.random(m, n, density=dens, random_state=rng, data_rvs=rng.standard_normal, format="csc")
    center = rng.standard_normal(m)
    width = rng.random(m) * rng.choice([0.0, 1.0, 5.0], size=m)
    l, u = center - width, center + width
    kind = rng.integers(0, 5, size=m)
    l = np.where(kind == 0, -np.inf, l)
    u = np.where(kind == 1, np.inf, u)
    both = kind == 2
    l = np.where(both, -np.inf, l); u = np.where(both, np.inf, u)
    return dict(P=sp.triu(P, format="csc"), q=q, A=A, l=l, u=u)


@pytest.mark.parametrize("block", range(int(__import__("os").environ.get("OSQP_FUZZ_BLOCKS", "4"))))  # 25 problems each
def test_random_small_problems_follow_the_oracle(product_lib, oracle_lib, block):
    rng = np.random.default_rng(1000 + block)
    tally = {"exact": 0, "close": 0, "borderline": 0, "different": 0}
    notes = []
    solved = 0
    for k in range(25):
        prob = random_problem(rng)
        opts = dict(verbose=False, eps_abs=1e-5, eps_rel=1e-5, max_iter=2000, adaptive_rho_interval=25,
                    scaling=int(rng.choice([0, 1, 10])), alpha=float(rng.choice([1.0, 1.6])))
        res = []
        for lib, ls in ((oracle_lib, "qdldl"), (product_lib, "direct")):
            m = oq.Model(lib)
            oq.setup(m, linsys_solver=ls, **prob, **opts)
            res.append(oq.solve(m))
            oq.clean(m)
        verdict = compare(res[0], res[1], 1e-5)
        tally[verdict] += 1
        solved += res[0].info.status == "Solved"
        if verdict != "exact":
            notes.append("block %d problem %d (n=%d, m=%d): %s, oracle %s/%d, product %s/%d" % (
                block, k, prob["P"].shape[0], prob["A"].shape[0], verdict, res[0].info.status, res[0].info.iter, res[1].info.status, res[1].info.iter))
    assert tally["different"] == 0 and tally["borderline"] <= 1 and tally["exact"] >= 22, (tally, notes)
    assert solved > 0


@pytest.mark.parametrize("lean", ["0", "1"])
@pytest.mark.parametrize("smax", [3, 64])
def test_supernodal_solves_follow_the_oracle(product_lib, oracle_lib, smax, lean, monkeypatch):
    """The same comparison with the triangular solves forced through supernodes (csrc/direct.hip k_sn_*; blocks of at
    most `smax` pivots inverted once per factorisation): subtree and path supernodes, many levels at smax = 3.  Round 5: the
    numeric factorisation behind them is the multifrontal one (csrc/mfront.hpp), and with lean = 1 the factor's index arrays
    are built on the device from a lean host analysis (polish included: its reduced KKT system takes the host path)."""
    monkeypatch.setenv("OSQP_AMD_SNODE", "2")
    monkeypatch.setenv("OSQP_AMD_SNODE_MAX", str(smax))
    monkeypatch.setenv("OSQP_AMD_LEAN", lean)
    rng = np.random.default_rng(3000 + smax)
    tally = {"exact": 0, "close": 0, "borderline": 0, "different": 0}
    notes = []
    deepest = 0
    for k in range(25):
        prob = random_problem(rng)
        opts = dict(verbose=False, eps_abs=1e-5, eps_rel=1e-5, max_iter=2000, adaptive_rho_interval=25,
                    scaling=int(rng.choice([0, 1, 10])), alpha=float(rng.choice([1.0, 1.6])), polish=bool(k % 2))
        res = []
        for lib, ls in ((oracle_lib, "qdldl"), (product_lib, "direct")):
            m = oq.Model(lib)
            oq.setup(m, linsys_solver=ls, **prob, **opts)
            res.append(oq.solve(m))
            if lib is product_lib:
                st = oq.stats(m)
                assert st[19] >= 1  # the supernodal path is what ran
                assert st[22] == 1.0 and st[23] == float(lean == "1")  # ... behind a multifrontal factorisation, device-built when lean
                deepest = max(deepest, int(st[19]))
            oq.clean(m)
        verdict = compare(res[0], res[1], 1e-5)
        tally[verdict] += 1
        if verdict != "exact":
            notes.append("problem %d (n=%d, m=%d): %s, oracle %s/%d, product %s/%d" % (
                k, prob["P"].shape[0], prob["A"].shape[0], verdict, res[0].info.status, res[0].info.iter, res[1].info.status, res[1].info.iter))
        assert res[0].info.status_polish == res[1].info.status_polish or verdict != "exact", notes
    assert tally["different"] == 0 and tally["borderline"] <= 1 and tally["exact"] >= 22, (tally, notes)
    assert deepest >= (4 if smax == 3 else 1)


def feasible_problem(rng):
    n = int(rng.integers(2, 30))
    m = int(rng.integers(1, 40))
    M = sp.random(n, n, density=0.3, random_state=rng, data_rvs=rng.standard_normal)
    P = sp.triu((M @ M.T + 0.1 * sp.eye(n)).tocsc(), format="csc")
    A = sp.random(m, n, density=0.3, random_state=rng, data_rvs=rng.standard_normal, format="csc")
    x0 = rng.standard_normal(n)
    w = rng.random(m) * rng.choice([0.0, 1.0], size=m)
    return dict(P=P, q=rng.standard_normal(n), A=A, l=A @ x0 - w, u=A @ x0 + w), x0


@pytest.mark.parametrize("block", range(max(2, int(__import__("os").environ.get("OSQP_FUZZ_BLOCKS", "4")) // 2)))
def test_random_update_sequences_follow_the_oracle(product_lib, oracle_lib, block):
    """setup -> solve -> update_q -> solve -> update_bounds -> solve -> update_P_A (all values, then an index subset)
    -> solve -> update_rho -> solve -> warm start -> solve, the same calls on both libraries."""
    rng = np.random.default_rng(2000 + block)
    tally = {"exact": 0, "close": 0, "borderline": 0, "different": 0}
    notes = []
    for k in range(12):
        prob, x0 = feasible_problem(rng)
        n, m = prob["P"].shape[0], prob["A"].shape[0]
        opts = dict(verbose=False, eps_abs=1e-6, eps_rel=1e-6, max_iter=4000, adaptive_rho_interval=25)
        models = []
        for lib, ls in ((oracle_lib, "qdldl"), (product_lib, "direct")):
            mdl = oq.Model(lib)
            oq.setup(mdl, linsys_solver=ls, **prob, **opts)
            models.append(mdl)
        q2 = rng.standard_normal(n)
        shift = 0.1 * rng.standard_normal(m)
        Px2 = prob["P"].data * (0.5 + rng.random())  # a positive multiple keeps P positive semidefinite
        Ax2 = prob["A"].data * (1.0 + 0.2 * rng.standard_normal(prob["A"].nnz))
        nsub = max(1, prob["A"].nnz // 3)
        sub = np.sort(rng.choice(prob["A"].nnz, size=min(nsub, prob["A"].nnz), replace=False)) if prob["A"].nnz else np.zeros(0, int)
        Asub = rng.standard_normal(len(sub))
        steps = [
            lambda mm: None,
            lambda mm: oq.update(mm, q=q2),
            lambda mm: oq.update(mm, l=prob["l"] + shift - 0.05, u=prob["u"] + shift + 0.05),
            lambda mm: oq.update(mm, Px=Px2, Ax=Ax2),
            (lambda mm: oq.update(mm, Ax=Asub, Ax_idx=sub)) if len(sub) else (lambda mm: None),
            lambda mm: oq.update_settings(mm, rho=0.7),
            lambda mm: oq.warm_start(mm, x=x0, y=np.zeros(m)),
        ]
        for si, step in enumerate(steps):
            out = []
            for mdl in models:
                step(mdl)
                out.append(oq.solve(mdl))
            verdict = compare(out[0], out[1], 1e-6)
            tally[verdict] += 1
            if verdict != "exact":
                notes.append("block %d problem %d step %d (n=%d, m=%d): %s, oracle %s/%d, product %s/%d" % (
                    block, k, si, n, m, verdict, out[0].info.status, out[0].info.iter, out[1].info.status, out[1].info.iter))
            if verdict != "exact":
                break  # the two models are no longer in the same state (iterates, rho): later steps of this problem say nothing
        for mdl in models:
            oq.clean(mdl)
    total = sum(tally.values())
    assert tally["different"] == 0 and tally["borderline"] <= 1 and tally["exact"] >= 0.9 * total, (tally, notes)


def test_update_to_indefinite_P_is_refused_by_both(product_lib, oracle_lib):
    """A value update that makes P indefinite breaks the inertia of the KKT factor: the direct back-end of both
    libraries must report it through the exit flag (the Julia side raises on a non-zero flag)."""
    P = sp.csc_matrix(np.diag([1.0, 1.0]))
    A = sp.csc_matrix(np.eye(2))
    for lib, ls in ((oracle_lib, "qdldl"), (product_lib, "direct")):
        m = oq.Model(lib)
        oq.setup(m, P=P, q=np.ones(2), A=A, l=-np.ones(2), u=np.ones(2), verbose=False, linsys_solver=ls)
        assert oq.solve(m).info.status == "Solved"
        with pytest.raises(oq.OSQPError):
            oq.update(m, Px=np.array([1.0, -5.0]))
        oq.clean(m)


def test_two_workspaces_interleaved(product_lib):
    """Distinct workspaces are independent [REF SURVEY 8b threading]: two models set up together and driven in
    alternation (solve, update, warm start) give what each gives alone."""
    rng = np.random.default_rng(77)
    probs = [feasible_problem(rng)[0] for _ in range(2)]
    opts = dict(verbose=False, eps_abs=1e-6, eps_rel=1e-6, adaptive_rho_interval=25)

    def alone(prob, q2, ls):
        m = oq.Model(product_lib); oq.setup(m, linsys_solver=ls, **prob, **opts)
        a = oq.solve(m); oq.update(m, q=q2); b = oq.solve(m); oq.clean(m)
        return a, b

    q2s = [rng.standard_normal(p["P"].shape[0]) for p in probs]
    for ls in ("qdldl", "pcg"):
        ref = [alone(p, q2, ls) for p, q2 in zip(probs, q2s)]
        ms = []
        for p in probs:
            m = oq.Model(product_lib); oq.setup(m, linsys_solver=ls, **p, **opts); ms.append(m)
        first = [oq.solve(ms[0]), oq.solve(ms[1])]
        oq.update(ms[1], q=q2s[1]); oq.update(ms[0], q=q2s[0])
        second = [None, None]
        second[1] = oq.solve(ms[1]); second[0] = oq.solve(ms[0])
        for i in range(2):
            assert first[i].info.iter == ref[i][0].info.iter and np.array_equal(first[i].x, ref[i][0].x)
            assert second[i].info.iter == ref[i][1].info.iter and np.array_equal(second[i].x, ref[i][1].x)
        for m in ms:
            oq.clean(m)
