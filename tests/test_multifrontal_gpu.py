"""Numeric LDL' by supernodes (csrc/mfront.hpp, row K2): the multifrontal factorisation against the level-by-level one on
the same workspace layout -- one KKT solve with each factor on a random right-hand side, before and after a rho update
(every `update_settings!(rho=...)`, `update_P!/A!` refactors [REF src/interface.jl:330-406, 539-550]) -- and against the
CPU oracle's trajectory."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

import osqp_jl_amd as oq
import qp_zoo


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _kkt_solve(m, rhs):
    out = np.empty_like(rhs)
    assert m.lib.osqp_amd_apply(m.workspace, 3, _fptr(rhs), _fptr(out)) == 0
    return out


def _random_problem(rng, n, m, dens):
    M = sp.random(n, n, density=dens, random_state=rng, data_rvs=rng.standard_normal)
    P = sp.triu((M @ M.T + 0.1 * sp.eye(n)).tocsc(), format="csc")
    A = sp.random(m, n, density=dens, random_state=rng, data_rvs=rng.standard_normal, format="csc")
    l = -rng.random(m) - 0.1
    u = rng.random(m) + 0.1
    eq = rng.random(m) < 0.2
    u[eq] = l[eq]
    return dict(P=P, q=rng.standard_normal(n), A=A, l=l, u=u)


CASES = {
    "control-400": (lambda: qp_zoo.control(nx=8, nu=4, T=400), 64),
    "control-300-small-supernodes": (lambda: qp_zoo.control(nx=12, nu=6, T=300), 5),
    "control-200-singletons": (lambda: qp_zoo.control(nx=6, nu=3, T=200), 1),
    "portfolio": (lambda: qp_zoo.portfolio(n=300, k=10), 16),
    "svm": (lambda: qp_zoo.svm(n=20, m=150), 64),
    "random-60": (lambda: _random_problem(np.random.default_rng(5), 60, 90, 0.08), 64),
    "random-200": (lambda: _random_problem(np.random.default_rng(6), 200, 150, 0.02), 24),
    "grid2d-40": (lambda: qp_zoo.grid2d(40), 64),
    "grid3d-9": (lambda: qp_zoo.grid3d(9), 32),
}


@pytest.mark.gpu
@pytest.mark.parametrize("share_min", ["", "3"])
@pytest.mark.parametrize("case", sorted(CASES))
def test_multifrontal_factor_is_the_level_factor(product_lib, monkeypatch, case, share_min):
    """`share_min`: update matrices of at least that many entries share memory over their lifetimes (written at their supernode's
    level, read at the parent's; csrc/direct.hip place_update_matrices -- by default from 16 384 entries on, which no problem of
    test size reaches; 3 = every update matrix beyond one row)."""
    make, smax = CASES[case]
    prob = make()
    if share_min:
        monkeypatch.setenv("OSQP_AMD_MF_SHARE_MIN", share_min)
    monkeypatch.setenv("OSQP_AMD_SNODE", "2")
    monkeypatch.setenv("OSQP_AMD_SNODE_MAX", str(smax))
    n, mm = prob["P"].shape[0], prob["A"].shape[0]
    rhs = np.random.default_rng(11).standard_normal(n + mm)
    sols = {}
    for mf in ("0", "1"):
        monkeypatch.setenv("OSQP_AMD_MF", mf)
        m = oq.Model(product_lib)
        oq.setup(m, linsys_solver="direct", verbose=False, adaptive_rho=False, **prob)
        st = oq.stats(m)
        assert st[19] >= 1 and st[22] == float(mf == "1"), (st[19], st[22])
        first = _kkt_solve(m, rhs)
        oq.update_settings(m, rho=0.731)
        second = _kkt_solve(m, rhs)
        assert oq.stats(m)[8] == 2
        sols[mf] = (first, second)
        oq.clean(m)
    for k in range(2):
        a, b = sols["0"][k], sols["1"][k]
        assert np.all(np.isfinite(b))
        assert np.max(np.abs(a - b)) <= 1e-9 * max(1.0, np.max(np.abs(a))), (case, k, np.max(np.abs(a - b)), np.max(np.abs(a)))
    assert np.max(np.abs(sols["1"][0] - sols["1"][1])) > 1e-6  # the rho update did change the factor


@pytest.mark.gpu
@pytest.mark.parametrize("split", ["1", "0"])
@pytest.mark.parametrize("max_front", [8, 40])
@pytest.mark.parametrize("case", sorted(CASES))
def test_fronts_beyond_lds_give_the_level_factor(product_lib, monkeypatch, case, max_front, split):
    """Round 6: fronts of more rows than one workgroup's LDS holds are factorised out of global memory (csrc/mfront_big.hpp:
    pivot block in LDS, panel and update matrix in global tiles) instead of sending the whole matrix back to the level-by-level
    factorisation.  The zoo at test sizes has no front beyond 192 rows, so the threshold is lowered (OSQP_AMD_MF_MAX_FRONT): every
    front above `max_front` rows -- parents and children of LDS fronts among them -- takes the global-memory path, and the KKT
    solves must be those of the level-by-level factor, before and after a rho update.  `split`: the pivot blocks in one launch and
    the panels in 64-row pieces in a second (k_mfb_pivot + k_mfb_rows, the default) or one workgroup per front (k_mfb_panel)."""
    make, smax = CASES[case]
    prob = make()
    monkeypatch.setenv("OSQP_AMD_MFB_SPLIT", split)
    monkeypatch.setenv("OSQP_AMD_MF_SHARE_MIN", "3" if max_front == 8 else "100000000")  # (shared update matrices under the big fronts too)
    monkeypatch.setenv("OSQP_AMD_SNODE", "2")
    monkeypatch.setenv("OSQP_AMD_SNODE_MAX", str(smax))
    n, mm = prob["P"].shape[0], prob["A"].shape[0]
    rhs = np.random.default_rng(12).standard_normal(n + mm)
    sols = {}
    for mf in ("0", "1"):
        monkeypatch.setenv("OSQP_AMD_MF", mf)
        if mf == "1":
            monkeypatch.setenv("OSQP_AMD_MF_MAX_FRONT", str(max_front))
        m = oq.Model(product_lib)
        oq.setup(m, linsys_solver="direct", verbose=False, adaptive_rho=False, **prob)
        st = oq.stats(m)
        assert st[19] >= 1 and st[22] == float(mf == "1"), (st[19], st[22])
        first = _kkt_solve(m, rhs)
        oq.update_settings(m, rho=0.731)
        second = _kkt_solve(m, rhs)
        sols[mf] = (first, second)
        oq.clean(m)
    for k in range(2):
        a, b = sols["0"][k], sols["1"][k]
        assert np.all(np.isfinite(b))
        assert np.max(np.abs(a - b)) <= 1e-9 * max(1.0, np.max(np.abs(a))), (case, k, np.max(np.abs(a - b)), np.max(np.abs(a)))


@pytest.mark.gpu
@pytest.mark.parametrize("mf", ["0", "1"])
def test_multifrontal_factor_reports_wrong_inertia(product_lib, monkeypatch, mf):
    """A non-convex P must fail osqp_setup through the inertia count of the fronts' pivots [REF test/non_convex.jl:11-21]
    (an entry far below -rho |A_j|^2, so that the reduced matrix P + sigma I + A' rho A is indefinite too)."""
    monkeypatch.setenv("OSQP_AMD_SNODE", "2")
    monkeypatch.setenv("OSQP_AMD_MF", mf)
    prob = qp_zoo.control(nx=6, nu=3, T=100)
    m = oq.Model(product_lib)
    oq.setup(m, linsys_solver="direct", verbose=False, scaling=0, **prob)
    assert oq.stats(m)[22] == float(mf == "1")
    oq.clean(m)
    P = prob["P"].tolil()
    P[3, 3] = -1e6
    prob["P"] = sp.triu(P.tocsc(), format="csc")
    m = oq.Model(product_lib)
    with pytest.raises(oq.OSQPError):
        oq.setup(m, linsys_solver="direct", verbose=False, scaling=0, **prob)


@pytest.mark.gpu
@pytest.mark.parametrize("smax", [4, 64])
def test_multifrontal_solve_follows_the_oracle(product_lib, oracle_lib, monkeypatch, smax):
    """Whole solves with adaptive rho (several refactorisations): the oracle's iteration count and solution."""
    monkeypatch.setenv("OSQP_AMD_SNODE", "2")
    monkeypatch.setenv("OSQP_AMD_SNODE_MAX", str(smax))
    prob = qp_zoo.control(nx=8, nu=4, T=400)
    opts = dict(verbose=False, eps_abs=1e-6, eps_rel=1e-6, max_iter=4000, adaptive_rho_interval=25)
    mo = oq.Model(oracle_lib)
    oq.setup(mo, linsys_solver="qdldl", **opts, **prob)
    ro = oq.solve(mo)
    m = oq.Model(product_lib)
    oq.setup(m, linsys_solver="direct", **opts, **prob)
    assert oq.stats(m)[22] == 1.0
    rp = oq.solve(m)
    assert rp.info.status == ro.info.status == "Solved" and rp.info.iter == ro.info.iter
    assert rp.info.rho_updates == ro.info.rho_updates and oq.stats(m)[8] == 1 + rp.info.rho_updates
    assert np.max(np.abs(ro.x - rp.x)) <= 1e-7 * max(1.0, np.max(np.abs(ro.x)))
    oq.clean(m); oq.clean(mo)


LEAN_CASES = {
    "control-400": (lambda: qp_zoo.control(nx=8, nu=4, T=400), 64, "1"),
    "control-300-min-degree": (lambda: qp_zoo.control(nx=12, nu=6, T=300), 7, "0"),
    "portfolio": (lambda: qp_zoo.portfolio(n=300, k=10), 16, "0"),
    "random-200": (lambda: _random_problem(np.random.default_rng(6), 200, 150, 0.02), 24, "0"),
}


@pytest.mark.gpu
@pytest.mark.parametrize("threads", ["1", "5"])
@pytest.mark.parametrize("case", sorted(LEAN_CASES))
def test_lean_setup_builds_the_same_factor_on_the_device(product_lib, monkeypatch, case, threads):
    """Round 5 (setup time of the direct back-end): a lean host analysis hands over the unsorted rows of the pattern of L
    and the device builds the CSC arrays, the scatter maps and the supernode lists from them (csrc/direct.hip
    lean_device_*).  Same arrays as the host-built ones, so the KKT solves agree BIT FOR BIT, before and after a matrix
    update (the scatter maps) and a rho update."""
    make, smax, ordering = LEAN_CASES[case]
    prob = make()
    monkeypatch.setenv("OSQP_AMD_SNODE", "2")
    monkeypatch.setenv("OSQP_AMD_SNODE_MAX", str(smax))
    monkeypatch.setenv("OSQP_AMD_FIRST_ORDERING", ordering)
    monkeypatch.setenv("OSQP_AMD_ND", "0")       # like with like: no second opinion of another ordering on the full analysis
    monkeypatch.setenv("OSQP_AMD_MD_FIFO", "0")  # (a lean analysis keeps the ordering it was asked for)
    monkeypatch.setenv("OSQP_AMD_HOST_THREADS", threads)
    n, mm = prob["P"].shape[0], prob["A"].shape[0]
    rhs = np.random.default_rng(12).standard_normal(n + mm)
    A = sp.csc_matrix(prob["A"])
    sols = {}
    for lean in ("0", "1"):
        monkeypatch.setenv("OSQP_AMD_LEAN", lean)
        m = oq.Model(product_lib)
        oq.setup(m, linsys_solver="direct", verbose=False, adaptive_rho=False, **prob)
        st = oq.stats(m)
        assert st[19] >= 1 and st[22] == 1.0 and st[23] == float(lean == "1"), (st[19], st[22], st[23])
        first = _kkt_solve(m, rhs)
        oq.update_A(m, A.data * 1.25, None)
        second = _kkt_solve(m, rhs)
        oq.update_settings(m, rho=0.37)
        third = _kkt_solve(m, rhs)
        r = oq.solve(m)
        sols[lean] = (first, second, third, r.x.copy(), r.info.iter, st[4], st[5], st[19], st[11])
        oq.clean(m)
    for k in range(4):
        assert np.array_equal(sols["0"][k], sols["1"][k]), (case, k, np.max(np.abs(sols["0"][k] - sols["1"][k])))
    assert sols["0"][4:] == sols["1"][4:]  # iterations, nnz(L), levels, supernode levels, bytes of a solve
    assert np.max(np.abs(sols["1"][0] - sols["1"][1])) > 1e-9


FORM_CASES = {
    "control-400": (lambda: qp_zoo.control(nx=8, nu=4, T=400), 64),
    "control-300-odd-blocks": (lambda: qp_zoo.control(nx=12, nu=6, T=300), 23),
    "portfolio": (lambda: qp_zoo.portfolio(n=300, k=10), 40),
    "random-200": (lambda: _random_problem(np.random.default_rng(6), 200, 150, 0.02), 33),
}


@pytest.mark.gpu
@pytest.mark.parametrize("wave_min", ["1", "1000000000"])
@pytest.mark.parametrize("case", sorted(FORM_CASES))
def test_forms_of_the_supernode_level_kernels_agree(product_lib, monkeypatch, case, wave_min):
    """Round 5, second pass over the level kernels (csrc/direct.hip): entries outside the blocks taken flat through LDS
    (k_sn_level_f / _wf), inverted blocks of the wavefront-form supernodes folded in place after every factorisation
    (k_sn_fold / sn_block_fold), single pivots a lane each.  Each switched off in turn against all on -- the same KKT solves
    (the order of a row's sum differs between forms, not its terms) before and after a refactorisation (the blocks are folded
    again), with every level in the wavefront form (OSQP_AMD_SNODE_WAVE_MIN=1) and with none; odd and even block sizes."""
    make, smax = FORM_CASES[case]
    prob = make()
    monkeypatch.setenv("OSQP_AMD_SNODE", "2")
    monkeypatch.setenv("OSQP_AMD_SNODE_MAX", str(smax))
    monkeypatch.setenv("OSQP_AMD_SNODE_WAVE_MIN", wave_min)
    n, mm = prob["P"].shape[0], prob["A"].shape[0]
    rhs = np.random.default_rng(13).standard_normal(n + mm)
    sols = {}
    for off in ("", "OSQP_AMD_SNODE_FLAT", "OSQP_AMD_SNODE_FOLD", "OSQP_AMD_SNODE_SINGLE"):
        for name in ("OSQP_AMD_SNODE_FLAT", "OSQP_AMD_SNODE_FOLD", "OSQP_AMD_SNODE_SINGLE"):
            monkeypatch.setenv(name, "0" if name == off else "1")
        m = oq.Model(product_lib)
        oq.setup(m, linsys_solver="direct", verbose=False, adaptive_rho=False, **prob)
        assert oq.stats(m)[19] >= 1
        first = _kkt_solve(m, rhs)
        oq.update_settings(m, rho=0.413)
        second = _kkt_solve(m, rhs)
        sols[off] = (first, second)
        oq.clean(m)
    for off, (first, second) in sols.items():
        for k, v in enumerate((first, second)):
            ref = sols[""][k]
            assert np.all(np.isfinite(v))
            assert np.max(np.abs(v - ref)) <= 1e-10 * max(1.0, np.max(np.abs(ref))), (case, off, k, np.max(np.abs(v - ref)))
    assert np.max(np.abs(sols[""][0] - sols[""][1])) > 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["control-400", "portfolio"])
def test_lone_leaves_outside_the_blocks(product_lib, monkeypatch, case):
    """Round 5: tree leaves whose column holds one entry (box-constraint rows) stay out of their parent's subtree supernode --
    supernodes of one pivot instead of a row and a column of an inverted block (csrc/symbolic.hip build_supernodes; by
    default only from 400 000 of them on, forced here).  Same KKT solves with the rule off (0), on (1) and on with the
    leaves not counted towards the subtree size (2), through the multifrontal factorisation and a refactorisation."""
    make, smax = FORM_CASES[case]
    prob = make()
    monkeypatch.setenv("OSQP_AMD_SNODE", "2")
    monkeypatch.setenv("OSQP_AMD_SNODE_MAX", str(smax))
    monkeypatch.setenv("OSQP_AMD_SNODE_WAVE_MIN", "1")
    n, mm = prob["P"].shape[0], prob["A"].shape[0]
    rhs = np.random.default_rng(14).standard_normal(n + mm)
    sols, levels = {}, {}
    for leaf in ("0", "1", "2"):
        monkeypatch.setenv("OSQP_AMD_SNODE_LEAF", leaf)
        m = oq.Model(product_lib)
        oq.setup(m, linsys_solver="direct", verbose=False, adaptive_rho=False, **prob)
        st = oq.stats(m)
        assert st[19] >= 1 and st[22] == 1.0
        levels[leaf] = st[19]
        first = _kkt_solve(m, rhs)
        oq.update_settings(m, rho=0.29)
        sols[leaf] = (first, _kkt_solve(m, rhs))
        oq.clean(m)
    for leaf in ("1", "2"):
        for k in range(2):
            ref, v = sols["0"][k], sols[leaf][k]
            assert np.all(np.isfinite(v))
            assert np.max(np.abs(v - ref)) <= 1e-9 * max(1.0, np.max(np.abs(ref))), (case, leaf, k, np.max(np.abs(v - ref)))
    if case == "control-400":
        assert levels["1"] >= levels["0"]  # (the leaves are a level of their own below the subtrees)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["lds-fronts", "global-fronts", "global-fronts-no-tree"])
@pytest.mark.parametrize("kmax", [24, 150])
@pytest.mark.parametrize("case", ["control-400", "grid2d-40", "grid3d-9", "portfolio", "random-200"])
def test_dense_top_over_the_supernodes_gives_the_level_factor(product_lib, monkeypatch, case, kmax, variant):
    """Round 6: a dense top block OVER the supernode partition (csrc/direct_sndense_kernels.hpp): the supernodes of the last
    levels are not factorised by fronts -- their Schur complement is assembled from the boundary children's update matrices,
    inverted by the block sweeps, and a solve multiplies by it once instead of walking the chain of those levels in both
    directions.  Forced on small problems (OSQP_AMD_SN_DENSE=2, at most `kmax` pivots), below LDS fronts, below fronts out of
    global memory whose top part hands front vectors up (the one-launch tree), and with one launch per level (no vectors):
    the KKT solves must be those of the level-by-level factor, before and after a rho update, and the inertia count must
    still add up (setup succeeds)."""
    make, smax = CASES[case]
    prob = make()
    monkeypatch.setenv("OSQP_AMD_SNODE", "2")
    monkeypatch.setenv("OSQP_AMD_SNODE_MAX", str(min(smax, 16)))
    n, mm = prob["P"].shape[0], prob["A"].shape[0]
    rhs = np.random.default_rng(15).standard_normal(n + mm)
    sols, dense = {}, {}
    for mode in ("level", "dense"):
        monkeypatch.setenv("OSQP_AMD_MF", "0" if mode == "level" else "1")
        monkeypatch.setenv("OSQP_AMD_SN_DENSE", "0" if mode == "level" else "2")
        monkeypatch.setenv("OSQP_AMD_SN_DENSE_MAX", str(kmax))
        monkeypatch.setenv("OSQP_AMD_MF_SHARE_MIN", "3" if kmax == 24 else "100000000")  # (boundary children's update matrices are never released)
        if mode == "dense" and variant != "lds-fronts":
            monkeypatch.setenv("OSQP_AMD_MF_MAX_FRONT", "12")
            monkeypatch.setenv("OSQP_AMD_SNODE_TOP", "1")  # (front vectors below a dense top are opt-in: the rows of D must take them)
        if mode == "dense" and variant == "global-fronts-no-tree":
            monkeypatch.setenv("OSQP_AMD_SNODE_TREE", "0")
        m = oq.Model(product_lib)
        oq.setup(m, linsys_solver="direct", verbose=False, adaptive_rho=False, **prob)
        st = oq.stats(m)
        assert st[19] >= 1 and st[22] == float(mode == "dense"), (st[19], st[22])
        dense[mode] = st[25]
        first = _kkt_solve(m, rhs)
        oq.update_settings(m, rho=0.731)
        second = _kkt_solve(m, rhs)
        third = _kkt_solve(m, rhs)
        assert np.array_equal(second, third)  # (the counters of the one-launch tree are back at rest; a fixed order of sums)
        sols[mode] = (first, second)
        oq.clean(m)
    assert dense["level"] == 0 and 1 <= dense["dense"] <= kmax, dense
    for k in range(2):
        a, b = sols["level"][k], sols["dense"][k]
        assert np.all(np.isfinite(b))
        assert np.max(np.abs(a - b)) <= 1e-9 * max(1.0, np.max(np.abs(a))), (case, k, np.max(np.abs(a - b)), np.max(np.abs(a)))
    assert np.max(np.abs(sols["dense"][0] - sols["dense"][1])) > 1e-6


@pytest.mark.gpu
def test_pivot_block_in_registers_is_the_lds_sweep(product_lib, monkeypatch):
    """Round 6: the pivot block of a block Gauss-Jordan step swept in registers (csrc/direct_dense_kernels.hpp k_gj_pivot_r:
    pivot row through LDS, pivot column by v_readlane) uses the expression of the LDS form element by element: the inverse --
    hence every KKT solve -- has the same bits.  Through the dense top over the supernodes (always block sweeps, 3 steps here).
    So do the sweeps restricted to the tiles the block pattern says can change (direct.hip gj_symbolic): what they skip are
    subtractions of exact zeros."""
    prob = qp_zoo.grid2d(40)
    monkeypatch.setenv("OSQP_AMD_SNODE", "2")
    monkeypatch.setenv("OSQP_AMD_SNODE_MAX", "16")
    monkeypatch.setenv("OSQP_AMD_SN_DENSE", "2")
    monkeypatch.setenv("OSQP_AMD_SN_DENSE_MAX", "150")
    n, mm = prob["P"].shape[0], prob["A"].shape[0]
    rhs = np.random.default_rng(16).standard_normal(n + mm)
    sols = {}
    for lds, sparse, fuse in (("0", "1", "1"), ("1", "1", "1"), ("0", "0", "1"), ("0", "1", "0")):
        monkeypatch.setenv("OSQP_AMD_GJ_PIVOT_LDS", lds)
        monkeypatch.setenv("OSQP_AMD_GJ_SPARSE", sparse)  # 0: every tile in every step (the block pattern of the sweeps not used)
        monkeypatch.setenv("OSQP_AMD_GJ_FUSE", fuse)      # 0: the pivot block swept by a launch of its own, not inside the previous update
        m = oq.Model(product_lib)
        oq.setup(m, linsys_solver="direct", verbose=False, adaptive_rho=False, **prob)
        assert oq.stats(m)[25] > 64
        sols[lds, sparse, fuse] = _kkt_solve(m, rhs)
        oq.clean(m)
    ref = sols["0", "1", "1"]
    assert np.all(np.isfinite(ref))
    for key, v in sols.items():
        assert np.array_equal(ref, v), key
