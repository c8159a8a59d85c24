"""Host-side symbolic analysis of the direct back-end (csrc/symbolic.hip) through osqp_amd_symbolic_probe: orderings,
level schedule and the supernode partition of the triangular solves.  No device work: runs on the CPU box."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

import osqp_jl_amd as oq
import qp_zoo


def probe(lib, prob, ordering, smax=64):
    P = sp.triu(sp.csc_matrix(prob["P"])).tocsc()
    A = sp.csc_matrix(prob["A"])
    P.sort_indices(); A.sort_indices()
    n, m = P.shape[0], A.shape[0]
    arrs = [np.ascontiguousarray(a, dtype=np.int64) for a in (P.indptr, P.indices, A.indptr, A.indices)]
    out = np.zeros(15)
    ptrs = [a.ctypes.data_as(C.POINTER(C.c_longlong)) for a in arrs]
    rc = lib.osqp_amd_symbolic_probe(n, m, *ptrs, ordering, smax, out.ctypes.data_as(C.POINTER(C.c_double)), 15)
    assert rc == 0
    return dict(N=int(out[0]), nnzL=int(out[1]), levels=int(out[2]), supernodes=int(out[3]), sn_levels=int(out[4]),
                outside=int(out[5]), block_doubles=int(out[6]), largest=int(out[7]), ok=bool(out[8]), inside=int(out[9]),
                cost_levels=out[10], cost_supernodes=out[11], pays=bool(out[12]), graph_depth=int(out[13]), lean_same=bool(out[14]))


@pytest.mark.parametrize("ordering", [0, 1, 2])
@pytest.mark.parametrize("smax", [1, 3, 16, 64])
@pytest.mark.parametrize("name", sorted(qp_zoo.ZOO))
def test_supernode_partition_invariants(product_lib, name, ordering, smax):
    """Slots are a permutation, supernodes are numbered level by level, every entry of L is either below the diagonal
    of its supernode's block or points to a supernode of a strictly lower level (checked inside the probe)."""
    r = probe(product_lib, qp_zoo.ZOO[name](), ordering, smax)
    n, m = qp_zoo.ZOO[name]()["P"].shape[0], qp_zoo.ZOO[name]()["A"].shape[0]
    assert r["ok"] and r["N"] == n + m
    # round 5: a lean analysis (rows of the pattern only; csrc/symbolic.hpp) completed on the host IS the full analysis, and
    # its supernode partition is the full one's with list lengths that bound the real ones
    assert r["lean_same"]
    assert r["largest"] <= smax and r["inside"] + r["outside"] == r["nnzL"]
    assert r["sn_levels"] <= r["levels"]
    if smax == 1:  # singletons: nothing inside a block and the supernode graph is the elimination tree
        assert r["inside"] == 0 and r["supernodes"] == r["N"] and r["sn_levels"] == r["levels"]


@pytest.mark.parametrize("leaf", ["1", "2"])
@pytest.mark.parametrize("name", sorted(qp_zoo.ZOO))
def test_lone_leaves_keep_the_partition_invariants(product_lib, monkeypatch, name, leaf):
    """OSQP_AMD_SNODE_LEAF: leaves with a one-entry column as supernodes of their own (by default from 400 000 on)."""
    monkeypatch.setenv("OSQP_AMD_SNODE_LEAF", "0")
    off = probe(product_lib, qp_zoo.ZOO[name](), 0, 16)
    monkeypatch.setenv("OSQP_AMD_SNODE_LEAF", leaf)
    on = probe(product_lib, qp_zoo.ZOO[name](), 0, 16)
    assert on["ok"] and on["lean_same"] and on["nnzL"] == off["nnzL"] and on["supernodes"] >= off["supernodes"]
    if leaf == "1":
        assert on["block_doubles"] <= off["block_doubles"]


def test_supernodes_on_random_patterns(product_lib):
    rng = np.random.default_rng(7)
    for _ in range(30):
        n, m = int(rng.integers(1, 40)), int(rng.integers(0, 50))
        M = sp.random(n, n, density=0.15, random_state=rng)
        prob = dict(P=(M @ M.T).tocsc(), A=sp.random(m, n, density=0.2, random_state=rng, format="csc"))
        for ordering in (0, 1, 2):
            r = probe(product_lib, prob, ordering, int(rng.integers(1, 12)))
            assert r["ok"] and r["lean_same"], (n, m, ordering, r)


def test_long_horizon_control_takes_supernodes(product_lib):
    """The class the level schedule is slow on (a single banded multi-stage problem): nested dissection
    leaves ~300 pivot levels, the supernode graph ~15, and the engine's rule picks it; the classes with a dense top
    block keep the level schedule + dense product."""
    r = probe(product_lib, qp_zoo.control(nx=12, nu=6, T=800), 1)
    assert r["ok"] and r["levels"] > 200 and r["sn_levels"] <= 20 and r["pays"]
    for name, kw in (("portfolio", dict(n=4000, k=100)), ("svm", dict(n=100, m=4000)), ("lasso_data", dict(n=200, m=4000))):
        r = probe(product_lib, qp_zoo.ZOO[name](**kw), 0)
        assert r["ok"] and not r["pays"], (name, r)


@pytest.mark.parametrize("ordering", [0, 1])
def test_threaded_analysis_is_the_single_threaded_one(product_lib, monkeypatch, ordering):
    """Round 4: nested dissection hands the two sides of a large piece to two threads, the pattern of L and the supernode
    lists are built by row / column ranges on host threads (control-1e6: 4.7 -> 2.0 s).  Forced onto small problems
    (OSQP_AMD_HOST_THREADS) every number the probe reports -- fill, levels, the supernode partition and the invariants it
    checks entry by entry -- is the single-threaded one."""
    probs = [qp_zoo.control(nx=12, nu=6, T=300), qp_zoo.portfolio(n=1500, k=40), qp_zoo.svm(n=60, m=1500)]
    for prob in probs:
        monkeypatch.setenv("OSQP_AMD_HOST_THREADS", "1")
        one = probe(product_lib, prob, ordering)
        for nt in ("3", "8"):
            monkeypatch.setenv("OSQP_AMD_HOST_THREADS", nt)
            assert probe(product_lib, prob, ordering) == one
        assert one["ok"]


def test_graph_depth_separates_long_graphs_from_random_ones(product_lib):
    """What sends nested dissection first on large problems (csrc/direct.hip: breadth-first depth of the KKT graph >= 400):
    a multi-stage control problem is as deep as its horizon, the dense-row classes a handful of levels."""
    assert probe(product_lib, qp_zoo.control(nx=8, nu=4, T=500), 0)["graph_depth"] >= 500
    for name, kw in (("portfolio", dict(n=1500, k=40)), ("svm", dict(n=60, m=1500)), ("equality_qp", dict(n=300))):
        assert probe(product_lib, qp_zoo.ZOO[name](**kw), 0)["graph_depth"] <= 12, name
