"""Full-size parity on BASELINE.json's affordable configurations (bench.py's seed and settings): the HIP engine
against the CPU oracle AND against an independent numpy evaluation of OSQP's stopping criteria on unscaled,
host-regenerated data.  rand-1e6 itself: the host-side evaluation of the engine's solution runs here (the box's host
holds the 24 GB instance); the 25-iteration iterate comparison with the CPU oracle (10 minutes) is the committed record
profiles/r03_rand1e6_parity.json; its size-independent properties are in test_gpu_parity.py.  Tolerances: the solver's own eps (1e-4 requested; x and y
compared at 2e-4 * scale, the north-star's statement), iteration counts within one termination check (25)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

import bench
import osqp_jl_amd as oq
from osqp_jl_amd import batch
from test_gpu_parity import _data_to_scipy

pytestmark = pytest.mark.gpu


def kkt_check(oracle_lib, kind, n, k, seed, x, y, eps=1e-4, slack=2.0):
    """OSQP's stopping criteria re-evaluated in numpy from host-generated, unscaled data."""
    d = oracle_lib.oracle_generate(kind, n, k, seed)
    P, q, A, l, u = _data_to_scipy(d.contents)
    oracle_lib.oracle_data_free(d)
    Pfull = P + sp.triu(P, 1).T
    Ax = A @ x
    z = np.clip(Ax, l, u)
    Px, Aty = Pfull @ x, A.T @ y
    pri = np.max(np.abs(Ax - z))
    dua = np.max(np.abs(Px + q + Aty))
    eps_pri = eps + eps * max(np.max(np.abs(Ax)), np.max(np.abs(z)))
    eps_dua = eps + eps * max(np.max(np.abs(Px)), np.max(np.abs(Aty)), np.max(np.abs(q)))
    assert pri <= slack * eps_pri, (pri, eps_pri)
    assert dua <= slack * eps_dua, (dua, eps_dua)
    # dual sign convention: y < 0 only where the lower bound is active, y > 0 only where the upper one is
    tol = 1e-3 * max(1.0, np.max(np.abs(y)))
    assert np.all((y > -tol) | (Ax - l < 10 * slack * eps_pri)) and np.all((y < tol) | (u - Ax < 10 * slack * eps_pri))
    return 0.5 * x @ (Pfull @ x) + q @ x


def test_control_1e6_matches_oracle_at_full_size(product_lib, oracle_lib, monkeypatch):
    """Round 4: the trisolve path at scale (bench.py `control-1e6`: n = 1 000 002, m = 1 666 674, nnz(L) = 6.7e7, nested
    dissection, supernodal solves with the wavefront forms and the top-of-tree launch, packed blocks, numeric factorisation
    by bisection on long-row levels) against the CPU oracle on the same host-built data: an exact solve on both sides, so the
    same status and the SAME iteration count, x / y to 2e-4 of scale, objective; then OSQP's stopping criteria and the dual
    sign convention re-evaluated in numpy on the unscaled data."""
    import qp_zoo

    monkeypatch.delenv("OSQP_AMD_FIRST_ORDERING", raising=False)  # the library picks nested dissection first by the graph's depth
    T = bench.WORKLOADS["control-1e6"][1]
    prob = bench.control_problem(T)
    res = []
    for lib, ls in ((product_lib, "direct"), (oracle_lib, "qdldl")):
        m = oq.Model(lib)
        oq.setup(m, linsys_solver=ls, **prob, **bench.SETTINGS)
        if lib is product_lib:
            st = oq.stats(m)
            assert st[0] == 0 and st[19] > 2  # direct back-end, supernodal solves
        res.append(oq.solve(m))
        oq.clean(m)
    rp, ro = res
    assert rp.info.status == ro.info.status == "Solved"
    assert rp.info.iter == ro.info.iter, (rp.info.iter, ro.info.iter)
    assert np.max(np.abs(rp.x - ro.x)) <= 2e-4 * max(1.0, np.max(np.abs(ro.x)))
    assert np.max(np.abs(rp.y - ro.y)) <= 2e-4 * max(1.0, np.max(np.abs(ro.y)))
    assert abs(ro.info.obj_val - rp.info.obj_val) <= 1e-4 * max(1.0, abs(ro.info.obj_val))
    pri, eps_pri, dua, eps_dua = qp_zoo.kkt_check(prob, rp.x, rp.y, 1e-4)
    assert pri <= 2.0 * eps_pri and dua <= 2.0 * eps_dua, (pri, eps_pri, dua, eps_dua)
    Ax = prob["A"] @ rp.x
    tol = 1e-3 * max(1.0, np.max(np.abs(rp.y)))
    assert np.all((rp.y > -tol) | (Ax - prob["l"] < 20 * eps_pri)) and np.all((rp.y < tol) | (prob["u"] - Ax < 20 * eps_pri))


def test_grid2d_700_matches_oracle_at_full_size(product_lib, oracle_lib, monkeypatch):
    """Round 6: the direct back-end on a structure it was NOT tuned on -- a 700 x 700 grid QP (bench.py `grid2d-5e5`,
    tests/qp_zoo.py grid2d: n = m = 490 000, a 2-D KKT graph with separators of ~700 nodes).  Up to round 5 one front above
    192 rows sent the whole factorisation back to two launches per pivot level, and the level-structure dissection carried a
    pendant constraint row per separator node (twice the fill of minimum degree).  Now: the dissection is the ordering taken
    (by the graph's depth), its fill is below minimum degree's 1.96e7 entries, the factorisation is the multifrontal one with
    its large fronts out of global memory (stats[22]), the index arrays are built on the device from a lean host analysis
    (stats[23]) -- and the solve is the CPU oracle's: same status, the SAME iteration count, x / y to 2e-4 of scale, objective,
    OSQP's stopping criteria re-evaluated in numpy on the unscaled data."""
    import qp_zoo

    monkeypatch.delenv("OSQP_AMD_FIRST_ORDERING", raising=False)
    prob = qp_zoo.grid2d(bench.WORKLOADS["grid2d-5e5"][1])
    res = []
    for lib, ls in ((product_lib, "direct"), (oracle_lib, "qdldl")):
        m = oq.Model(lib)
        oq.setup(m, linsys_solver=ls, **prob, **bench.SETTINGS)
        if lib is product_lib:
            st = oq.stats(m)
            assert st[0] == 0 and st[19] > 2  # direct back-end, supernodal solves
            assert st[22] == 1.0 and st[23] == 1.0, (st[22], st[23])  # multifrontal factorisation, device-built index arrays
            assert st[4] <= 1.96e7, st[4]  # nnz(L) of the ordering taken: not above minimum degree's
        res.append(oq.solve(m))
        if lib is product_lib:  # a rho update refactors [REF src/interface.jl:539-550]: the second solve must still be the oracle's answer
            oq.update_settings(m, rho=0.3)
            res.append(oq.solve(m))
            # the one-launch tree of this structure holds more workgroups than are resident at once (csrc/direct.hip: up to twice):
            # no wait inside it may have timed out (slot 21 counts the restarts on the per-level kernels)
            assert oq.stats(m)[21] == 0
        oq.clean(m)
    rp, rp2, ro = res
    assert rp.info.status == ro.info.status == rp2.info.status == "Solved"
    assert rp.info.iter == ro.info.iter, (rp.info.iter, ro.info.iter)
    for r in (rp, rp2):
        assert np.max(np.abs(r.x - ro.x)) <= 2e-4 * max(1.0, np.max(np.abs(ro.x)))
        assert np.max(np.abs(r.y - ro.y)) <= 2e-4 * max(1.0, np.max(np.abs(ro.y)))
    assert abs(ro.info.obj_val - rp.info.obj_val) <= 1e-4 * max(1.0, abs(ro.info.obj_val))
    pri, eps_pri, dua, eps_dua = qp_zoo.kkt_check(prob, rp.x, rp.y, 1e-4)
    assert pri <= 2.0 * eps_pri and dua <= 2.0 * eps_dua, (pri, eps_pri, dua, eps_dua)


def test_grid3d_36_matches_oracle(product_lib, oracle_lib, monkeypatch):
    """Round 6: a second structure nobody tuned for -- a 36 x 36 x 36 grid QP (tests/qp_zoo.py grid3d: 7-point Laplacian, a box
    on every variable and a band on the difference of neighbours along one axis: rows of A with two entries; N = 1.4e5 pivots,
    separators are planes of ~1 300 nodes, fronts of ~2 000 rows).  Default settings of the direct back-end: the multifrontal
    factorisation with its fronts out of global memory, update matrices placed by lifetimes, a dense top over the supernode
    partition (stats[25]) -- and the CPU oracle's solve: same status, the same iteration count, x / y to 1e-6 of scale."""
    import qp_zoo

    monkeypatch.delenv("OSQP_AMD_FIRST_ORDERING", raising=False)
    prob = qp_zoo.grid3d(36)
    res = []
    for lib, ls in ((product_lib, "direct"), (oracle_lib, "qdldl")):
        m = oq.Model(lib)
        oq.setup(m, linsys_solver=ls, **prob, **bench.SETTINGS)
        if lib is product_lib:
            st = oq.stats(m)
            assert st[0] == 0 and st[19] > 2 and st[22] == 1.0 and st[25] >= 512, (st[0], st[19], st[22], st[25])
        res.append(oq.solve(m))
        if lib is product_lib:
            oq.update_settings(m, rho=0.3)
            res.append(oq.solve(m))
        oq.clean(m)
    rp, rp2, ro = res
    assert rp.info.status == ro.info.status == rp2.info.status == "Solved"
    assert rp.info.iter == ro.info.iter, (rp.info.iter, ro.info.iter)
    assert np.max(np.abs(rp.x - ro.x)) <= 1e-6 * max(1.0, np.max(np.abs(ro.x)))
    assert np.max(np.abs(rp.y - ro.y)) <= 1e-6 * max(1.0, np.max(np.abs(ro.y)))
    assert np.max(np.abs(rp2.x - ro.x)) <= 2e-4 * max(1.0, np.max(np.abs(ro.x)))
    pri, eps_pri, dua, eps_dua = qp_zoo.kkt_check(prob, rp.x, rp.y, 1e-4)
    assert pri <= 2.0 * eps_pri and dua <= 2.0 * eps_dua, (pri, eps_pri, dua, eps_dua)


@pytest.mark.parametrize("name", ["rand-1e5", "lasso-5e5"])
def test_bench_config_matches_oracle_at_full_size(product_lib, oracle_lib, name):
    kind, n, k, linsys = bench.WORKLOADS[name]
    res = []
    for lib in (product_lib, oracle_lib):
        m = oq.Model(lib)
        oq.setup_generated(m, kind, n, k, 1, linsys_solver=linsys, **bench.SETTINGS)
        res.append(oq.solve(m))
        oq.clean(m)
    rp, ro = res
    assert rp.info.status == ro.info.status == "Solved"
    assert abs(rp.info.iter - ro.info.iter) <= 25, (rp.info.iter, ro.info.iter)
    assert np.max(np.abs(rp.x - ro.x)) <= 2e-4 * max(1.0, np.max(np.abs(ro.x)))
    assert np.max(np.abs(rp.y - ro.y)) <= 2e-4 * max(1.0, np.max(np.abs(ro.y)))
    obj = kkt_check(oracle_lib, kind, n, k, 1, rp.x, rp.y)
    assert abs(obj - rp.info.obj_val) <= 1e-6 * max(1.0, abs(obj))
    assert abs(ro.info.obj_val - rp.info.obj_val) <= 1e-4 * max(1.0, abs(ro.info.obj_val))


def test_pcg_tolerance_rule_does_not_bend_the_answer(product_lib, oracle_lib, monkeypatch):
    """The inexact-ADMM rule is the builder's own: solve rand-1e5 with the rule's constant
    tightened 100x (every inner solve ~2 digits more accurate) and require the same solution to 2e-4."""
    kind, n, k, linsys = bench.WORKLOADS["rand-1e5"]
    out = []
    for lam in (None, "0.00015"):
        if lam is None:
            monkeypatch.delenv("OSQP_AMD_PCG_LAMBDA", raising=False)
        else:
            monkeypatch.setenv("OSQP_AMD_PCG_LAMBDA", lam)
        m = oq.Model(product_lib)
        oq.setup_generated(m, kind, n, k, 1, linsys_solver=linsys, **bench.SETTINGS)
        st0 = oq.stats(m)
        r = oq.solve(m)
        out.append((r, oq.stats(m)[6] - st0[6]))
        oq.clean(m)
    (ra, cga), (rb, cgb) = out
    assert ra.info.status == rb.info.status == "Solved"
    assert cgb > 1.3 * cga  # the tight rule really did more inner work
    assert np.max(np.abs(ra.x - rb.x)) <= 2e-4 * max(1.0, np.max(np.abs(rb.x)))
    assert np.max(np.abs(ra.y - rb.y)) <= 2e-4 * max(1.0, np.max(np.abs(rb.y)))
    kkt_check(oracle_lib, kind, n, k, 1, rb.x, rb.y)


def test_mpc_batch_all_4096_instances(product_lib, oracle_lib):
    """Config 5 at full size on one device, bench.py's settings (eps = 1e-4 on both sides):
      * every one of the 4096 instances Solved, and OSQP's stopping criteria (primal and dual residual on unscaled,
        host-regenerated data) plus the dual sign convention re-evaluated on the host for EVERY instance;
      * 256 instances spread over the range (every 16th) compared with the oracle one by one at the tolerance of the
        other full-size configurations: x, y within 2e-4 * scale, termination at most one check (25 iterations) apart.
        The kernel solves the reduced system (P + sigma I + A' rho A) x~ = b with an explicit inverse where the oracle
        factorises the quasi-definite KKT matrix: same iteration in exact arithmetic, rounding apart."""
    total, seed = 4096, 1
    opts = dict(bench.SETTINGS)
    solver = batch.device_mpc_solver(product_lib, 0, **opts)
    x, y, info = batch.solve_mpc_sharded(solver, total, seed)
    x, y, info = x.cpu().numpy(), y.cpu().numpy(), info.cpu().numpy()
    assert np.all(info[:, 1] == 1), np.unique(info[:, 1], return_counts=True)
    assert np.all(np.isfinite(x)) and np.all(np.isfinite(y))
    eps = 1e-4
    worst = dict(pri=0.0, dua=0.0, dx=0.0, dy=0.0, dit=0)
    for i in range(total):
        d = oracle_lib.oracle_generate(2, 100, i, seed)
        P, q, A, l, u = _data_to_scipy(d.contents)
        oracle_lib.oracle_data_free(d)
        Pfull = P + sp.triu(P, 1).T
        Ax, Px, Aty = A @ x[i], Pfull @ x[i], A.T @ y[i]
        z = np.clip(Ax, l, u)
        pri, dua = np.max(np.abs(Ax - z)), np.max(np.abs(Px + q + Aty))
        eps_pri = eps + eps * max(np.max(np.abs(Ax)), np.max(np.abs(z)))
        eps_dua = eps + eps * max(np.max(np.abs(Px)), np.max(np.abs(Aty)), np.max(np.abs(q)))
        assert pri <= 2 * eps_pri and dua <= 2 * eps_dua, (i, pri, eps_pri, dua, eps_dua)
        worst["pri"], worst["dua"] = max(worst["pri"], pri / eps_pri), max(worst["dua"], dua / eps_dua)
        tol = 1e-3 * max(1.0, np.max(np.abs(y[i])))
        assert np.all((y[i] > -tol) | (Ax - l < 20 * eps_pri)) and np.all((y[i] < tol) | (u - Ax < 20 * eps_pri)), i
        if i % 16:
            continue
        m = oq.Model(oracle_lib)
        oq.setup(m, P=P, q=q, A=A, l=l, u=u, **opts)
        r = oq.solve(m)
        oq.clean(m)
        assert r.info.status == "Solved"
        dit = abs(r.info.iter - int(info[i, 0]))
        dx = np.max(np.abs(x[i] - r.x)) / max(1.0, np.max(np.abs(r.x)))
        dy = np.max(np.abs(y[i] - r.y)) / max(1.0, np.max(np.abs(r.y)))
        assert dit <= 25, (i, r.info.iter, info[i, 0])
        assert dx <= 2e-4 and dy <= 2e-4, (i, dx, dy, r.info.iter, info[i, 0])
        worst["dx"], worst["dy"], worst["dit"] = max(worst["dx"], dx), max(worst["dy"], dy), max(worst["dit"], dit)
    print("mpc-batch parity, worst over the batch:", worst)


def _mem_available_gib():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                return int(line.split()[1]) / 1048576.0
    except OSError:
        pass
    return 0.0


@pytest.mark.skipif(os.environ.get("OSQP_AMD_SKIP_RAND1E6") == "1" or _mem_available_gib() < 80.0,
                    reason="the headline instance needs ~40 GiB of host memory for the host-side evaluation (and 60 GB of HBM)")
def test_rand1e6_solution_meets_osqp_criteria_on_host_regenerated_data(product_lib, oracle_lib):
    """BASELINE.json's headline configuration (n = m = 1e6, nnz(A) = 1e9) itself: the engine's cold-start solution at
    bench.py's seed and settings, re-evaluated on the HOST -- (P, q, A, l, u) regenerated by oracle/gen.c (data the
    engine never saw: it generates its own copy in HBM), OSQP's stopping criteria and dual sign convention in scipy fp64
    on the unscaled data, as kkt_check does for the smaller configurations.  The iterate-level comparison with the CPU
    oracle on this instance (25 iterations: 10 minutes of host time) is the committed record
    profiles/r03_rand1e6_parity.json (tools/cpu_rand1e6.py): max |dx| / max |x| = 5e-14 after W + K = 25 iterations."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import cpu_rand1e6

    kind, n, k, linsys = bench.WORKLOADS["rand-1e6"]
    m = oq.Model(product_lib)
    oq.setup_generated(m, kind, n, k, 1, linsys_solver=linsys, **bench.SETTINGS)
    r = oq.solve(m)
    oq.clean(m)
    assert r.info.status == "Solved"
    res = cpu_rand1e6.host_kkt(oracle_lib, n, k, 1, r.x, r.y)
    print("rand-1e6 on host data:", {a: res[a] for a in ("pri_over_eps", "dua_over_eps", "dual_sign_violations", "objective")})
    assert res["nnz_A"] == 10**9
    assert res["pri_res"] <= 2 * res["eps_pri"] and res["dua_res"] <= 2 * res["eps_dua"]
    assert res["dual_sign_violations"] == 0
    assert abs(res["objective"] - r.info.obj_val) <= 1e-6 * max(1.0, abs(res["objective"]))
    # the engine's own residuals are the ones the host sees (unscaled termination)
    assert abs(res["dua_res"] - r.info.dua_res) <= 1e-6 * max(res["eps_dua"], 1e-300)


@pytest.mark.skipif(os.environ.get("OSQP_AMD_TEST_HOST_SETUP") != "1" or _mem_available_gib() < 120.0,
                    reason="set OSQP_AMD_TEST_HOST_SETUP=1 on a host with 120 GiB free: 24 GB of CSC arrays through osqp_setup (~2 minutes)")
def test_rand1e6_through_osqp_setup_from_host_arrays(product_lib, oracle_lib):
    """The reference entry point at the headline size [REF src/interface.jl:113-155]: the rand-1e6 instance built on the
    host (oracle/gen.c), handed to osqp_setup as CSC arrays, must solve to the very solution of the device-generated
    instance of the same seed (same arithmetic behind two front doors)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("host_setup_rand1e6", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "host_setup_rand1e6.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rec = mod.run()
    assert rec["status"] == rec["generated_status"] == "Solved"
    assert rec["iter"] == rec["generated_iter"]
    assert rec["bit_identical"], (rec["max_abs_dx"], rec["max_abs_dy"])


@pytest.mark.skipif(os.environ.get("OSQP_AMD_SKIP_RAND1E6") == "1" or _mem_available_gib() < 80.0,
                    reason="the CPU oracle on the headline instance holds ~46 GiB of host memory (and needs ~3 minutes)")
def test_rand1e6_iterates_match_oracle(product_lib, oracle_lib):
    """Round-4 review, item 7a: the iterate-level comparison on BASELINE.json's headline configuration inside the suite the
    driver runs (it used to be a builder-committed record only).  Three ADMM iterations from the cold start at bench.py's
    seed and settings on the HIP engine and on the CPU oracle: identical CG iteration counts, iterates equal to 1e-11
    relative (measured 6e-14 / 2e-13 after 25 iterations: profiles/r04_rand1e6_parity.json)."""
    kind, n, k, linsys = bench.WORKLOADS["rand-1e6"]
    out = {}
    for name, lib, ls in (("engine", product_lib, linsys), ("oracle", oracle_lib, "pcg")):
        m = oq.Model(lib)
        oq.setup_generated(m, kind, n, k, 1, linsys_solver=ls, **bench.SETTINGS)
        assert lib.osqp_amd_iterate(m.workspace, 3) == 0
        nn, mm = oq.dimensions(m)
        x, y = np.empty(nn), np.empty(mm)
        assert lib.osqp_amd_get_iterate(m.workspace, x.ctypes.data_as(C.POINTER(C.c_double)), y.ctypes.data_as(C.POINTER(C.c_double))) == 0
        st = oq.stats(m)
        out[name] = (x, y, int(st[6]), int(st[1]))
        oq.clean(m)
    (xe, ye, cge, nnze), (xo, yo, cgo, nnzo) = out["engine"], out["oracle"]
    assert nnze == nnzo == 10**9
    assert cge == cgo and cge > 0, (cge, cgo)
    dx = np.max(np.abs(xe - xo)) / max(np.max(np.abs(xo)), 1e-300)
    dy = np.max(np.abs(ye - yo)) / max(np.max(np.abs(yo)), 1e-300)
    print("rand-1e6 after 3 iterations: CG %d = %d, max|dx|/max|x| = %.2e, max|dy|/max|y| = %.2e" % (cge, cgo, dx, dy))
    assert dx <= 1e-11 and dy <= 1e-11, (dx, dy)
