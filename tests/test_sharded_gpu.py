"""Row-sharded solve of one QP against the single-device solve of the same problem.

Several ranks share the one GPU of the test box through the host-staged transport (gloo); the RCCL transport is
exercised with a one-rank communicator (RCCL refuses two ranks on one device).  Everything but the wire is the
code the multi-GPU run uses."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_ranks(tmp_path, world, transport, case, settings=None, extra=()):
    out = str(tmp_path / "out.json")
    port = str(_free_port())
    procs = []
    for r in range(world):
        cmd = [sys.executable, os.path.join(HERE, "_sharded_worker.py"), str(r), str(world), port, out, transport, case,
               json.dumps(settings or {})] + list(extra)
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace"))
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, logs[r][-4000:])
    return [json.load(open("%s.%d" % (out, r))) for r in range(world)]


SETTINGS = dict(eps_abs=1e-6, eps_rel=1e-6, adaptive_rho_interval=25, verbose=False, linsys_solver="pcg")


def single(product_lib, kind, n, per_row, seed, second=False):
    import osqp_jl_amd as oq
    m = oq.Model(product_lib)
    oq.setup_generated(m, kind, n, per_row, seed, **SETTINGS)
    r = oq.solve(m)
    rec = dict(status=r.info.status, iter=r.info.iter, obj=r.info.obj_val, x=r.x.copy(), y=r.y.copy())
    if second:
        nn, mm = oq.dimensions(m)
        oq.update_q(m, np.random.default_rng(5).standard_normal(nn))
        oq.warm_start(m, x=r.x.copy(), y=r.y.copy())
        r2 = oq.solve(m)
        rec["second"] = dict(status=r2.info.status, iter=r2.info.iter, obj=r2.info.obj_val, x=r2.x.copy())
    oq.clean(m)
    return rec


@pytest.mark.parametrize("world,kind,n,per_row", [(2, 0, 3000, 12), (3, 0, 2501, 9), (2, 1, 900, 0), (4, 0, 40000, 256)])
def test_sharded_matches_single(product_lib, tmp_path, world, kind, n, per_row):
    ref = single(product_lib, kind, n, per_row, 7, second=True)
    recs = run_ranks(tmp_path, world, "host", "gen:%d:%d:%d:7" % (kind, n, per_row), SETTINGS, extra=["--second"])
    for rec in recs:
        assert rec["status"] == ref["status"] == "Solved"
        # same algorithm, same decisions; only the summation order of dot products differs (rank-ordered partial sums)
        assert abs(rec["iter"] - ref["iter"]) <= 25
        assert abs(rec["obj"] - ref["obj"]) <= 1e-5 * max(1.0, abs(ref["obj"]))
        assert np.max(np.abs(np.array(rec["x"]) - ref["x"])) <= 1e-4 * max(1.0, np.max(np.abs(ref["x"])))
        assert np.max(np.abs(np.array(rec["y"]) - ref["y"])) <= 1e-4 * max(1.0, np.max(np.abs(ref["y"])))
        assert rec["second"]["status"] == ref["second"]["status"]
        assert abs(rec["second"]["obj"] - ref["second"]["obj"]) <= 1e-5 * max(1.0, abs(ref["second"]["obj"]))
        assert rec["stats"][13] == world and rec["stats"][14] > 0
    # every rank holds the same full solution, bit for bit
    for rec in recs[1:]:
        assert rec["x"] == recs[0]["x"] and rec["y"] == recs[0]["y"] and rec["iter"] == recs[0]["iter"]


def test_sharded_setup_keeps_only_the_row_block(tmp_path):
    """A rank looks at the problem column range by column range and keeps its row blocks of A, A' and P only
    (csrc/engine.hip setup_sharded): the high-water mark of its device memory falls with the number of ranks."""
    case = "gen:0:40000:256:7"
    one = run_ranks(tmp_path, 1, "host", case, SETTINGS)[0]
    four = run_ranks(tmp_path, 4, "host", case, SETTINGS)
    assert one["status"] == "Solved" and all(r["status"] == "Solved" for r in four)
    peak1 = one["stats"][20]
    assert peak1 > 3e8  # ~1e7 non-zeros in each of A, A', P
    for r in four:
        assert r["stats"][20] <= 0.4 * peak1, (r["stats"][20], peak1)


SHARDABLE_CASES = [
    "case_basic_qp", "case_update_q", "case_update_l", "case_update_u", "case_update_max_iter",
    "case_update_check_termination", "case_update_rho", "case_time_limit", "case_non_convex_big_sigma",
    "case_dual_infeasible_lp", "case_dual_infeasible_qp", "case_primal_dual_infeasible_warm",
    "case_primal_dual_infeasible_cold", "case_primal_infeasible_random", "case_unconstrained", "case_feasibility",
    "case_warm_start", "case_moi_lp", "case_equality_lsq", "case_bounds_validation",
    # round 4: polish on a row block (the iterative form on the operator of the indirect back-end, csrc/pcg.hip)
    "case_polish_basic", "case_polish_unconstrained", "case_polish_random",
]


def test_reference_known_answers_sharded(tmp_path):
    """The reference's known-answer cases with every setup cut over two ranks."""
    recs = run_ranks(tmp_path, 2, "host", "cases:" + ",".join(SHARDABLE_CASES))
    for rec in recs:
        bad = {k: v for k, v in rec.items() if v != "ok"}
        assert not bad, "\n".join("%s:\n%s" % kv for kv in bad.items())


def test_rccl_transport_one_rank(product_lib, tmp_path):
    """ncclCommInitRank / ncclAllGather resolved from the process's librccl, on the engine's stream."""
    ref = single(product_lib, 0, 3000, 12, 7)
    rec = run_ranks(tmp_path, 1, "rccl", "gen:0:3000:12:7", SETTINGS)[0]
    assert rec["status"] == "Solved" and rec["iter"] == ref["iter"]
    assert np.max(np.abs(np.array(rec["x"]) - ref["x"])) <= 1e-9 * max(1.0, np.max(np.abs(ref["x"])))
    assert rec["stats"][14] > 0


def update_sequence(oq, lib, oracle_lib, kind, n, per_row, seed, settings, comm=None, refused=False):
    """Host-array setup of a generated instance, a solve, new matrix values -- some diagonal entries of P by index, all of A
    in full [REF src/interface.jl:330-406] -- and a second solve.  Run by every rank of the sharded test and by the
    single-device reference."""
    import scipy.sparse as sp
    from test_gpu_parity import _data_to_scipy

    d = oracle_lib.oracle_generate(kind, n, per_row, seed)
    P, q, A, l, u = _data_to_scipy(d.contents)
    oracle_lib.oracle_data_free(d)
    P = sp.triu(P).tocsc(); A = A.tocsc()
    P.sort_indices(); A.sort_indices()
    m = oq.Model(lib)
    kw = dict(comm=comm) if comm is not None else {}
    oq.setup(m, P=P, q=q, A=A, l=l, u=u, **kw, **settings)
    r1 = oq.solve(m)
    cols = np.arange(0, P.shape[0], 7)
    diag_pos = P.indptr[cols + 1] - 1  # the diagonal is the last entry of a column of triu(P)
    assert np.all(P.indices[diag_pos] == cols)
    if refused:  # a compact row block: the update must come back as an error before anything was touched, the workspace stays usable
        try:
            oq.update_P_A(m, P.data[diag_pos] * 1.5 + 0.25, diag_pos.astype(np.int64), A.data * 1.1, None)
            raised, msg = False, ""
        except oq.OSQPError:
            raised, msg = True, lib.osqp_amd_last_error().decode()
        r2 = oq.solve(m)
        rec = {"raised": raised, "message": msg, "compact": oq.stats(m)[18], "status1": r1.info.status, "status2": r2.info.status,
               "x1": np.asarray(r1.x).tolist(), "x2": np.asarray(r2.x).tolist()}
        oq.clean(m)
        return rec
    oq.update_P_A(m, P.data[diag_pos] * 1.5 + 0.25, diag_pos.astype(np.int64), A.data * 1.1, None)
    r2 = oq.solve(m)
    rec = {"status1": r1.info.status, "iter1": int(r1.info.iter), "status2": r2.info.status, "iter2": int(r2.info.iter),
           "obj1": float(r1.info.obj_val), "obj2": float(r2.info.obj_val), "x2": np.asarray(r2.x).tolist(), "y2": np.asarray(r2.y).tolist()}
    oq.clean(m)
    return rec


@pytest.mark.parametrize("world,kind,n,per_row", [(2, 0, 3000, 12), (3, 0, 2501, 9)])
def test_update_matrices_on_a_sharded_workspace(product_lib, oracle_lib, tmp_path, world, kind, n, per_row):
    """Round 4: osqp_update_P / _A / _P_A on a row block (refused until round 3): every rank is handed the new values, each
    entry of its blocks picks its own by the caller's nnz index recorded at setup, then unscale / rescale / refresh as on one
    device.  Against the single-device workspace driven through the same sequence."""
    import osqp_jl_amd as oq

    ref = update_sequence(oq, product_lib, oracle_lib, kind, n, per_row, 7, SETTINGS)
    recs = run_ranks(tmp_path, world, "host", "update:%d:%d:%d:7" % (kind, n, per_row), SETTINGS)
    assert ref["status1"] == ref["status2"] == "Solved" and abs(ref["obj2"] - ref["obj1"]) > 1e-6 * abs(ref["obj1"])  # the update matters
    for rec in recs:
        assert rec["status1"] == rec["status2"] == "Solved"
        assert abs(rec["iter2"] - ref["iter2"]) <= 25
        assert abs(rec["obj2"] - ref["obj2"]) <= 1e-5 * max(1.0, abs(ref["obj2"]))
        assert np.max(np.abs(np.array(rec["x2"]) - np.array(ref["x2"]))) <= 1e-4 * max(1.0, np.max(np.abs(ref["x2"])))
        assert np.max(np.abs(np.array(rec["y2"]) - np.array(ref["y2"]))) <= 1e-4 * max(1.0, np.max(np.abs(ref["y2"])))
    for rec in recs[1:]:
        assert rec["x2"] == recs[0]["x2"] and rec["iter2"] == recs[0]["iter2"]


def test_update_matrices_is_refused_cleanly_on_a_compact_row_block(tmp_path, monkeypatch):
    """Advisor (round 4, high): a row block that went compact (OSQP_AMD_COMPACT_NNZ) has released its CSR values, so
    osqp_update_P / _A must refuse with exit flag 6 BEFORE unscaling anything -- not write through a released array and leave
    the other ranks hanging in the next collective.  The workspace stays usable: the next solve returns the first one's answer."""
    monkeypatch.setenv("OSQP_AMD_PANEL", "2")  # sliced copies whatever the size (they are what a compact block keeps)
    monkeypatch.setenv("OSQP_AMD_COMPACT_NNZ", "0")
    recs = run_ranks(tmp_path, 2, "host", "updaterefused:0:40000:24:7", SETTINGS)
    for rec in recs:
        assert rec["compact"] == 1.0 and rec["raised"] and "compact" in rec["message"], rec["message"]
        assert rec["status1"] == rec["status2"] == "Solved"
        assert np.max(np.abs(np.array(rec["x1"]) - np.array(rec["x2"]))) <= 1e-5 * max(1.0, np.max(np.abs(rec["x1"])))
