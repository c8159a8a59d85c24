#!/usr/bin/env python3
"""bench.py -- ADMM iterations/sec of the HIP engine on BASELINE.json's workload.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload rand-1e6|rand-1e5|lasso-5e5|control-1e6|mpc-batch]

A "step" is one ADMM iteration of the hot path (rhs build, KKT solve by the back-end the workload needs, fused
x/z/y update, residual evaluation every `check_termination` = 25 iterations) on a synthetic QP generated in HBM
before the timed region.  Protocol: W untimed warm-up iterations from the cold start, then exactly K timed ones
between barrier + synchronize, MAX over ranks, one JSON line from rank 0.

Launch.  One process per GPU.  Under `torch.distributed.run` (RANK / WORLD_SIZE in the environment) this process is
one rank.  Started plainly with `--gpus N`, N > 1, it SPAWNS the N ranks itself (one child per device, RCCL
rendezvous on 127.0.0.1, a free port) and relays rank 0's line -- `python bench.py --gpus 8` is an 8-GPU run.

At N > 1 the line carries three multi-GPU legs:
  * headline `value`: replicas -- every rank owns an independent QP (seed 1 + rank), no data-path collective, one
    final gather of the per-instance results (weak scaling);
  * `batch`: BASELINE.json config 5 -- 4096 MPC QPs (n = 100, m = 200) cut into contiguous blocks
    i -> floor(i / (4096 / N)), one workgroup per QP, ONE in-place ncclAllGather of the packed results issued by the
    library itself (strong scaling);
  * `sharded`: the 1-GPU run's QP (seed 1) cut into row blocks over the ranks, all-gather of every sparse product's
    input vector over RCCL (strong scaling of one solve);
  plus `collective_ranks_seen` (a sum of ones over the torch.distributed group) with its `collective_backend`, and in every leg
  `comm_ranks_seen` / `transport_ranks` (distinct rank ids gathered on the library's own communicator / what the transport
  itself reports: RCCL's ncclCommCount).  `batch` and `sharded` run after the replica leg in child
  processes of their own (their own process groups on the next ports, a time limit) so that neither an exception nor a
  hang in a transport can take the headline numbers with it.  `--mode sharded|batch` promotes a leg to the headline.

The line carries `roofline` for the dominant kernel (measured live with HIP events on the engine's stream; `traffic`
from FETCH_SIZE / WRITE_SIZE collected live by two rocprofv3 --pmc passes over a child of this script) and
`cpu_baseline` (the CPU oracle on the host cores).
"""
import argparse
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
LDS_PEAK_GBS = 150000.0   # aggregate ds_read_b64 rate, 256 B/clk/CU x 256 CUs x 2.4 GHz (MI355X_MICROARCH.md, LDS section)

WORKLOADS = {
    # name: (kind, n, per_row, linsys)
    "rand-1e6": (0, 1_000_000, 1000, "pcg"),
    "rand-1e5": (0, 100_000, 100, "pcg"),
    "rand-2e4": (0, 20_000, 20, "pcg"),
    "lasso-5e5": (1, 500_000, 0, "qdldl"),
    # long-horizon linear MPC as ONE banded QP (tests/qp_zoo.py `control`, nx = 12, nu = 6, T = 55 555: n = 1 000 002,
    # m = 1 666 674): nested dissection + supernodal triangular solves -- the trisolve path on a factor with real fill
    # (nnz(L) = 6.7e7, 519 pivot levels, 11 supernode levels).  Built on the host and handed to osqp_setup as CSC arrays.
    "control-1e6": ("control", 55_555, 0, "direct"),
    # a 2-D structure the direct back-end was NOT tuned on (tests/qp_zoo.py `grid2d`): P = 5-point Laplacian + 0.1 I on a
    # 1000 x 1000 grid, box constraints on every variable (A = I): n = m = 1e6.  Separators of ~1000 nodes: fronts of up to
    # ~2000 rows, far beyond one workgroup's LDS (csrc/mfront_big.hpp).  Built on the host, handed to osqp_setup as CSC arrays.
    "grid2d-1e6": ("grid", 1000, 0, "direct"),
    "grid2d-5e5": ("grid", 700, 0, "direct"),
}

SETTINGS = dict(verbose=False, eps_abs=1e-4, eps_rel=1e-4, check_termination=25, adaptive_rho_interval=50,
                polish=False, max_iter=4000)

BATCH_TOTAL = 4096
# the other BASELINE.json configurations, reported beside the headline line (`other_workloads`)
OTHER_WORKLOADS = ("rand-1e5", "lasso-5e5", "mpc-batch", "control-1e6", "grid2d-1e6")
CPU_RECORDS = {"rand-1e6": os.path.join(ROOT, "profiles", "r04_cpu_rand1e6.json")}
CPU_RECORD_WINDOW = (5, 20)  # W, K of the committed CPU record (the driver's protocol)


def _sha16(paths):
    import hashlib

    h = hashlib.sha256()
    for p in sorted(paths):
        h.update(os.path.basename(p).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def oracle_sha16():
    """The CPU oracle's sources: a CPU record measured with other sources is stale."""
    d = os.path.join(ROOT, "oracle")
    return _sha16([os.path.join(d, f) for f in os.listdir(d) if f.endswith((".c", ".h")) or f == "Makefile"])


def protocol_sha16(workload, warm, iters):
    """What a CPU record depends on besides the oracle sources: the workload tuple, the settings, the window."""
    import hashlib

    blob = json.dumps({"workload": WORKLOADS.get(workload, workload), "settings": SETTINGS, "warm": warm, "iters": iters}, sort_keys=True)
    return hashlib.sha256(blob.encode()).hexdigest()[:16]


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=25)
    ap.add_argument("--workload", default=os.environ.get("OSQP_AMD_BENCH_WORKLOAD", "rand-1e6"))
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--cpu-full", action="store_true",
                    help="re-measure the CPU oracle on rand-1e6 itself at the driver's window (W = 5, K = 20) in a child process and rewrite "
                         "the committed record (about 10 minutes and 46 GiB of host memory)")
    ap.add_argument("--mode", choices=["replicas", "sharded", "batch"], default=os.environ.get("OSQP_AMD_BENCH_MODE", "replicas"),
                    help="which multi-GPU leg is the headline value (all are measured when N > 1)")
    ap.add_argument("--no-sharded", action="store_true", help="skip the row-sharded leg")
    ap.add_argument("--no-batch", action="store_true", help="skip the sharded mpc-batch leg at N > 1")
    ap.add_argument("--traffic", choices=["live", "file", "off"], default=os.environ.get("OSQP_AMD_BENCH_TRAFFIC", "live"),
                    help="roofline.traffic: two live rocprofv3 --pmc passes (default), the committed profiles/pmc_traffic.json, or none")
    ap.add_argument("--child", choices=["sharded", "batch", "pmc"], default=None, help=argparse.SUPPRESS)  # internal legs
    ap.add_argument("--one-gpu-its", type=float, default=0.0, help=argparse.SUPPRESS)
    ap.add_argument("--spawn-check", action="store_true", help=argparse.SUPPRESS)  # launch logic only (no GPU): CPU test hook
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------------
# launch: spawn one rank per GPU when started without a launcher
# ----------------------------------------------------------------------------------------------------------------
def self_spawn(args):
    """`python bench.py --gpus N` from a clean environment: N children (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set
    as torch.distributed.run would), rank 0 on our stdout.  Returns the exit code."""
    n = args.gpus
    if not args.spawn_check and os.environ.get("OSQP_AMD_BENCH_ONE_DEVICE") != "1":
        import torch

        have = torch.cuda.device_count()
        if have < n:
            print(json.dumps({"error": f"--gpus {n} but only {have} HIP device(s) are visible"}))
            return 2
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "OSQP_AMD_BENCH_SPAWNED": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    return rc


def spawn_check(rank, world):
    """The launch path without a GPU: the ranks meet over gloo and count themselves."""
    import torch
    import torch.distributed as dist

    t = torch.ones(1, dtype=torch.float64)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"n_gpus": world, "ranks_seen": int(t.item()), "spawned": os.environ.get("OSQP_AMD_BENCH_SPAWNED") == "1"}))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------------------------
def main():
    args = parse_args()
    if "RANK" not in os.environ and args.gpus > 1 and args.child is None:
        sys.exit(self_spawn(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        args.gpus = world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.spawn_check:
        return spawn_check(rank, world)

    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist

    import osqp_jl_amd as oq

    # test hooks for a 1-GPU box: all ranks on device 0 over gloo (RCCL refuses two ranks on one GPU)
    if os.environ.get("OSQP_AMD_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    backend = os.environ.get("OSQP_AMD_BENCH_BACKEND", "nccl")
    if world > 1:
        dist.init_process_group(backend, rank=rank, world_size=world)

    lib = oq.load_library()  # HIP engine; hard error if missing
    assert lib.osqp_amd_set_device(local_rank) == 0
    ctx = dict(args=args, oq=oq, lib=lib, torch=torch, dist=dist, rank=rank, local_rank=local_rank, world=world, backend=backend)

    if args.child == "pmc":
        return pmc_child(ctx)
    if args.child == "sharded":
        kind, n, per_row, _ = WORKLOADS[args.workload]
        rec = sharded_leg(ctx, kind, n, per_row, args.one_gpu_its)
        if rank == 0:
            print("CHILD_RECORD " + json.dumps(rec))
            sys.stdout.flush()
        dist.destroy_process_group()
        return
    if args.child == "batch":
        rec = batch_leg(ctx, want_cpu=False)
        if rank == 0:
            print("CHILD_RECORD " + json.dumps(rec))
            sys.stdout.flush()
        if world > 1:
            dist.destroy_process_group()
        return

    if args.workload == "mpc-batch":
        rec = batch_leg(ctx, want_cpu=(not args.no_cpu and world == 1))
        if rank == 0:
            line = batch_line(args, world, rec)
            line["device"] = device_identity(torch, local_rank)
            print(json.dumps(line))
            sys.stdout.flush()
        if world > 1:
            dist.destroy_process_group()
        return
    replica_bench(ctx)


def control_problem(T, nx=12, nu=6, seed=6):
    """Linear MPC over T stages as one QP (the statement of tests/qp_zoo.py `control`): min sum x_t'Q x_t + u_t'R u_t
    s.t. x_{t+1} = A x_t + B u_t, x_0 given, box bounds; variables [x_0 .. x_T; u_0 .. u_{T-1}]."""
    import numpy as np
    import scipy.sparse as sp

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


def build_model(oq, lib, workload, seed, oracle=False):
    """A workspace of `lib` (the HIP engine, or the oracle in the CPU leg) for a workload: generated in place by the
    library's own generator, or -- control-1e6 -- built on the host and handed over through the reference entry point
    osqp_setup (CSC arrays), which is how a drop-in caller would.  Returns (model, n, setup seconds)."""
    kind, n, per_row, linsys = WORKLOADS[workload]
    model = oq.Model(lib)
    if kind == "control":
        # no hint: the library finds the long KKT graph itself and sends nested dissection first (csrc/direct.hip)
        prob = control_problem(n)
        t0 = time.time()
        oq.setup(model, linsys_solver="qdldl" if oracle else linsys, **prob, **SETTINGS)
        return model, int(prob["P"].shape[0]), time.time() - t0
    if kind == "grid":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import qp_zoo

        prob = qp_zoo.grid2d(n)
        t0 = time.time()
        oq.setup(model, linsys_solver="qdldl" if oracle else linsys, **prob, **SETTINGS)
        return model, int(prob["P"].shape[0]), time.time() - t0
    t0 = time.time()
    oq.setup_generated(model, kind, n, per_row, seed, linsys_solver=linsys, **SETTINGS)
    return model, n, time.time() - t0


# ----------------------------------------------------------------------------------------------------------------
# headline leg: one independent QP per rank
# ----------------------------------------------------------------------------------------------------------------
def start_cpu_full(args, rank, world):
    """The CPU oracle on the headline workload ITSELF, in this run: started as a child process before the GPU legs (one host
    thread and ~46 GiB against the GPU's work: different resources), collected by cpu_leg with a 20-minute cap.  Only when
    the host can afford it (>= 64 GiB available) and OSQP_AMD_BENCH_CPU_FULL is not 0; otherwise the committed record."""
    if rank != 0 or world != 1 or args.no_cpu or args.child is not None or args.workload not in CPU_RECORDS:
        return None
    if os.environ.get("OSQP_AMD_BENCH_CPU_FULL", "1") == "0" and not getattr(args, "cpu_full", False):
        return {"proc": None, "reason": "OSQP_AMD_BENCH_CPU_FULL=0"}
    try:
        avail = [int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0] / 1048576.0
    except Exception:
        avail = 0.0
    if avail < 64.0:
        return {"proc": None, "reason": "host has %.0f GiB available, the CPU run needs 46" % avail}
    W, K = CPU_RECORD_WINDOW
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    rec = os.path.join(ROOT, "gpurun_out", "cpu_full_record.json")
    try:
        os.remove(rec)
    except OSError:
        pass
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    cmd = [sys.executable, os.path.join(ROOT, "tools", "cpu_rand1e6.py"), "--phases", "cpu", "--warm", str(W), "--iters", str(K),
           "--out", os.path.join(ROOT, "gpurun_out", "cpu_full_run.json"), "--cpu-record", rec]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    return {"proc": subprocess.Popen(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE), "t0": time.time(), "record": rec}


def replica_bench(ctx):
    args, oq, lib, torch, dist, rank, world = (ctx[k] for k in ("args", "oq", "lib", "torch", "dist", "rank", "world"))
    ctx["cpu_full"] = start_cpu_full(args, rank, world)
    model, n, setup_s = build_model(oq, lib, args.workload, 1 + rank)
    ws = model.workspace

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up: W untimed ADMM iterations from the cold start
    if args.warmup > 0:
        assert lib.osqp_amd_iterate(ws, args.warmup) == 0
    st0 = oq.stats(model)
    barrier()
    t0 = time.perf_counter()
    assert lib.osqp_amd_iterate(ws, args.steps) == 0
    barrier()
    elapsed = time.perf_counter() - t0
    st1 = oq.stats(model)
    collective_ranks_seen = None
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        ones = torch.ones(1, dtype=torch.float64, device="cuda")
        dist.all_reduce(ones)  # how many ranks the collective library really connected
        collective_ranks_seen = int(ones.item())
    cg_per_admm = (st1[6] - st0[6]) / max(args.steps, 1)

    # time-to-eps: a full cold-start solve to eps_abs = eps_rel = 1e-4
    oq.update_settings(model, warm_start=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = oq.solve(model)
    torch.cuda.synchronize()
    solve_s = time.perf_counter() - t0
    st2 = oq.stats(model)

    # roofline of the dominant kernel, measured live with HIP events on the engine's stream
    st = st2
    nnzA, nnzPf = st[1], st[2]
    pmc_names = []
    if st[0] == 2:   # indirect back-end: SpMV y = A x
        variant = int(st[12]) if len(st) > 12 else 0
        kname = ["k_spmv<G> (CSR, y = A x)", "k_spmv_panel (LDS-staged x panels, y = A x)", "k_spmv_sell + k_panel_reduce (LDS-staged x panels, sliced-ELL tiles, y = A x)",
                 "k_spmv_sell + k_panel_reduce (wide x panels through L2, sliced-ELL tiles, y = A x)"][variant]
        pmc_names = ["k_spmv_sell", "k_panel_reduce"] if variant >= 2 else ["k_spmv"]
        which, abytes = 0, st[10]
    else:            # direct back-end: the kernels of one iteration (right-hand side | triangular solves | update, fused as the factor allows)
        kname, which = "direct ADMM iteration: rhs + forward | backward + update around the LDL' factor (k_direct2_fwd + k_direct2_bwd_update on a two-level factor)", 5
        abytes = st[11] + 8.0 * (6 * n + 12 * int(oq.dimensions(model)[1]))  # trisolve bytes + the vector updates
        pmc_names = ["k_direct2_fwd", "k_direct2_bwd_update"]  # the two-launch iteration of a two-level factor
        if int(st[19]) > 0:  # supernodal solves: right-hand side | level 0 | tree (forward, backward) | level 0 | update
            kname = ("direct ADMM iteration on a supernodal factor: k_direct_rhs | k_sn_level_w / _wf / _f forward (levels below the tree launch) | "
                     "k_sn_tree forward | k_sn_tree backward | k_sn_level_f / _wf / _w + k_sn_single_bwd backward | k_direct_update")
            pmc_names = {"sum": ["k_direct_rhs", "k_sn_level", "k_sn_single", "k_sn_tree", "k_direct_update"], "per": "k_direct_rhs"}
    ms = float(lib.osqp_amd_time_kernel(ws, which, 20))
    achieved = abytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    roofline = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "traffic_source": None, "ms_per_launch": round(ms, 4),
                "algorithmic_bytes_per_launch": abytes}
    # the whole step against the same peak: SURVEY.md 8d bytes of one ADMM iteration / measured time per step
    step_bytes = step_algorithmic_bytes(st, n, int(oq.dimensions(model)[1]), cg_per_admm)
    roofline["step"] = {"algorithmic_bytes_per_step": step_bytes, "achieved": round(step_bytes / (elapsed / args.steps) / 1e9, 1),
                        "frac": round(step_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                        "formula": "SURVEY.md 8d B_iter with the measured CG iterations per ADMM iteration"}

    # final gather of per-instance results (the only collective of the replica leg)
    summary = torch.tensor([float(res.info.iter), float(res.info.status_val), res.info.pri_res, res.info.dua_res,
                            res.info.obj_val, solve_s], dtype=torch.float64, device="cuda")
    if world > 1:
        gathered = [torch.zeros_like(summary) for _ in range(world)]
        dist.all_gather(gathered, summary)
        summaries = [g.cpu().tolist() for g in gathered]
    else:
        summaries = [summary.cpu().tolist()]

    its_per_s = args.steps * world / elapsed
    out = None
    if rank == 0:
        out = {
            "metric": "ADMM iterations/sec", "value": round(its_per_s, 3), "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": args.workload, "n": n, "m": int(oq.dimensions(model)[1]), "nnz_A": int(nnzA),
                       "nnz_P_full": int(nnzPf), "nnz_P_triu": int(st[3]), "backend": "pcg" if st[0] == 2 else "direct-ldl",
                       "eps_abs": 1e-4, "eps_rel": 1e-4, "check_termination": 25, "adaptive_rho_interval": 50,
                       "instances": world, "sharding": "one independent QP per GPU, final gather of the per-instance results"},
            "cg_iters_per_admm_iter": round(cg_per_admm, 3),
            "time_to_eps_s": round(solve_s, 4), "iters_to_eps": int(res.info.iter), "status": res.info.status,
            "cg_iters_to_eps": int(st2[6] - st1[6]),
            "pri_res": res.info.pri_res, "dua_res": res.info.dua_res, "rho_updates": int(res.info.rho_updates),
            "setup_s": round(setup_s, 3), "device_gb": round(st[9] / 1e9, 2), "device_peak_gb": round(st[20] / 1e9, 2),
            # the reference reports run_time = setup + solve [REF src/types.jl:92-96]: the same solve priced on that clock
            "run_time_s": round(setup_s + solve_s, 4), "iterations_per_s_incl_setup": round(res.info.iter / (setup_s + solve_s), 3),
            "per_rank": summaries, "collective_ranks_seen": collective_ranks_seen, "collective_backend": ctx["backend"] if world > 1 else None, "launch": "self-spawned" if os.environ.get("OSQP_AMD_BENCH_SPAWNED") == "1" else ("torchrun" if world > 1 else "single process"),
            "roofline": roofline, "cpu_baseline": None,
        }

    oq.clean(model)  # the replica's memory goes before any other leg is built
    del model

    if rank == 0:
        out["device"] = device_identity(torch, ctx["local_rank"])
    if rank == 0 and world == 1 and args.child is None and args.workload == "rand-1e6" and os.environ.get("OSQP_AMD_BENCH_OTHERS", "1") != "0":
        out["other_workloads"] = other_workloads(ctx)

    if rank == 0 and world == 1 and args.traffic != "off" and pmc_names:
        tr, src = (None, None)
        if args.traffic == "live":
            tr, src = live_traffic(args, pmc_names)
        if tr is None:
            tr, src = file_traffic(args)
        out["roofline"]["traffic"], out["roofline"]["traffic_source"] = tr, src

    if rank == 0 and not args.no_cpu and world == 1:  # the CPU leg belongs to the 1-GPU line only
        out["cpu_baseline"] = cpu_leg(oq, args, ctx.get("cpu_full"))
        out["cpu_baseline"]["all_cores_bandwidth_bound"] = all_cores_bound(out["roofline"]["step"]["algorithmic_bytes_per_step"])

    if world > 1:
        legs = []
        if not args.no_batch:
            legs.append(("batch", 1))
        if st[0] == 2 and not args.no_sharded:
            legs.append(("sharded", 2))
        for name, port_offset in legs:
            rec = run_child_leg(args, rank, name, port_offset, its_per_s / world)
            if rank == 0:
                out[name] = rec
        if rank == 0 and args.mode in ("sharded", "batch") and args.mode in out and "error" not in out[args.mode]:
            leg = out[args.mode]
            out.update({"value": leg["value"], "ms_per_step": leg["ms_per_step"], "scaling": "strong"})
            if args.mode == "sharded":
                out.update({"time_to_eps_s": leg["time_to_eps_s"], "iters_to_eps": leg["iters_to_eps"], "status": leg["status"]})
                out["config"]["instances"] = 1
            else:
                out["unit"] = leg["unit"]
            out["config"]["sharding"] = leg["sharding"]
            out["config"]["headline_leg"] = args.mode
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def kernel_roofline(oq, lib, model, n, st):
    """(kernel label, algorithmic bytes per launch, ms per launch) of the dominant kernel of the model's back-end, timed with HIP
    events on the engine's stream (osqp_amd_time_kernel)."""
    if st[0] == 2:
        which, abytes, label = 0, st[10], "spmv y = A x"
    else:
        which, label = 5, "direct ADMM iteration (rhs | triangular solves | update)"
        abytes = st[11] + 8.0 * (6 * n + 12 * int(oq.dimensions(model)[1]))
    ms = float(lib.osqp_amd_time_kernel(model.workspace, which, 20))
    return label, abytes, ms


def other_workload(ctx, name):
    """One of the other single-QP configurations of BASELINE.json through the same protocol as the headline (W untimed
    iterations from the cold start, K timed, then a full cold solve to eps): a compact record for the `other_workloads` block."""
    args, oq, lib, torch = (ctx[k] for k in ("args", "oq", "lib", "torch"))
    model, n, setup_s = build_model(oq, lib, name, 1)
    ws = model.workspace
    if args.warmup > 0:
        assert lib.osqp_amd_iterate(ws, args.warmup) == 0
    st0 = oq.stats(model)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    assert lib.osqp_amd_iterate(ws, args.steps) == 0
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    st1 = oq.stats(model)
    cg = (st1[6] - st0[6]) / max(args.steps, 1)
    oq.update_settings(model, warm_start=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = oq.solve(model)
    torch.cuda.synchronize()
    solve_s = time.perf_counter() - t0
    st = oq.stats(model)
    m = int(oq.dimensions(model)[1])
    label, abytes, ms = kernel_roofline(oq, lib, model, n, st)
    step_bytes = step_algorithmic_bytes(st, n, m, cg)
    rec = {"value": round(args.steps / elapsed, 3), "unit": "iterations/s", "ms_per_step": round(1e3 * elapsed / args.steps, 4),
           "time_to_eps_s": round(solve_s, 4), "iters_to_eps": int(res.info.iter), "status": res.info.status,
           "n": n, "m": m, "backend": "pcg" if st[0] == 2 else "direct-ldl", "cg_iters_per_admm_iter": round(cg, 3), "setup_s": round(setup_s, 3),
           "roofline": {"kernel": label, "frac": round(abytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms > 0 else None, "ms_per_launch": round(ms, 4),
                        "step_frac": round(step_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4)}}
    oq.clean(model)
    return rec


def other_workloads(ctx):
    """The BASELINE.json configurations next to the headline one, measured in the same process (about 20 s of GPU time):
    rand-1e5, lasso-5e5, control-1e6 (and grid2d-1e6, the direct back-end on a structure it was not tuned on) as single
    QPs, mpc-batch through the batched kernel.  A workload that fails reports its error and the others go on."""
    out = {}
    for name in OTHER_WORKLOADS:
        if name == ctx["args"].workload:
            continue
        try:
            if name == "mpc-batch":
                rec = batch_leg(ctx, want_cpu=False, traffic=False)
                out[name] = {k: rec[k] for k in ("value", "unit", "ms_per_step", "instances", "instances_per_s", "mean_iters_per_instance", "solved")}
                out[name]["roofline"] = {"bound": "lds", "frac": rec["roofline"]["frac"], "fp64_tflops": rec["roofline"]["fp64"]["achieved_tflops"]}
            else:
                out[name] = other_workload(ctx, name)
        except Exception as e:  # noqa: BLE001 -- the headline line must not depend on the side block
            out[name] = {"error": str(e)[:200]}
    return out


def device_identity(torch, local_rank):
    """Which device the bench ran on (the driver's own busy sampler has read 0 % on card0 in several rounds)."""
    try:
        p = torch.cuda.get_device_properties(local_rank)
        ident = {"index": local_rank, "name": p.name, "gcn_arch": getattr(p, "gcnArchName", None), "compute_units": p.multi_processor_count,
                 "memory_gb": round(p.total_memory / 1e9, 1)}
        for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"):
            if hasattr(p, k):
                ident[k] = int(getattr(p, k))
        if "pci_bus_id" in ident:
            ident["pci"] = "%04x:%02x:%02x.0" % (ident.get("pci_domain_id", 0), ident["pci_bus_id"], ident.get("pci_device_id", 0))
        for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
            if k in os.environ:
                ident[k] = os.environ[k]
        return ident
    except Exception as e:  # noqa: BLE001
        return {"index": local_rank, "error": str(e)[:100]}


def step_algorithmic_bytes(st, n, m, cg_per_admm, k=25):
    """algorithmic bytes of one ADMM iteration (V = 8, I = 4)."""
    nnzA, nnzPt = st[1], st[3]
    resid = (12.0 * (2.0 * nnzA + nnzPt) + 8.0 * (5 * n + 4 * m)) / k
    vec = 8.0 * (6 * n + 12 * m)
    if st[0] == 2:
        return cg_per_admm * (12.0 * (nnzPt + 2.0 * nnzA) + 80.0 * n) + vec + resid
    N = n + m
    return 2.0 * (12.0 * st[4] + 4.0 * (N + 1)) + 40.0 * N + vec + resid


# ----------------------------------------------------------------------------------------------------------------
# HBM traffic of the dominant kernel from the PMC counters, collected live (separate rocprofv3 passes; the guide's
# gfx950 correction: FETCH_SIZE counts 64 B per 128-B request of a wide streaming read -> doubled; both in KB)
# ----------------------------------------------------------------------------------------------------------------
def pmc_child(ctx):
    """Under rocprofv3 --pmc: build the workload and launch its dominant kernel a few times, nothing else."""
    args, oq, lib = ctx["args"], ctx["oq"], ctx["lib"]
    if args.workload == "mpc-batch":
        from osqp_jl_amd import batch
        b = batch.MpcBatch(lib, BATCH_TOTAL, 1, device=ctx["local_rank"], comm=None, **SETTINGS)
        packed = b.alloc()
        for _ in range(4):
            b.solve(packed)
        ctx["torch"].cuda.synchronize()
        print("PMC_CHILD_OK 0")
        b.close()
        return
    model, _, _ = build_model(oq, lib, args.workload, 1)
    which = 0 if oq.stats(model)[0] == 2 else 5  # the product of the indirect back-end / the iteration kernels of the direct one
    ms = float(lib.osqp_amd_time_kernel(model.workspace, which, 6))
    print("PMC_CHILD_OK %.4f" % ms)
    oq.clean(model)


def live_traffic(args, names, limit_s=240.0):
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, None
    per_counter = {}
    tmp = tempfile.mkdtemp(prefix="oq_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [rocprof, "--kernel-trace", "--pmc", counter, "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
                   "--child", "pmc", "--workload", args.workload, "--no-cpu"]
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
            p = subprocess.run(cmd, env=env, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=limit_s)
            if p.returncode != 0 or b"PMC_CHILD_OK" not in p.stdout:
                return None, None
            dbs = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
            if not dbs:
                return None, None
            per_counter[counter] = pmc_average(dbs[0], counter, names)
        if any(v is None for v in per_counter.values()):
            return None, None
        fetch = sum(per_counter["FETCH_SIZE"].values()) * 1024.0 * 2.0
        write = sum(per_counter["WRITE_SIZE"].values()) * 1024.0
        return fetch + write, ("live: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --child pmc`; "
                               "per launch = sum over %s of the per-dispatch averages; FETCH_SIZE x2 (gfx950), KB -> B; fetch %.4g B, write %.4g B"
                               % (" + ".join(names["sum"] if isinstance(names, dict) else names), fetch, write))
    except Exception:
        return None, None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_average(db, counter, names):
    """Per-dispatch average of `counter` for each kernel whose name contains one of `names` (the product launches only)."""
    import sqlite3

    c = sqlite3.connect(db)
    tables = {r[0] for r in c.execute("select name from sqlite_master where type in ('table', 'view')")}

    def pick(prefix):
        hits = sorted(t for t in tables if t == prefix or t.startswith(prefix + "_"))
        return hits[0] if hits else None

    ev, info, disp, sym = pick("rocpd_pmc_event"), pick("rocpd_info_pmc"), pick("rocpd_kernel_dispatch"), pick("rocpd_info_kernel_symbol")
    if not (ev and info and disp and sym):
        return None
    rows = c.execute(
        f"select s.kernel_name, avg(e.value) from {ev} e join {info} p on e.pmc_id = p.id "
        f"join {disp} d on e.event_id = d.event_id join {sym} s on d.kernel_id = s.id where p.name = ? group by s.kernel_name",
        (counter,)).fetchall()
    out = {}
    if isinstance(names, dict):
        # an iteration made of MANY launches of a few templates (the per-level supernodal solves): every dispatch of the
        # listed kernels summed, divided by the dispatches of the kernel that runs exactly once per iteration
        tot = c.execute(
            f"select s.kernel_name, sum(e.value), count(*) from {ev} e join {info} p on e.pmc_id = p.id "
            f"join {disp} d on e.event_id = d.event_id join {sym} s on d.kernel_id = s.id where p.name = ? group by s.kernel_name",
            (counter,)).fetchall()
        per = sum(n_ for k, _, n_ in tot if names["per"] in k)
        if not per:
            return None
        for nm in names["sum"]:
            vals = [v for k, v, _ in tot if nm in k]
            out[nm] = sum(vals) / per
        return out
    for nm in names:
        vals = [v for k, v in rows if nm in k]
        if not vals:
            return None
        out[nm] = max(vals)  # several instantiations of one template: the product's (largest) one
    return out


def file_traffic(args):
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(args.workload)
        if pm:
            return pm["fetch_bytes"] + pm["write_bytes"], "file: profiles/pmc_traffic.json (collected in an earlier run; the live collection was not available)"
    except Exception:
        pass
    return None, None


# ----------------------------------------------------------------------------------------------------------------
# legs in child processes (every rank starts one child; the children form their own process group)
# ----------------------------------------------------------------------------------------------------------------
def run_child_leg(args, rank, leg, port_offset, one_gpu_its, limit_s=300.0):
    """Whatever happens to the children -- exception, hang, crash -- this process keeps its own numbers."""
    # the launcher's agent hosts the rendezvous store of THIS group only: the children host their own on the next ports
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + port_offset)
    cmd = [sys.executable, os.path.abspath(__file__), "--child", leg, "--one-gpu-its", repr(float(one_gpu_its)), "--workload", args.workload,
           "--steps", str(args.steps), "--warmup", str(args.warmup), "--gpus", str(args.gpus), "--no-cpu"]
    try:
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=limit_s)
    except subprocess.TimeoutExpired:
        return {"error": "the %s leg did not finish within %.0f s" % (leg, limit_s)}
    if rank != 0:
        return {}
    for line in p.stdout.decode(errors="replace").splitlines():
        if line.startswith("CHILD_RECORD "):
            return json.loads(line[len("CHILD_RECORD "):])
    tail = (p.stderr.decode(errors="replace").strip().splitlines() or ["no output"])[-1]
    return {"error": "child exit code %d: %s" % (p.returncode, tail[:300])}


def make_comm(ctx):
    from osqp_jl_amd import sharded

    if ctx["world"] == 1:
        return None, "none (one rank)"
    # test hooks: the library's transport chosen apart from torch's backend, and the collective library by path
    transport = os.environ.get("OSQP_AMD_BENCH_TRANSPORT") or ("host" if ctx["backend"] == "gloo" else "rccl")
    if transport == "host":
        return sharded.HostComm(lib=ctx["lib"]), "host/gloo"
    return sharded.RcclComm(lib=ctx["lib"], librccl_path=os.environ.get("OSQP_AMD_BENCH_RCCL_LIB")), "rccl"


def comm_ranks_seen(ctx, comm):
    """All-gather of the rank ids on the library's own communicator: how many distinct ranks it reached."""
    torch = ctx["torch"]
    if comm is None:
        return 1
    buf = torch.full((comm.world,), -1.0, dtype=torch.float64, device="cuda")
    buf[comm.rank] = float(comm.rank)
    assert ctx["lib"].osqp_amd_comm_all_gather(comm.handle, buf.data_ptr(), 1) == 0
    return int(len(set(int(v) for v in buf.cpu().tolist() if v >= 0)))


def transport_ranks(comm):
    """What the transport itself says its communicator spans (RCCL: ncclCommCount); None without a communicator."""
    return None if comm is None else comm.info()[2]


def sharded_leg(ctx, kind, n, per_row, one_gpu_its):
    """One QP (seed 1, the 1-GPU run's instance) cut into row blocks over the ranks; same timing protocol."""
    args, oq, lib, torch, dist, world = (ctx[k] for k in ("args", "oq", "lib", "torch", "dist", "world"))
    comm, transport = make_comm(ctx)
    seen = comm_ranks_seen(ctx, comm)
    model = oq.Model(lib)
    t0 = time.time()
    oq.setup_generated(model, kind, n, per_row, 1, comm=comm, linsys_solver="pcg", **SETTINGS)
    setup_s = time.time() - t0
    ws = model.workspace

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        assert lib.osqp_amd_iterate(ws, args.warmup) == 0
    st0 = oq.stats(model)
    barrier()
    t0 = time.perf_counter()
    assert lib.osqp_amd_iterate(ws, args.steps) == 0
    barrier()
    elapsed = time.perf_counter() - t0
    st1 = oq.stats(model)
    tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())
    oq.update_settings(model, warm_start=0)
    barrier()
    t0 = time.perf_counter()
    res = oq.solve(model)
    torch.cuda.synchronize()
    solve_s = time.perf_counter() - t0
    ms_spmv = float(lib.osqp_amd_time_kernel(ws, 0, 20))
    ms_xchg = float(lib.osqp_amd_time_kernel(ws, 7, 20))
    st = oq.stats(model)
    value = args.steps / elapsed
    rec = {
        "value": round(value, 3), "unit": "iterations/s", "scaling": "strong", "ms_per_step": round(1e3 * elapsed / args.steps, 4),
        "speedup_vs_one_gpu_replica": round(value / one_gpu_its, 3) if one_gpu_its > 0 else None,
        "time_to_eps_s": round(solve_s, 4), "iters_to_eps": int(res.info.iter), "status": res.info.status,
        "pri_res": res.info.pri_res, "dua_res": res.info.dua_res,
        "cg_iters_per_admm_iter": round((st1[6] - st0[6]) / max(args.steps, 1), 3),
        "exchanges_per_admm_iter": round((st1[14] - st0[14]) / max(args.steps, 1), 2),
        "exchange_bytes_per_admm_iter": round((st1[15] - st0[15]) / max(args.steps, 1), 1),
        "setup_s": round(setup_s, 3), "device_gb_per_rank": round(st[9] / 1e9, 2), "transport": transport, "comm_ranks_seen": seen, "transport_ranks": transport_ranks(comm),
        "local_rows": [int(st[16]), int(st[17])],
        "spmv_local_ms": round(ms_spmv, 4),
        "spmv_local_GBs": round(st[10] / (ms_spmv * 1e-3) / 1e9, 1) if ms_spmv > 0 else None,
        "allgather_n_ms": round(ms_xchg, 4),
        "sharding": f"one QP, rows of A, A' and P cut into {world} blocks; all-gather of each product's input vector",
    }
    oq.clean(model)
    comm.close()
    return rec


def batch_leg(ctx, want_cpu, traffic=True):
    """BASELINE.json config 5: 4096 independent MPC QPs (n = 100, m = 200) cut into contiguous blocks over the ranks
    (instance i -> rank floor(i / (4096 / N))), resident in HBM; a step = one solve of the whole batch: every rank its
    block, one workgroup per QP, then ONE in-place all-gather of the packed [x | y | info] rows on the library's own
    communicator (osqp_amd_batch_mpc_solve = rows K11 + K12)."""
    args, oq, lib, torch, dist, rank, local_rank, world = (ctx[k] for k in ("args", "oq", "lib", "torch", "dist", "rank", "local_rank", "world"))
    from osqp_jl_amd import batch

    comm, transport = make_comm(ctx)
    seen = comm_ranks_seen(ctx, comm)
    b = batch.MpcBatch(lib, BATCH_TOTAL, 1, device=local_rank, comm=comm, **SETTINGS)
    packed = b.alloc()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    steps, warm = max(args.steps, 1), max(args.warmup, 1)
    for _ in range(min(warm, 5)):
        b.solve(packed)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        b.solve(packed)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    x, y, info = batch.split_packed(packed.numpy())  # the packed array is the library's own allocation (no torch on this path)
    iters = float(info[:, 0].sum())
    import hashlib
    packed_sha16 = hashlib.sha256(packed.numpy().tobytes()).hexdigest()[:16]  # the gathered batch, bit for bit (equal for every world size)
    # every rank must hold the whole batch after the gather: a checksum of checksums over the ranks
    check = torch.tensor([float(x.sum()), float(y.sum()), float(iters)], dtype=torch.float64, device="cuda")
    same = True
    if world > 1:
        allc = [torch.zeros_like(check) for _ in range(world)]
        dist.all_gather(allc, check)
        same = all(bool((c == allc[0]).all().item()) for c in allc)
    ms = 1e3 * elapsed / steps
    # on-chip roofline of the kernel (HBM sees each instance once): operand bytes per ADMM iteration of one instance =
    # the dense M^-1 b (8 n^2, registers) + the sparse products over A (A'(rho z - y), A x~: value + 2 indices + operand per entry)
    n_, m_, nnzA = batch.MPC_N, batch.MPC_M, 800
    lds_per_iter = 8.0 * n_ * n_ + 2 * nnzA * (8 + 2 + 2 + 8) + 8.0 * (6 * n_ + 10 * m_)
    lds_rate = lds_per_iter * iters * steps / elapsed / 1e9
    per_inst_bytes = 8.0 * (nnzA + n_ + n_ + 2 * m_) + 8.0 * (n_ + m_ + 4)
    rec = {
        "value": round(iters * steps / elapsed, 1), "unit": "iterations/s", "scaling": "strong", "ms_per_step": round(ms, 4),
        "instances": BATCH_TOTAL, "instances_per_rank": BATCH_TOTAL // world, "instances_per_s": round(BATCH_TOTAL * steps / elapsed, 1),
        "mean_iters_per_instance": round(iters / BATCH_TOTAL, 2), "solved": int((info[:, 1] == 1).sum()),
        "transport": transport, "comm_ranks_seen": seen, "transport_ranks": transport_ranks(comm), "every_rank_holds_the_whole_batch": bool(same),
        "packed_sha16": packed_sha16,
        "gather_bytes_per_rank": (BATCH_TOTAL // world) * 304 * 8 * (world - 1),
        "sharding": f"{BATCH_TOTAL // world} instances per GPU (contiguous blocks), one in-place all-gather of [x|y|info] at the end of each solve",
        "roofline": {"bound": "lds", "kernel": "k_batch_quad (one QP per four wavefronts, three per compute unit; the inverse in registers, "
                                                "everything else in LDS; OSQP_AMD_BATCH_QUAD=0: k_batch_solve, one QP per 512 threads)",
                     "achieved": round(lds_rate, 1), "peak": LDS_PEAK_GBS, "unit": "GB/s", "frac": round(lds_rate / LDS_PEAK_GBS / world, 5),
                     "traffic": None,
                     "lds_bytes_per_admm_iteration": lds_per_iter,
                     "hbm_GBs": round(per_inst_bytes * BATCH_TOTAL * steps / elapsed / 1e9, 3),
                     # the kernel is bound by fp64 instruction issue, so the same work as a fraction of the vector fp64 peak: per QP
                     # 2 n^2 + 4 nnz(A) + 20 (n + m) flops per ADMM iteration and 2 n^3 per Gauss-Jordan inversion
                     "fp64": {"achieved_tflops": round((iters * steps * (2.0 * n_ * n_ + 4.0 * nnzA + 20.0 * (n_ + m_)) +
                                                        float(info[:, 5].sum() + BATCH_TOTAL if info.shape[1] > 5 else 2 * BATCH_TOTAL) * steps * 2.0 * n_ ** 3) / elapsed / 1e12, 3),
                              "peak_tflops": 78.6, "note": "vector fp64 peak of MI355X; inversions counted as rho updates + 1 per instance (2 when the packed rows carry no count)"},
                     "note": "on-chip operand bytes per ADMM iteration (8 n^2 of M^-1 -- held in registers since round 2 -- plus values, 16-bit "
                             "indices and operands of the two sparse products and the vector updates from LDS) against the aggregate "
                             "ds_read peak, 256 CUs x 256 B/clk x 2.4 GHz per GPU; HBM sees each instance once (hbm_GBs); what is achieved "
                             "below the peak is the latency of the barrier-separated phases of an iteration, not bandwidth"},
    }
    if want_cpu:
        rec["cpu_baseline"] = batch_cpu_leg(oq, args)
    b.close()
    if traffic and rank == 0 and world == 1 and args.child is None and args.traffic == "live":  # HBM bytes of one launch, two --pmc passes over a child
        tr, src = live_traffic(args, ["k_batch_solve" if os.environ.get("OSQP_AMD_BATCH_QUAD") == "0" else "k_batch_quad"])
        rec["roofline"]["traffic"], rec["roofline"]["traffic_source"] = tr, src
    if comm is not None:
        comm.close()
    return rec


def batch_line(args, world, rec):
    out = {"metric": "ADMM iterations/sec", "value": rec["value"], "unit": "iterations/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": rec["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic",
           "config": {"workload": "mpc-batch", "instances": BATCH_TOTAL, "n": 100, "m": 200, "eps_abs": 1e-4, "eps_rel": 1e-4,
                      "sharding": rec["sharding"]}}
    for k in ("instances_per_s", "mean_iters_per_instance", "solved", "transport", "comm_ranks_seen", "transport_ranks", "every_rank_holds_the_whole_batch", "packed_sha16", "roofline"):
        out[k] = rec[k]
    out["cpu_baseline"] = rec.get("cpu_baseline")
    return out


def _batch_cpu_worker(task):
    """One host core: solve instances first, first + stride, ... with the CPU oracle until the deadline."""
    first, stride, seconds, opts = task
    import osqp_jl_amd as oq

    ora = oq.load_library(oq.ORACLE_LIB_PATH)
    t0 = time.perf_counter()
    k, its, i = 0, 0, first
    while time.perf_counter() - t0 < seconds:
        m = oq.Model(ora)
        oq.setup_generated(m, 2, 100, i % BATCH_TOTAL, 1, **opts)
        its += oq.solve(m).info.iter
        oq.clean(m)
        k += 1
        i += stride
    return k, its, time.perf_counter() - t0


def batch_cpu_leg(oq, args):
    """The oracle on the same instances: one thread (the reference library is single-threaded), and -- because the
    batch is embarrassingly parallel -- every host core at once, one process per core."""
    import multiprocessing as mp

    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    opts = dict(SETTINGS)
    k, its, spent = _batch_cpu_worker((0, 1, args.cpu_seconds / 2.0, opts))
    out = {"value": round(its / spent, 2), "unit": "iterations/s", "cores": 1, "host_cores": os.cpu_count(), "kind": "port",
           "sample": f"{k} instances of the batch solved one after another by the CPU oracle (setup + solve) in {spent:.1f} s",
           "instances_per_s": round(k / spent, 2)}
    cores = usable_cores()[0]
    try:
        with mp.get_context("fork").Pool(cores) as pool:
            res = pool.map(_batch_cpu_worker, [(r, cores, args.cpu_seconds / 2.0, opts) for r in range(cores)])
        kk, ii, tmax = sum(r[0] for r in res), sum(r[1] for r in res), max(r[2] for r in res)
        out["all_cores"] = {"cores": cores, "host_cores": os.cpu_count(), "cgroup_cpu_quota": usable_cores()[1],
                            "value": round(ii / tmax, 1), "instances_per_s": round(kk / tmax, 1),
                            "sample": f"{kk} instances over {cores} processes (one per host core) in {tmax:.1f} s"}
    except Exception as e:  # the single-thread figure stands on its own
        out["all_cores"] = {"error": str(e)[:200]}
    return out


# ----------------------------------------------------------------------------------------------------------------
# CPU baseline of the single-QP workloads
# ----------------------------------------------------------------------------------------------------------------
def cpu_leg(oq, args, cpu_full=None):
    """CPU oracle (oracle/, a port of the published algorithm; libosqp itself is not available in this image), 1 thread.
    Workloads the bounded leg can hold are timed live.  rand-1e6 (2.5e9 stored entries: 90 s of setup, ~16 s per ADMM
    iteration) is measured on the workload itself by tools/cpu_rand1e6.py at the driver's window; its record
    (profiles/r03_cpu_rand1e6.json) is `value`.  `--cpu-full` re-measures it in this run (a child process, ~10 minutes);
    otherwise the committed record is read and checked against hashes of the oracle sources and of the protocol
    (workload, settings, window) -- `"stale": true` when either no longer matches.  Beside it: the same family at
    n = 1e5 timed live (`live_sample`), and a context figure `all_cores_bandwidth_bound` -- the rate a perfectly
    threaded host implementation would be capped at by the host's measured memory bandwidth.  No scaled estimates."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    name = args.workload
    if name not in CPU_RECORDS:
        return cpu_live(oq, args, name)
    out = {"value": None, "unit": "iterations/s", "cores": 1, "host_cores": os.cpu_count(), "kind": "port", "config": name, "live": False}
    W, K = CPU_RECORD_WINDOW
    record_path = CPU_RECORDS[name]
    if cpu_full and cpu_full.get("proc") is not None:  # started before the GPU legs (start_cpu_full): collect it, 20-minute cap
        p = cpu_full["proc"]
        try:
            _, err = p.communicate(timeout=max(1.0, 1200.0 - (time.time() - cpu_full["t0"])))
            out["live"] = p.returncode == 0 and os.path.exists(cpu_full["record"])
            if not out["live"]:
                out["cpu_full_error"] = ((err or b"").decode(errors="replace").strip().splitlines() or ["exit %s" % p.returncode])[-1][:200]
        except subprocess.TimeoutExpired:
            p.kill()
            out["cpu_full_error"] = "the CPU run did not finish within 20 minutes of this run: committed record used"
        if out["live"]:
            record_path = cpu_full["record"]
            out["cpu_full_wall_s"] = round(time.time() - cpu_full["t0"], 1)
    elif cpu_full:
        out["cpu_full_skipped"] = cpu_full.get("reason")
    try:
        rec = json.load(open(record_path))
        out["value"] = rec.get("value")
        if rec.get("value") is None:
            out["reason"] = rec.get("reason", "not run")
        else:
            out["sample"] = (f"{name} itself: K = {rec['iters']} ADMM iterations of the CPU oracle after W = {rec.get('warm')} (PCG back-end, 1 thread of "
                             f"{rec.get('host_cores')} cores, {rec.get('cpu_model', 'host CPU')}) in {rec['seconds']} s after a {rec['setup_s']} s setup, "
                             f"{rec['cg_iters_per_admm_iter']} CG iterations per ADMM iteration, peak RSS {rec['peak_rss_gib']} GiB; measured by "
                             f"tools/cpu_rand1e6.py on the GPU box's host ({'IN THIS RUN, concurrently with the GPU legs' if out['live'] else 'record committed as profiles/' + os.path.basename(CPU_RECORDS[name])})")
            out["cg_iters_per_admm_iter"] = rec["cg_iters_per_admm_iter"]
            now = {"oracle_sha16": oracle_sha16(), "protocol_sha16": protocol_sha16(name, W, K)}
            out["stale"] = any(rec.get(k) != v for k, v in now.items())
            out["record_hashes"] = {k: rec.get(k) for k in now}
            if out["stale"]:
                out["current_hashes"] = now
    except Exception as e:
        out["reason"] = "not run: no committed record (%s)" % str(e)[:100]
    out["live_sample"] = cpu_live(oq, args, "rand-1e5")
    return out


def _stream_worker(task):
    """One host core: a = b + c over arrays far beyond the caches, `reps` times; returns (bytes moved, seconds)."""
    n, reps = task
    import numpy as np

    b, c = np.ones(n), np.ones(n)
    a = np.empty(n)
    np.add(b, c, out=a)  # touch
    t0 = time.perf_counter()
    for _ in range(reps):
        np.add(b, c, out=a)
    return 24.0 * n * reps, time.perf_counter() - t0


def usable_cores():
    """Host cores this process may really use: the scheduler affinity, capped by the cgroup's CPU quota (a container on a
    256-core host is often allowed a fraction of it -- the all-cores figures below are within that allowance and say so)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except Exception:
            pass
    if quota:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def host_stream_gbs(cores=None, n=8_000_000, reps=12):
    """STREAM-style add (2 reads + 1 write, 24 B per element, write-allocate not counted) on every host core at once, one
    process per core: the memory bandwidth a perfectly threaded host implementation could draw on."""
    import multiprocessing as mp

    cores = cores or usable_cores()[0]
    try:
        with mp.get_context("fork").Pool(cores) as pool:
            res = pool.map(_stream_worker, [(n, reps)] * cores)
        return sum(r[0] for r in res) / max(r[1] for r in res) / 1e9, cores
    except Exception:
        return None, cores


def all_cores_bound(step_bytes):
    """Context, not a measurement of any implementation: host memory bandwidth / SURVEY.md 8d bytes per ADMM iteration."""
    gbs, cores = host_stream_gbs()
    if not gbs:
        return None
    return {"host_stream_GBs": round(gbs, 1), "cores": cores, "host_cores": os.cpu_count(), "cgroup_cpu_quota": usable_cores()[1],
            "iterations_per_s": round(gbs * 1e9 / step_bytes, 3),
            "note": "an upper bound for ANY host implementation of this step (bandwidth-perfect, all cores): measured STREAM-add bandwidth of "
                    "the host / algorithmic bytes of one ADMM iteration at the measured CG count; nobody's code runs at it"}


def cpu_live(oq, args, sample):
    ora = oq.load_library(oq.ORACLE_LIB_PATH)
    m, _, setup_s = build_model(oq, ora, sample, 1, oracle=True)
    ws = m.workspace
    ora.osqp_amd_iterate(ws, 5)  # warm the caches
    st0 = oq.stats(m)
    iters, spent = 0, 0.0
    chunk = 5
    while spent < args.cpu_seconds and iters < 2000:
        t0 = time.perf_counter()
        ora.osqp_amd_iterate(ws, chunk)
        spent += time.perf_counter() - t0
        iters += chunk
    st = oq.stats(m)
    oq.clean(m)
    return {"value": round(iters / spent, 4), "unit": "iterations/s", "cores": 1, "host_cores": os.cpu_count(), "kind": "port", "config": sample,
            "live": True,
            "sample": f"{sample}: {iters} ADMM iterations of the CPU oracle ({'PCG' if st[0] == 2 else 'LDL'} back-end, 1 thread) "
                      f"in {spent:.1f} s after a {setup_s:.1f} s setup",
            "cg_iters_per_admm_iter": round((st[6] - st0[6]) / max(iters, 1), 3), "nnz_A": int(st[1])}


if __name__ == "__main__":
    main()
