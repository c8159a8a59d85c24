#!/usr/bin/env python3
"""bench.py -- ADMM iterations/sec of the HIP engine on BASELINE.json's workload.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload rand-1e6|rand-1e5|lasso-5e5]

A "step" is one ADMM iteration of the hot path (rhs build, KKT solve by the
back-end the workload needs, fused x/z/y update, residual evaluation every
`check_termination`=25 iterations) on a synthetic QP generated in HBM before the
timed region.  Each rank (one process per GPU) owns an independent QP instance
(seed = 1 + rank): the path shards over instances with no data-path collective;
the only exchange is the final RCCL gather of per-instance results (weak scaling).

With more than one rank and an indirect-back-end workload the line also carries
`sharded`: the SAME QP as the 1-GPU run (seed 1) cut into row blocks over the
ranks (all-gather of the product inputs over RCCL), timed
the same way -- strong scaling of one solve.  `--mode sharded` makes that the
headline `value` instead of the replicas.  The sharded leg runs after the
replica leg in child processes of its own (one per rank, their own process
group on MASTER_PORT + 1, a time limit), so that neither an exception nor a
hang nor a crash in the transport can take the replica numbers with it.

The JSON line carries `roofline` for the dominant kernel (CSR SpMV y = A x,
measured live with HIP events on the engine's stream) and `cpu_baseline` (the
CPU oracle timed on rank 0's host core on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling

WORKLOADS = {
    # name: (kind, n, per_row, linsys)
    "rand-1e6": (0, 1_000_000, 1000, "pcg"),
    "rand-1e5": (0, 100_000, 100, "pcg"),
    "rand-2e4": (0, 20_000, 20, "pcg"),
    "lasso-5e5": (1, 500_000, 0, "qdldl"),
}

SETTINGS = dict(verbose=False, eps_abs=1e-4, eps_rel=1e-4, check_termination=25, adaptive_rho_interval=50,
                polish=False, max_iter=4000)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=25)
    ap.add_argument("--workload", default=os.environ.get("OSQP_AMD_BENCH_WORKLOAD", "rand-1e6"))
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--mode", choices=["replicas", "sharded"], default=os.environ.get("OSQP_AMD_BENCH_MODE", "replicas"),
                    help="which multi-GPU leg is the headline value (both are measured when N > 1)")
    ap.add_argument("--no-sharded", action="store_true", help="skip the row-sharded leg")
    ap.add_argument("--sharded-child", type=float, default=None, help=argparse.SUPPRESS)  # internal: run only the sharded leg
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import osqp_jl_amd as oq

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    # test hooks for a 1-GPU box: all ranks on device 0 over gloo (RCCL refuses two ranks on one GPU)
    if os.environ.get("OSQP_AMD_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("OSQP_AMD_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)

    lib = oq.load_library()  # HIP engine; hard error if missing
    assert lib.osqp_amd_set_device(local_rank) == 0

    if args.sharded_child is not None:  # child of a multi-rank run: the row-sharded leg alone, its record on rank 0's stdout
        kind, n, per_row, _ = WORKLOADS[args.workload]
        rec = sharded_leg(args, oq, lib, torch, dist, rank, world, kind, n, per_row, args.sharded_child)
        if rank == 0:
            print("SHARDED_RECORD " + json.dumps(rec))
            sys.stdout.flush()
        dist.destroy_process_group()
        return

    if args.workload == "mpc-batch":
        return bench_batch(args, oq, lib, torch, dist, rank, local_rank, world)

    kind, n, per_row, linsys = WORKLOADS[args.workload]
    model = oq.Model(lib)
    t0 = time.time()
    oq.setup_generated(model, kind, n, per_row, 1 + rank, linsys_solver=linsys, **SETTINGS)
    setup_s = time.time() - t0
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
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    cg_per_admm = (st1[6] - st0[6]) / max(args.steps, 1)

    # time-to-eps: a full cold-start solve to eps_abs = eps_rel = 1e-4
    oq.update_settings(model, warm_start=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = oq.solve(model)
    torch.cuda.synchronize()
    solve_s = time.perf_counter() - t0

    # roofline of the dominant kernel, measured live with HIP events on the engine's stream
    st = oq.stats(model)
    nnzA, nnzPf = st[1], st[2]
    if st[0] == 2:   # indirect back-end: CSR SpMV y = A x
        variant = int(st[12]) if len(st) > 12 else 0
        kname = ["k_spmv<G> (CSR, y = A x)", "k_spmv_panel (LDS-staged x panels, y = A x)", "k_spmv_sell (LDS-staged x panels, sliced-ELL tiles, y = A x)",
                 "k_spmv_sell (wide x panels through L2, sliced-ELL tiles, y = A x)"][variant]
        which, abytes = 0, st[10]
    else:            # direct back-end: forward+backward triangular solve
        kname, which, abytes = "sptrsv forward+backward", 3, st[11]
    ms = float(lib.osqp_amd_time_kernel(ws, which, 20))
    achieved = abytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    traffic = None  # HBM bytes per launch from PMC counters: collected in separate rocprofv3 passes, committed under profiles/
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(args.workload)
        if pm and st[0] == 2 and int(st[12]) == 2:
            traffic = pm["fetch_bytes"] + pm["write_bytes"]
    except Exception:
        pass
    roofline = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "ms_per_launch": round(ms, 4),
                "algorithmic_bytes_per_launch": abytes}

    # final gather of per-instance results over RCCL (the only collective of the path)
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
                       "nnz_P_full": int(nnzPf), "backend": "pcg" if st[0] == 2 else "direct-ldl",
                       "eps_abs": 1e-4, "eps_rel": 1e-4, "check_termination": 25, "adaptive_rho_interval": 50,
                       "instances": world, "sharding": "one independent QP per GPU, final RCCL all_gather of results"},
            "cg_iters_per_admm_iter": round(cg_per_admm, 3),
            "time_to_eps_s": round(solve_s, 4), "iters_to_eps": int(res.info.iter), "status": res.info.status,
            "pri_res": res.info.pri_res, "dua_res": res.info.dua_res, "rho_updates": int(res.info.rho_updates),
            "setup_s": round(setup_s, 3), "device_gb": round(st[9] / 1e9, 2),
            "per_rank": summaries,
            "roofline": roofline, "cpu_baseline": None,
        }

    def emit():
        if rank == 0:
            print(json.dumps(out))
            sys.stdout.flush()

    if rank == 0 and not args.no_cpu and world == 1:  # the CPU leg belongs to the 1-GPU line only
        out["cpu_baseline"] = cpu_leg(oq, args)

    if world > 1 and st[0] == 2 and not args.no_sharded:
        oq.clean(model)  # the replica's 80 GB go before the sharded copy is built
        sh = run_sharded_child(args, rank, its_per_s / world)
        if rank == 0:
            out["sharded"] = sh
            if args.mode == "sharded" and "error" not in sh:
                out.update({"value": sh["value"], "ms_per_step": sh["ms_per_step"], "scaling": "strong",
                            "time_to_eps_s": sh["time_to_eps_s"], "iters_to_eps": sh["iters_to_eps"], "status": sh["status"]})
                out["config"]["sharding"] = sh["sharding"]
                out["config"]["instances"] = 1
    emit()
    if world > 1:
        dist.destroy_process_group()


def run_sharded_child(args, rank, one_gpu_its, limit_s=240.0):
    """Every rank starts one child (same script, --sharded-child) that joins a process group of the children on
    MASTER_PORT + 1; the record comes back on rank 0's child's stdout.  Whatever happens to the children -- exception,
    hang, crash -- this process keeps its own numbers."""
    import subprocess

    # the launcher's agent hosts the rendezvous store of THIS group only: the children host their own on the next port
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 1)
    cmd = [sys.executable, os.path.abspath(__file__), "--sharded-child", repr(float(one_gpu_its)), "--workload", args.workload,
           "--steps", str(args.steps), "--warmup", str(args.warmup), "--gpus", str(args.gpus), "--no-cpu"]
    try:
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=limit_s)
    except subprocess.TimeoutExpired:
        return {"error": "the row-sharded leg did not finish within %.0f s" % limit_s}
    if rank != 0:
        return {}
    for line in p.stdout.decode(errors="replace").splitlines():
        if line.startswith("SHARDED_RECORD "):
            return json.loads(line[len("SHARDED_RECORD "):])
    tail = (p.stderr.decode(errors="replace").strip().splitlines() or ["no output"])[-1]
    return {"error": "child exit code %d: %s" % (p.returncode, tail[:300])}


def sharded_leg(args, oq, lib, torch, dist, rank, world, kind, n, per_row, one_gpu_its):
    """One QP (seed 1, the 1-GPU run's instance) cut into row blocks over the ranks; same timing protocol."""
    from osqp_jl_amd import sharded

    host = os.environ.get("OSQP_AMD_BENCH_BACKEND", "nccl") == "gloo"
    comm = sharded.HostComm(lib=lib) if host else sharded.RcclComm(lib=lib)
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
        "setup_s": round(setup_s, 3), "device_gb_per_rank": round(st[9] / 1e9, 2), "transport": "host/gloo" if host else "rccl",
        "local_rows": [int(st[16]), int(st[17])],
        "spmv_local_ms": round(ms_spmv, 4),
        "spmv_local_GBs": round(st[10] / (ms_spmv * 1e-3) / 1e9, 1) if ms_spmv > 0 else None,
        "allgather_n_ms": round(ms_xchg, 4),
        "sharding": f"one QP, rows of A, A' and P cut into {world} blocks; all-gather of each product's input vector",
    }
    oq.clean(model)
    comm.close()
    return rec


def bench_batch(args, oq, lib, torch, dist, rank, local_rank, world):
    """BASELINE.json config 5: 4096 independent MPC QPs (n=100, m=200) sharded over the
    ranks, one workgroup per QP, one RCCL all-gather of the packed results at the end.
    A step = one solve of the whole batch (every rank solves its block)."""
    from osqp_jl_amd import batch

    total = 4096
    opts = dict(SETTINGS)
    solver = batch.device_mpc_solver(lib, local_rank, **opts)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):
        x, y, info = batch.solve_mpc_sharded(solver, total, 1, rank=rank, world=world, dist=dist if world > 1 else None)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x, y, info = batch.solve_mpc_sharded(solver, total, 1, rank=rank, world=world, dist=dist if world > 1 else None)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    info = info.cpu().numpy()
    iters = float(info[:, 0].sum())
    cpu = None
    if rank == 0 and not args.no_cpu and world == 1:
        ora = oq.load_library(oq.ORACLE_LIB_PATH)
        t0 = time.perf_counter()
        k, its = 0, 0
        while time.perf_counter() - t0 < args.cpu_seconds:
            m = oq.Model(ora)
            oq.setup_generated(m, 2, 100, k, 1, **opts)
            its += oq.solve(m).info.iter
            k += 1
        spent = time.perf_counter() - t0
        cpu = {"value": round(its / spent, 2), "unit": "iterations/s", "cores": 1, "host_cores": os.cpu_count(), "kind": "port",
               "sample": f"{k} of the 4096 instances solved one after another by the CPU oracle (setup + solve) in {spent:.1f} s",
               "instances_per_s": round(k / spent, 2)}
    if rank == 0:
        # LDS-resident kernel: HBM sees each instance's data once (8 B x (nnzA + n + n + 2m) in, 8 B x (n + m + 4) out)
        per_inst_bytes = 8.0 * (800 + 100 + 100 + 400) + 8.0 * (100 + 200 + 4)
        out = {
            "metric": "ADMM iterations/sec", "value": round(iters * args.steps / elapsed, 1), "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "mpc-batch", "instances": total, "n": 100, "m": 200, "eps_abs": 1e-4, "eps_rel": 1e-4,
                       "sharding": f"{total // world} instances per GPU, one RCCL all_gather of [x|y|info] at the end"},
            "instances_per_s": round(total * args.steps / elapsed, 1), "mean_iters_per_instance": round(iters / total, 2),
            "solved": int((info[:, 1] == 1).sum()),
            "roofline": {"bound": "hbm", "kernel": "k_batch_solve (LDS-resident; HBM traffic is load + store of each instance only)",
                         "achieved": round(per_inst_bytes * total * args.steps / elapsed / 1e9, 3), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(per_inst_bytes * total * args.steps / elapsed / 1e9 / HBM_PEAK_GBS, 6),
                         "traffic": None, "note": "latency/LDS-bound by design: ~126 KB of LDS per instance, one 512-thread workgroup per CU"},
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def cpu_leg(oq, args):
    """CPU oracle (oracle/, a port of the published algorithm; libosqp itself is not
    available in this image) on a bounded sample: the rand-1e5 member of the same
    family when the workload is rand-1e6 (whose 2.5e9 non-zeros do not fit a
    bounded CPU run), the workload itself otherwise."""
    import subprocess

    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    ora = oq.load_library(oq.ORACLE_LIB_PATH)
    name = args.workload
    sample = name
    if name == "rand-1e6":
        sample = "rand-1e5"
    kind, n, per_row, linsys = WORKLOADS[sample]
    m = oq.Model(ora)
    t0 = time.perf_counter()
    oq.setup_generated(m, kind, n, per_row, 1, linsys_solver=linsys, **SETTINGS)
    setup_s = time.perf_counter() - t0
    ws = m.workspace
    ora.osqp_amd_iterate(ws, 5)  # warm the caches
    iters, spent = 0, 0.0
    chunk = 5
    while spent < args.cpu_seconds and iters < 2000:
        t0 = time.perf_counter()
        ora.osqp_amd_iterate(ws, chunk)
        spent += time.perf_counter() - t0
        iters += chunk
    st = oq.stats(m)
    v = iters / spent
    out = {"value": round(v, 4), "unit": "iterations/s", "cores": 1, "host_cores": os.cpu_count(), "kind": "port",
           "sample": f"{sample}: {iters} ADMM iterations of the CPU oracle ({'PCG' if st[0] == 2 else 'LDL'} back-end, 1 thread) "
                     f"in {spent:.1f} s after a {setup_s:.1f} s setup",
           "nnz_A": int(st[1])}
    if sample != name:
        kindw, nw, kw, _ = WORKLOADS[name]
        scale = (nw * kw) / (n * per_row)
        out["value_scaled_to_workload"] = round(v / scale, 5)
        out["scaling_note"] = f"per-iteration work of {name} is {scale:.0f}x that of {sample} (nnz ratio); value_scaled_to_workload = value / {scale:.0f}"
    return out


if __name__ == "__main__":
    main()
