"""One rank of a row-sharded solve (launched by test_sharded_gpu.py / bench.py's checks; not a test module).

usage: _sharded_worker.py RANK WORLD PORT OUT_JSON TRANSPORT CASE
  TRANSPORT: host (gloo, any number of ranks on one GPU) | rccl
  CASE: gen:<kind>:<n>:<per_row>:<seed> | update:<kind>:<n>:<per_row>:<seed> (host-array setup, solve, osqp_update_P_A by
        index / in full, solve again) | cases:<name,name,...> (functions of tests/qp_cases.py, run with every
        setup routed through the communicator) | batch:<total>:<seed> (the sharded MPC batch, osqp_jl_amd.batch.MpcBatch)
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    rank, world, port, out, transport, case = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6]
    settings = json.loads(sys.argv[7]) if len(sys.argv) > 7 else {}
    import numpy as np
    import torch
    import torch.distributed as dist
    import osqp_jl_amd as oq
    from osqp_jl_amd import sharded

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % port, rank=rank, world_size=world)
    torch.cuda.set_device(0)
    lib = oq.load_library()
    comm = sharded.HostComm(lib=lib) if transport == "host" else sharded.RcclComm(lib=lib)
    if case.startswith("batch:"):
        from osqp_jl_amd import batch

        _, total, seed = case.split(":")
        b = batch.MpcBatch(lib, int(total), int(seed), device=0, comm=comm, **settings)
        packed = b.solve()
        again = b.solve(packed.clone())  # a second solve of the resident batch gives the same bits
        np.save("%s.%d.npy" % (out, rank), packed.numpy())
        with open("%s.%d" % (out, rank), "w") as f:
            json.dump({"first": b.first, "per": b.per, "same": bool(np.array_equal(again.numpy(), packed.numpy())),
                       "exchanges": 0}, f)
        b.close()
        comm.close()
        dist.barrier()
        dist.destroy_process_group()
        return
    if case.startswith("update:") or case.startswith("updaterefused:"):
        from test_sharded_gpu import update_sequence

        _, kind, n, per_row, seed = case.split(":")
        rec = update_sequence(oq, lib, oq.load_library(oq.ORACLE_LIB_PATH), int(kind), int(n), int(per_row), int(seed), settings, comm,
                              refused=case.startswith("updaterefused:"))
        with open("%s.%d" % (out, rank), "w") as f:
            json.dump(rec, f)
        comm.close()
        dist.barrier()
        dist.destroy_process_group()
        return
    m = oq.Model(lib)
    if case.startswith("gen:"):
        _, kind, n, per_row, seed = case.split(":")
        oq.setup_generated(m, int(kind), int(n), int(per_row), int(seed), comm=comm, **settings)
    else:
        import traceback
        import qp_cases

        class Routed:  # the package with `setup` keeping this rank's row block
            def __getattr__(self, name):
                return getattr(oq, name)

            @staticmethod
            def setup(model, *a, **k):
                return oq.setup(model, *a, comm=comm, **k)

        outcome = {}
        for name in case.split(":")[1].split(","):
            try:
                getattr(qp_cases, name)(Routed(), lib, "pcg")
                outcome[name] = "ok"
            except Exception:
                outcome[name] = traceback.format_exc()
        with open("%s.%d" % (out, rank), "w") as f:
            json.dump(outcome, f)
        comm.close()
        dist.barrier()
        dist.destroy_process_group()
        return
    res = oq.solve(m)
    rec = {"status": res.info.status, "iter": int(res.info.iter), "obj": float(res.info.obj_val),
           "pri_res": float(res.info.pri_res), "dua_res": float(res.info.dua_res), "rho_updates": int(res.info.rho_updates),
           "x": np.asarray(res.x).tolist(), "y": np.asarray(res.y).tolist(),
           "prim_inf_cert": np.asarray(res.prim_inf_cert).tolist(), "dual_inf_cert": np.asarray(res.dual_inf_cert).tolist(),
           "stats": oq.stats(m).tolist()}
    # a second solve after a vector update and a warm start exercises the sliced uploads
    if case.startswith("gen:") and "--second" in sys.argv:
        n_, m_ = oq.dimensions(m)
        rng = np.random.default_rng(5)
        oq.update_q(m, rng.standard_normal(n_))
        oq.warm_start(m, x=np.asarray(res.x), y=np.asarray(res.y))
        res2 = oq.solve(m)
        rec["second"] = {"status": res2.info.status, "iter": int(res2.info.iter), "obj": float(res2.info.obj_val),
                         "x": np.asarray(res2.x).tolist()}
    with open("%s.%d" % (out, rank), "w") as f:
        json.dump(rec, f)
    oq.clean(m)
    comm.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
