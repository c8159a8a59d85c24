"""bench.py with more than one rank on the ONE GPU of the test box (every rank on device 0, torch.distributed over gloo):
what the driver's 2 / 4 / 8-GPU runs execute, minus the second device.  No scaling number comes out of this -- the ranks
share a device -- it is the launch path, the collectives and the error handling that are exercised:

  * eight ranks: the replica leg sees 8 ranks in its collective, the batched leg (BASELINE.json config 5: 4096 MPC QPs cut
    into 8 contiguous blocks, one all-gather) leaves every rank with the whole batch, bit for bit what one rank computes;
  * a collective library that cannot connect its ranks (tests/stub_rccl.c with STUB_RCCL_FAIL_INIT): the batched and the
    row-sharded legs report `{"error": ...}` and the replica line is printed all the same.
"""
import json
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "LOCAL_WORLD_SIZE", "OSQP_AMD_BENCH_SPAWNED")
           and not k.startswith("TORCHELASTIC_")}
    env.update({"OSQP_AMD_BENCH_ONE_DEVICE": "1", "OSQP_AMD_BENCH_BACKEND": "gloo", "OSQP_AMD_BENCH_CPU_FULL": "0"})
    env.update(extra)
    return env


def _bench(argv, extra_env, timeout=900):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=_env(extra_env), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=timeout)
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert p.returncode == 0 and lines, p.stderr.decode()[-3000:] + p.stdout.decode()[-1000:]
    return json.loads(lines[-1])


def test_eight_ranks_on_one_device_gather_the_single_rank_batch(product_lib):
    one = _bench(["--workload", "mpc-batch", "--steps", "2", "--warmup", "1", "--no-cpu", "--traffic", "off"], {})
    assert one["solved"] == 4096 and one["n_gpus"] == 1
    rec = _bench(["--gpus", "8", "--workload", "rand-2e4", "--steps", "5", "--warmup", "2", "--no-cpu", "--traffic", "off"], {})
    assert rec["n_gpus"] == 8 and rec["collective_ranks_seen"] == 8 and len(rec["per_rank"]) == 8
    assert rec["launch"] == "self-spawned" and rec["scaling"] == "weak"
    b = rec["batch"]
    assert "error" not in b, b
    assert b["comm_ranks_seen"] == 8 and b["instances_per_rank"] == 512 and b["solved"] == 4096 and b["every_rank_holds_the_whole_batch"]
    assert b["packed_sha16"] == one["packed_sha16"]  # sharding does not change a bit of any instance's result
    s = rec["sharded"]
    assert "error" not in s and s["comm_ranks_seen"] == 8 and s["status"] == "Solved", s


def test_a_collective_library_that_cannot_connect_leaves_the_replica_line(product_lib):
    stub = os.path.join(tempfile.gettempdir(), "libstub_rccl_fail_%d.so" % os.getpid())
    subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-o", stub, os.path.join(ROOT, "tests", "stub_rccl.c")])
    try:
        rec = _bench(["--gpus", "2", "--workload", "rand-2e4", "--steps", "5", "--warmup", "2", "--no-cpu", "--traffic", "off"],
                     {"OSQP_AMD_BENCH_TRANSPORT": "rccl", "OSQP_AMD_BENCH_RCCL_LIB": stub, "STUB_RCCL_FAIL_INIT": "1"})
    finally:
        os.remove(stub)
    assert rec["n_gpus"] == 2 and rec["collective_ranks_seen"] == 2 and rec["value"] > 0 and rec["status"] == "Solved"
    assert "error" in rec["batch"] and "error" in rec["sharded"], (rec["batch"], rec["sharded"])
    assert "osqp_amd_comm_create_rccl" in rec["batch"]["error"] or "rccl" in rec["batch"]["error"].lower(), rec["batch"]
