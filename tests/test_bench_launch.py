"""bench.py's launch path on CPU: `python bench.py --gpus N` from a clean environment must spawn N ranks itself
(one per device; here the ranks only meet over gloo and count themselves -- `--spawn-check` touches no GPU), and under
an external launcher (RANK / WORLD_SIZE set) it must be exactly one rank of that launch."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "LOCAL_WORLD_SIZE", "OSQP_AMD_BENCH_SPAWNED")
           and not k.startswith("TORCHELASTIC_")}
    return env


def _last_json(out):
    lines = [l for l in out.decode().splitlines() if l.startswith("{")]
    assert lines, out.decode()[-2000:]
    return json.loads(lines[-1])


def test_gpus_n_spawns_n_ranks():
    for n in (2, 3):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--spawn-check"], env=_clean_env(),
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        rec = _last_json(p.stdout)
        assert rec == {"n_gpus": n, "ranks_seen": n, "spawned": True}


def test_under_a_launcher_it_is_one_rank():
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--spawn-check"], env=_clean_env(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert _last_json(p.stdout) == {"n_gpus": 2, "ranks_seen": 2, "spawned": False}


def test_single_process_default():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--spawn-check"], env=_clean_env(), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert _last_json(p.stdout) == {"n_gpus": 1, "ranks_seen": 1, "spawned": False}
