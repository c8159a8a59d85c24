"""The reference's ccall sequence performed by a plain-C program (tests/c_harness.c: dlopen, hand-built csc /
OSQPData / OSQPSettings, osqp_setup, osqp_solve, results read through the raw byte offsets of the Julia mirrors
[REF src/interface.jl:132-210, src/types.jl:173-217]) -- independent of the Python ctypes mirror.
CPU: against the oracle (validates the harness and the oracle's ABI).  GPU: against libosqp_amd.so."""
import json
import os
import subprocess

import pytest

import osqp_jl_amd as oq

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "c_harness")
    # -std=c11 -pedantic: the header must be plain C (its _Static_assert layout checks are compiled here)
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-Werror", "-I", ROOT, os.path.join(ROOT, "tests", "c_harness.c"),
                           "-o", exe, "-ldl", "-lm"])
    return exe


def _run(exe, lib_path):
    p = subprocess.run([exe, lib_path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    steps = [json.loads(l) for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert p.returncode == 0, p.stderr.decode()[-2000:] + p.stdout.decode()[-2000:]
    assert steps and steps[-1] == {"step": "done", "failures": 0}
    return {s["step"]: s for s in steps}


def test_c_harness_against_oracle(tmp_path, oracle_lib):
    steps = _run(_build(tmp_path), oq.ORACLE_LIB_PATH)
    assert steps["G1"]["status_val"] == 1 and steps["infeasible"]["status_val"] == -3


@pytest.mark.gpu
def test_c_harness_against_product(tmp_path, product_lib):
    steps = _run(_build(tmp_path), oq.PRODUCT_LIB_PATH)
    assert steps["G1"]["status"] == "solved" and abs(steps["G1"]["obj"] - 20.0) < 1e-5
    assert abs(steps["G2"]["obj"] - 100.0) < 1e-5 and steps["infeasible"]["status_val"] == -3
