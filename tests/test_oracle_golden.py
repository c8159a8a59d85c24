"""Pins the CPU oracle (oracle/, test infrastructure) against every known answer
the reference's own tests hold for the hot path.
Runs on CPU.  The same cases run against the HIP engine in test_gpu_parity.py."""
import pytest

import osqp_jl_amd as oq
import qp_cases


@pytest.mark.parametrize("linsys", ["qdldl", "pcg"])
@pytest.mark.parametrize("case", qp_cases.ALL_CASES, ids=lambda f: f.__name__)
def test_oracle_case(oracle_lib, case, linsys):
    if linsys == "pcg" and case.__name__ in qp_cases.DIRECT_ONLY:
        pytest.skip("inertia is only checked by a factorisation")
    case(oq, oracle_lib, linsys)
