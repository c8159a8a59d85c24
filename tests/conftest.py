import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    """The CPU oracle (test infrastructure), built on demand with gcc."""
    import osqp_jl_amd as oq

    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return oq.load_library(oq.ORACLE_LIB_PATH)


@pytest.fixture(scope="session")
def product_lib():
    """The HIP engine through its C ABI; never falls back to anything else."""
    import osqp_jl_amd as oq

    return oq.load_library()
