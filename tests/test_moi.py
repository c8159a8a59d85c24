"""the MathOptInterface face restated (osqp.jl_amd/moi.py) and the reference's hand-written tests of it
[REF test/MOI_wrapper.jl:280-812] -- against the CPU oracle here, against the HIP engine through the C ABI on the GPU."""
import pytest

import moi_cases


@pytest.mark.parametrize("case", moi_cases.ALL, ids=lambda f: f.__name__)
def test_moi_face_on_oracle(oracle_lib, case):
    case(oracle_lib)


@pytest.mark.gpu
@pytest.mark.parametrize("case", moi_cases.ALL, ids=lambda f: f.__name__)
def test_moi_face_on_product(product_lib, case):
    case(product_lib)


import moi_conformance


@pytest.mark.parametrize("case", moi_conformance.ALL, ids=lambda f: f.__name__)
def test_moi_conformance_subset_on_oracle(oracle_lib, case):
    """The known-answer problems of MathOptInterface's generic suite that the reference runs through `MOI.Test.runtests`
    [REF test/MOI_wrapper.jl:59-93], restated by name (tests/moi_conformance.py)."""
    case(oracle_lib)


@pytest.mark.gpu
@pytest.mark.parametrize("case", moi_conformance.ALL, ids=lambda f: f.__name__)
def test_moi_conformance_subset_on_product(product_lib, case):
    case(product_lib)
