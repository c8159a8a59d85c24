"""CPU-side checks of the boundary: the product library loads and exports every
symbol include/osqp_amd.h declares (no compute calls -- there is no GPU here),
and the struct layouts match the offsets the Julia side reads [REF src/types.jl]."""
import ctypes as C
import os
import re
import subprocess

import pytest

import osqp_jl_amd as oq
from osqp_jl_amd import types as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "osqp_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(osqp_[a-z_A-Z0-9]+)\s*\(", text)))


def test_header_declares_the_30_reference_symbols():
    names = _declared_symbols()
    for sym in T.ABI_SYMBOLS:
        assert sym in names
    assert len(T.ABI_SYMBOLS) == 30


def test_product_library_exports_every_declared_symbol():
    if not os.path.exists(oq.PRODUCT_LIB_PATH):
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "osqp.jl_amd", "csrc")])
    lib = C.CDLL(oq.PRODUCT_LIB_PATH, mode=os.RTLD_LOCAL | os.RTLD_NOW)
    for sym in _declared_symbols():
        assert hasattr(lib, sym), sym
    lib.osqp_version.restype = C.c_char_p
    assert lib.osqp_version() == b"0.6.2"
    s = T.Settings()
    lib.osqp_set_default_settings(C.byref(s))
    assert (s.rho, s.sigma, s.scaling, s.max_iter, s.alpha, s.check_termination) == (0.1, 1e-6, 10, 4000, 1.6, 25)
    assert (s.eps_abs, s.eps_rel, s.adaptive_rho, s.adaptive_rho_interval, s.polish, s.warm_start) == (1e-3, 1e-3, 1, 0, 0, 1)
    lib.osqp_cleanup.argtypes = [C.c_void_p]
    assert lib.osqp_cleanup(None) == 0


def test_oracle_exports_the_same_abi(oracle_lib):
    for sym in T.ABI_SYMBOLS:
        assert hasattr(oracle_lib, sym)


def test_struct_layouts_match_the_reference_mirrors():
    """Offsets from SURVEY.md 8b / [REF src/types.jl:11-217]."""
    assert C.sizeof(T.Ccsc) == 56 and T.Ccsc.nz.offset == 48
    assert C.sizeof(T.Data) == 56 and T.Data.u.offset == 48
    assert C.sizeof(T.Settings) == 176
    assert T.Settings.linsys_solver.offset == 104 and T.Settings.delta.offset == 112 and T.Settings.time_limit.offset == 168
    assert C.sizeof(T.CInfo) == 136 and T.CInfo.status_val.offset == 40 and T.CInfo.rho_estimate.offset == 128
    assert C.sizeof(T.Solution) == 16
    W = T.Workspace
    assert (W.data.offset, W.delta_y.offset, W.delta_x.offset, W.solution.offset, W.info.offset) == (0, 120, 136, 200, 208)
    assert W.first_run.offset == 224 and W.summary_printed.offset == 232


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a HIP device osqp_setup must fail (non-zero), never compute on the CPU."""
    import numpy as np
    import scipy.sparse as sp

    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except Exception:
        pass
    lib = oq.load_library()
    m = oq.Model(lib)
    with pytest.raises(oq.OSQPError):
        oq.setup(m, P=sp.identity(2, format="csc"), q=np.ones(2), A=sp.identity(2, format="csc"), l=-np.ones(2), u=np.ones(2), verbose=False)
