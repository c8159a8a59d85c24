"""ctypes mirrors of the C-ABI structs, field for field as the reference's Julia
mirrors [REF src/types.jl:5-217], plus the library loader.

The product library is ``csrc/libosqp_amd.so`` (HIP, built by
``__graft_entry__.build()``).  Loading it fails loudly when it is missing:
there is no CPU fallback in the product path.  Tests may pass the path of the
CPU oracle (``oracle/_build/libosqp_oracle.so``, same symbols) explicitly.
"""
import ctypes as C
import os

c_int = C.c_longlong  # [REF src/types.jl:5-9]  Cc_int = Clonglong
c_float = C.c_double
c_int_p = C.POINTER(c_int)
c_float_p = C.POINTER(c_float)


class Ccsc(C.Structure):
    """[REF src/types.jl:11-19]"""

    _fields_ = [
        ("nzmax", c_int),
        ("m", c_int),
        ("n", c_int),
        ("p", c_int_p),
        ("i", c_int_p),
        ("x", c_float_p),
        ("nz", c_int),
    ]


class Solution(C.Structure):
    """[REF src/types.jl:74-77]"""

    _fields_ = [("x", c_float_p), ("y", c_float_p)]


class CInfo(C.Structure):
    """[REF src/types.jl:81-99]"""

    _fields_ = [
        ("iter", c_int),
        ("status", C.c_char * 32),
        ("status_val", c_int),
        ("status_polish", c_int),
        ("obj_val", c_float),
        ("pri_res", c_float),
        ("dua_res", c_float),
        ("setup_time", c_float),
        ("solve_time", c_float),
        ("update_time", c_float),
        ("polish_time", c_float),
        ("run_time", c_float),
        ("rho_updates", c_int),
        ("rho_estimate", c_float),
    ]


class Data(C.Structure):
    """[REF src/types.jl:101-109]"""

    _fields_ = [
        ("n", c_int),
        ("m", c_int),
        ("P", C.POINTER(Ccsc)),
        ("A", C.POINTER(Ccsc)),
        ("q", c_float_p),
        ("l", c_float_p),
        ("u", c_float_p),
    ]


class Settings(C.Structure):
    """[REF src/types.jl:111-134]; linsys_solver is a 32-bit enum."""

    _fields_ = [
        ("rho", c_float),
        ("sigma", c_float),
        ("scaling", c_int),
        ("adaptive_rho", c_int),
        ("adaptive_rho_interval", c_int),
        ("adaptive_rho_tolerance", c_float),
        ("adaptive_rho_fraction", c_float),
        ("max_iter", c_int),
        ("eps_abs", c_float),
        ("eps_rel", c_float),
        ("eps_prim_inf", c_float),
        ("eps_dual_inf", c_float),
        ("alpha", c_float),
        ("linsys_solver", C.c_int),
        ("delta", c_float),
        ("polish", c_int),
        ("polish_refine_iter", c_int),
        ("verbose", c_int),
        ("scaled_termination", c_int),
        ("check_termination", c_int),
        ("warm_start", c_int),
        ("time_limit", c_float),
    ]


class Workspace(C.Structure):
    """[REF src/types.jl:173-217] (+ the library-private tail pointer)."""

    _fields_ = [
        ("data", C.POINTER(Data)),
        ("linsys_solver", C.c_void_p),
        ("pol", C.c_void_p),
        ("rho_vec", c_float_p),
        ("rho_inv_vec", c_float_p),
        ("constr_type", c_int_p),
        ("x", c_float_p),
        ("y", c_float_p),
        ("z", c_float_p),
        ("xz_tilde", c_float_p),
        ("x_prev", c_float_p),
        ("z_prev", c_float_p),
        ("Ax", c_float_p),
        ("Px", c_float_p),
        ("Aty", c_float_p),
        ("delta_y", c_float_p),
        ("Atdelta_y", c_float_p),
        ("delta_x", c_float_p),
        ("Pdelta_x", c_float_p),
        ("Adelta_x", c_float_p),
        ("D_temp", c_float_p),
        ("D_temp_A", c_float_p),
        ("E_temp", c_float_p),
        ("settings", C.POINTER(Settings)),
        ("scaling", C.c_void_p),
        ("solution", C.POINTER(Solution)),
        ("info", C.POINTER(CInfo)),
        ("timer", C.c_void_p),
        ("first_run", c_int),
        ("summary_printed", c_int),
        ("impl", C.c_void_p),
    ]


Workspace_p = C.POINTER(Workspace)

# the 30 symbols OSQP.jl binds: name -> (restype, argtypes)   [SURVEY.md 8b]
ABI_SYMBOLS = {
    "osqp_set_default_settings": (None, [C.POINTER(Settings)]),
    "osqp_setup": (c_int, [C.POINTER(Workspace_p), C.POINTER(Data), C.POINTER(Settings)]),
    "osqp_solve": (c_int, [Workspace_p]),
    "osqp_version": (C.c_char_p, []),
    "osqp_cleanup": (c_int, [Workspace_p]),
    "osqp_update_lin_cost": (c_int, [Workspace_p, c_float_p]),
    "osqp_update_lower_bound": (c_int, [Workspace_p, c_float_p]),
    "osqp_update_upper_bound": (c_int, [Workspace_p, c_float_p]),
    "osqp_update_bounds": (c_int, [Workspace_p, c_float_p, c_float_p]),
    "osqp_update_P": (c_int, [Workspace_p, c_float_p, c_int_p, c_int]),
    "osqp_update_A": (c_int, [Workspace_p, c_float_p, c_int_p, c_int]),
    "osqp_update_P_A": (c_int, [Workspace_p, c_float_p, c_int_p, c_int, c_float_p, c_int_p, c_int]),
    "osqp_update_max_iter": (c_int, [Workspace_p, c_int]),
    "osqp_update_polish": (c_int, [Workspace_p, c_int]),
    "osqp_update_polish_refine_iter": (c_int, [Workspace_p, c_int]),
    "osqp_update_verbose": (c_int, [Workspace_p, c_int]),
    "osqp_update_scaled_termination": (c_int, [Workspace_p, c_int]),
    "osqp_update_check_termination": (c_int, [Workspace_p, c_int]),
    "osqp_update_warm_start": (c_int, [Workspace_p, c_int]),
    "osqp_update_eps_abs": (c_int, [Workspace_p, c_float]),
    "osqp_update_eps_rel": (c_int, [Workspace_p, c_float]),
    "osqp_update_eps_prim_inf": (c_int, [Workspace_p, c_float]),
    "osqp_update_eps_dual_inf": (c_int, [Workspace_p, c_float]),
    "osqp_update_rho": (c_int, [Workspace_p, c_float]),
    "osqp_update_alpha": (c_int, [Workspace_p, c_float]),
    "osqp_update_delta": (c_int, [Workspace_p, c_float]),
    "osqp_update_time_limit": (c_int, [Workspace_p, c_float]),
    "osqp_warm_start_x": (c_int, [Workspace_p, c_float_p]),
    "osqp_warm_start_y": (c_int, [Workspace_p, c_float_p]),
    "osqp_warm_start": (c_int, [Workspace_p, c_float_p, c_float_p]),
}

# extension entry points (include/osqp_amd.h part 2); optional in the oracle
# int fn(void *ctx, double *host_buf, long long count)  (include/osqp_amd.h: osqp_amd_allgather_fn)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_longlong)

EXT_SYMBOLS = {
    "osqp_amd_setup_generated": (c_int, [C.POINTER(Workspace_p), c_int, c_int, c_int, C.c_ulonglong, C.POINTER(Settings)]),
    "osqp_amd_get_stats": (c_int, [Workspace_p, c_float_p, c_int]),
    "osqp_amd_time_kernel": (c_float, [Workspace_p, c_int, c_int]),
    "osqp_amd_iterate": (c_int, [Workspace_p, c_int]),
    "osqp_amd_get_iterate": (c_int, [Workspace_p, c_float_p, c_float_p]),
    "osqp_amd_apply": (c_int, [Workspace_p, c_int, c_float_p, c_float_p]),
    "osqp_amd_batch_solve": (
        c_int,
        [c_int, c_int, c_int, c_int_p, c_int_p, c_float_p, c_int_p, c_int_p, c_float_p,
         c_float_p, c_float_p, c_float_p, C.POINTER(Settings), c_float_p, c_float_p, C.POINTER(CInfo), c_int],
    ),
    "osqp_amd_batch_solve_generated": (
        c_int,
        [c_int, c_int, C.c_ulonglong, C.POINTER(Settings), C.c_void_p, C.c_void_p, C.c_void_p, c_int],
    ),
    "osqp_amd_comm_create_host": (c_int, [C.POINTER(C.c_void_p), c_int, c_int, ALLGATHER_FN, C.c_void_p]),
    "osqp_amd_comm_unique_id": (c_int, [C.c_void_p, C.c_char_p]),
    "osqp_amd_comm_create_rccl": (c_int, [C.POINTER(C.c_void_p), c_int, c_int, C.c_void_p, C.c_char_p]),
    "osqp_amd_comm_destroy": (c_int, [C.c_void_p]),
    "osqp_amd_comm_all_gather": (c_int, [C.c_void_p, C.c_void_p, c_int]),
    "osqp_amd_comm_info": (c_int, [C.c_void_p, c_int_p, c_int_p, c_int_p]),
    "osqp_amd_batch_mpc_create": (c_int, [C.POINTER(C.c_void_p), c_int, C.c_ulonglong, C.POINTER(Settings), C.c_void_p, c_int]),
    "osqp_amd_batch_mpc_solve": (c_int, [C.c_void_p, C.c_void_p]),
    "osqp_amd_batch_destroy": (c_int, [C.c_void_p]),
    "osqp_amd_batch_last_kernel": (c_int, []),
    "osqp_amd_device_alloc": (C.c_void_p, [c_int, c_int]),
    "osqp_amd_device_free": (c_int, [C.c_void_p, c_int]),
    "osqp_amd_device_copy": (c_int, [C.c_void_p, C.c_void_p, c_int, c_int, c_int]),
    "osqp_amd_setup_sharded": (c_int, [C.POINTER(Workspace_p), C.POINTER(Data), C.POINTER(Settings), C.c_void_p]),
    "osqp_amd_setup_generated_sharded": (
        c_int, [C.POINTER(Workspace_p), c_int, c_int, c_int, C.c_ulonglong, C.POINTER(Settings), C.c_void_p]),
    "osqp_amd_last_error": (C.c_char_p, []),
    "osqp_amd_set_device": (c_int, [c_int]),
    "osqp_amd_symbolic_probe": (c_int, [c_int, c_int, c_int_p, c_int_p, c_int_p, c_int_p, c_int, c_int, c_float_p, c_int]),
    # oracle-only helpers
    "oracle_generate": (C.POINTER(Data), [c_int, c_int, c_int, C.c_ulonglong]),
    "oracle_data_free": (None, [C.POINTER(Data)]),
}

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
PRODUCT_LIB_PATH = os.environ.get("OSQP_AMD_LIB") or os.path.join(_PKG_DIR, "csrc", "libosqp_amd.so")
ORACLE_LIB_PATH = os.path.join(os.path.dirname(_PKG_DIR), "oracle", "_build", "libosqp_oracle.so")

_libs = {}


def _preload_hip_runtime():
    """torch ships its own libamdhip64.so.7; a process that loads the system one first
    (through libosqp_amd.so) and torch's later ends up with two HIP runtimes and torch
    finds no GPU.  Loading torch's copy first (same SONAME) makes both share it.  Only
    needed when torch will be used in the process (bench.py, batch.py, the GPU tests)."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        cand = os.path.join(libdir, name)
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=os.RTLD_GLOBAL | os.RTLD_NOW)
            except OSError:
                pass


def load_library(path=None):
    """dlopen a library exporting the osqp_* ABI and declare its signatures.

    ``path=None`` means the product library; a missing product library is a
    hard error (no CPU fallback)."""
    if path is None:
        path = PRODUCT_LIB_PATH
        if not os.path.exists(path):
            raise RuntimeError(
                "libosqp_amd.so (the HIP engine) is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C osqp.jl_amd/csrc`.  There is no CPU fallback in the product path."
            )
    path = os.path.abspath(path)
    if path in _libs:
        return _libs[path]
    if path == os.path.abspath(PRODUCT_LIB_PATH):
        _preload_hip_runtime()
    lib = C.CDLL(path, mode=os.RTLD_LOCAL | os.RTLD_NOW)
    for name, (res, args) in ABI_SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in EXT_SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            continue
        fn.restype = res
        fn.argtypes = args
    _libs[path] = lib
    return lib
