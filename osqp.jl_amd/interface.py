"""Host-side mirror of the reference's native API [REF src/interface.jl:18-773]
over the C ABI declared in include/osqp_amd.h.

The reference is Julia; Julia is not available in this image, so the host side
that is actually exercised is this Python mirror: same names (minus the ``!``
that Python identifiers cannot carry), same argument meaning, same error
behaviour.  ``julia/OSQPAMD.jl`` is the Julia statement of the same layer.

    model = Model()
    setup(model, P=P, q=q, A=A, l=l, u=u, eps_abs=1e-4, ...)
    results = solve(model)
    update(model, q=q_new); update_settings(model, rho=0.2); warm_start(model, x=x0, y=y0)
"""
import ctypes as C
import math

import numpy as np
import scipy.sparse as sp

from . import types as T
from .constants import (
    OSQP_INFTY,
    QDLDL_SOLVER,
    MKL_PARDISO_SOLVER,
    AMD_PCG_SOLVER,
    AMD_DIRECT_SOLVER,
    SOLUTION_PRESENT,
    UPDATABLE_SETTINGS,
    status_map,
)


class OSQPError(RuntimeError):
    """Counterpart of the ErrorException the Julia layer throws on a non-zero exit flag."""


class Info:
    """[REF src/types.jl:219-236]"""

    __slots__ = (
        "iter", "status", "status_val", "status_polish", "obj_val", "pri_res", "dua_res",
        "setup_time", "solve_time", "update_time", "polish_time", "run_time", "rho_updates", "rho_estimate",
    )

    def __init__(self):
        for k in self.__slots__:
            setattr(self, k, 0)
        self.status = "Unsolved"

    def copy_from(self, cinfo):
        """[REF src/types.jl:238-254]; an unknown status_val raises KeyError as in the reference."""
        self.iter = int(cinfo.iter)
        self.status = status_map[int(cinfo.status_val)]
        self.status_val = int(cinfo.status_val)
        self.status_polish = int(cinfo.status_polish)
        for k in ("obj_val", "pri_res", "dua_res", "setup_time", "solve_time", "update_time", "polish_time",
                  "run_time", "rho_estimate"):
            setattr(self, k, float(getattr(cinfo, k)))
        self.rho_updates = int(cinfo.rho_updates)
        return self

    def __repr__(self):
        return "Info(" + ", ".join(f"{k}={getattr(self, k)!r}" for k in self.__slots__) + ")"


class Results:
    """[REF src/types.jl:256-272]"""

    def __init__(self):
        self.x = np.zeros(0)
        self.y = np.zeros(0)
        self.info = Info()
        self.prim_inf_cert = np.zeros(0)
        self.dual_inf_cert = np.zeros(0)

    def resize(self, n, m):
        if self.x.shape[0] != n:
            self.x = np.empty(n)
            self.dual_inf_cert = np.empty(n)
        if self.y.shape[0] != m:
            self.y = np.empty(m)
            self.prim_inf_cert = np.empty(m)
        return self


class Model:
    """[REF src/interface.jl:18-28]  Handle on a library workspace.

    ``lib`` is the dlopen'ed ABI library; the default is the product (HIP)
    library and it is an error if that is not built."""

    def __init__(self, lib=None):
        self.lib = lib if lib is not None else T.load_library()
        self.workspace = T.Workspace_p()  # NULL
        self.lcache = np.zeros(0)
        self.ucache = np.zeros(0)
        self.isempty = True

    def __del__(self):
        try:
            clean(self)
        except Exception:
            pass


def _as_f64(v):
    return np.ascontiguousarray(np.asarray(v, dtype=np.float64))


def _fptr(a):
    return a.ctypes.data_as(T.c_float_p)


def _iptr(a):
    return a.ctypes.data_as(T.c_int_p)


class ManagedCcsc:
    """[REF src/types.jl:21-47]  0-based CSC arrays with Cc_int indices, kept alive
    for the duration of the call; nz = -1 marks compressed-column form."""

    def __init__(self, M, canonical=True):
        M = sp.csc_matrix(M)
        if canonical and not M.has_canonical_format:
            # rows of a column ascending and unique, as a Julia SparseMatrixCSC always is (the library refuses anything else
            # with exit flag 1); on a copy: sp.csc_matrix(M) of a CSC input shares the caller's arrays
            M = M.copy()
            M.sum_duplicates()
        self.m, self.n = M.shape
        self.x = np.ascontiguousarray(M.data, dtype=np.float64)
        self.i = np.ascontiguousarray(M.indices, dtype=np.int64)
        self.p = np.ascontiguousarray(M.indptr, dtype=np.int64)
        self.ccsc = T.Ccsc(len(self.x), self.m, self.n, _iptr(self.p), _iptr(self.i), _fptr(self.x), -1)


def ccsc_to_scipy(c):
    """[REF src/types.jl:49-57]"""
    n, nzmax = int(c.n), int(c.nzmax)
    p = np.array([c.p[k] for k in range(n + 1)], dtype=np.int64)
    i = np.array([c.i[k] for k in range(nzmax)], dtype=np.int64)
    x = np.array([c.x[k] for k in range(nzmax)], dtype=np.float64)
    return sp.csc_matrix((x, i, p), shape=(int(c.m), n))


def default_settings(lib=None):
    """[REF src/types.jl:136-145]"""
    lib = lib if lib is not None else T.load_library()
    s = T.Settings()
    lib.osqp_set_default_settings(C.byref(s))
    return s


def linsys_solver_str_to_int(settings):
    """[REF src/interface.jl:749-773] (+ the two extension names)."""
    v = settings.get("linsys_solver")
    if v is None:
        return
    if not isinstance(v, str):
        raise OSQPError("linsys_solver is required to be a string")
    v = v.lower()
    table = {"qdldl": QDLDL_SOLVER, "mkl pardiso": MKL_PARDISO_SOLVER, "": QDLDL_SOLVER,
             "pcg": AMD_PCG_SOLVER, "direct": AMD_DIRECT_SOLVER}
    if v not in table:
        import warnings

        warnings.warn("Linear system solver not recognized. Using default QDLDL")
    settings["linsys_solver"] = table.get(v, QDLDL_SOLVER)


def make_settings(lib, settings):
    """[REF src/types.jl:147-171]  defaults from C, overridden by keyword, each
    converted to the field's C type; unknown keys are silently ignored."""
    s = default_settings(lib)
    settings = dict(settings)
    linsys_solver_str_to_int(settings)
    for name, ctype in T.Settings._fields_:
        if name in settings:
            v = settings[name]
            if ctype is T.c_float:
                v = float(v)
            else:
                if isinstance(v, float) and not float(v).is_integer():
                    raise OSQPError(f"setting {name} must be an integer")
                v = int(v)
            setattr(s, name, v)
    return s


def _istriu(P):
    """istriu(P) of [REF src/interface.jl:88-90] on a CSC matrix, without building its lower triangle."""
    if P.nnz == 0:
        return True
    first = P.indptr[:-1]
    nonempty = P.indptr[1:] > first
    if P.has_sorted_indices:  # the last stored row of every column decides
        last = P.indices[P.indptr[1:][nonempty] - 1]
        return not np.any(last > np.nonzero(nonempty)[0])
    cols = np.repeat(np.arange(P.shape[1], dtype=P.indices.dtype), np.diff(P.indptr))
    return not np.any(P.indices > cols)


def setup(model, P=None, q=None, A=None, l=None, u=None, comm=None, keep_A_order=False, **settings):
    """[REF src/interface.jl:35-162].  Extension: `comm` (sharded.HostComm / sharded.RcclComm) makes this rank keep
    its row block of one QP shared by all ranks of the communicator (osqp_amd_setup_sharded).  `keep_A_order` (tests): the
    CSC arrays of A go to the library as they are, rows of a column in the caller's order (a plain-C caller of libosqp may
    hand them over unsorted; Julia's SparseMatrixCSC never does)."""
    if P is None:
        if q is not None:
            n = len(q)
        elif A is not None:
            n = A.shape[1]
        else:
            raise OSQPError("The problem does not have any variables!")
    else:
        n = P.shape[0]
    m = 0 if A is None else A.shape[0]
    if (A is None and (l is not None or u is not None)) or (A is not None and l is None and u is None):
        raise OSQPError("A must be supplied together with l and u")
    if A is not None and l is None:
        l = -np.inf * np.ones(m)
    if A is not None and u is None:
        u = np.inf * np.ones(m)
    if P is None:
        P = sp.csc_matrix((n, n))
    if q is None:
        q = np.zeros(n)
    if A is None:
        A = sp.csc_matrix((m, n))
        l = np.zeros(m)
        u = np.zeros(m)
    q, l, u = _as_f64(q), _as_f64(l), _as_f64(u)
    if len(q) != n:
        raise OSQPError("Incorrect dimension of q")
    if len(l) != m:
        raise OSQPError("Incorrect dimensions of l")
    if len(u) != m:
        raise OSQPError("Incorrect dimensions of u")
    P = sp.csc_matrix(P)
    if not _istriu(P):
        P = sp.triu(P, format="csc")
    u = np.minimum(u, OSQP_INFTY)
    l = np.maximum(l, -OSQP_INFTY)
    model.lcache = np.empty(m)
    model.ucache = np.empty(m)
    managedP = ManagedCcsc(P)
    managedA = ManagedCcsc(sp.csc_matrix(A), canonical=False) if keep_A_order else ManagedCcsc(sp.csc_matrix(A))
    stgs = make_settings(model.lib, settings)
    data = T.Data(n, m, C.pointer(managedP.ccsc), C.pointer(managedA.ccsc), _fptr(q), _fptr(l), _fptr(u))
    workspace = T.Workspace_p()
    if comm is None:
        exitflag = model.lib.osqp_setup(C.byref(workspace), C.byref(data), C.byref(stgs))
    else:
        exitflag = model.lib.osqp_amd_setup_sharded(C.byref(workspace), C.byref(data), C.byref(stgs), comm.handle)
        model.comm = comm  # the communicator must outlive the workspace
    model.workspace = workspace
    if exitflag != 0:
        model.workspace = T.Workspace_p()
        raise OSQPError("Error in OSQP setup")
    model.isempty = False
    return model


def setup_generated(model, kind, n, per_row=0, seed=1, comm=None, **settings):
    """Extension: build one of the synthetic families directly where the library
    wants it (HBM for the product) and run setup on it (osqp_amd_setup_generated)."""
    stgs = make_settings(model.lib, settings)
    workspace = T.Workspace_p()
    if comm is None:
        exitflag = model.lib.osqp_amd_setup_generated(C.byref(workspace), kind, n, per_row, seed, C.byref(stgs))
    else:
        exitflag = model.lib.osqp_amd_setup_generated_sharded(C.byref(workspace), kind, n, per_row, seed, C.byref(stgs),
                                                              comm.handle)
        model.comm = comm
    if exitflag != 0:
        raise OSQPError("Error in OSQP setup")
    model.workspace = workspace
    (nn, m) = dimensions(model)
    model.lcache = np.empty(m)
    model.ucache = np.empty(m)
    model.isempty = False
    return model


def solve(model, results=None):
    """[REF src/interface.jl:164-217]"""
    if model.isempty:
        raise OSQPError("You are trying to solve an empty model. Please setup the model before calling solve!().")
    if results is None:
        results = Results()
    model.lib.osqp_solve(model.workspace)  # return value ignored, as in the reference
    workspace = model.workspace.contents
    info = results.info
    info.copy_from(workspace.info.contents)
    solution = workspace.solution.contents
    data = workspace.data.contents
    n, m = int(data.n), int(data.m)
    results.resize(n, m)
    if info.status in SOLUTION_PRESENT:
        C.memmove(results.x.ctypes.data, solution.x, 8 * n)
        C.memmove(results.y.ctypes.data, solution.y, 8 * m)
        results.prim_inf_cert.fill(np.nan)
        results.dual_inf_cert.fill(np.nan)
    else:
        results.x.fill(np.nan)
        results.y.fill(np.nan)
        if info.status in ("Primal_infeasible", "Primal_infeasible_inaccurate"):
            C.memmove(results.prim_inf_cert.ctypes.data, workspace.delta_y, 8 * m)
            results.dual_inf_cert.fill(np.nan)
        elif info.status in ("Dual_infeasible", "Dual_infeasible_inaccurate"):
            results.prim_inf_cert.fill(np.nan)
            C.memmove(results.dual_inf_cert.ctypes.data, workspace.delta_x, 8 * n)
        else:
            results.prim_inf_cert.fill(np.nan)
            results.dual_inf_cert.fill(np.nan)
    if info.status == "Non_convex":
        info.obj_val = math.nan
    return results


def version(lib=None):
    """[REF src/interface.jl:219-221]"""
    lib = lib if lib is not None else T.load_library()
    return lib.osqp_version().decode()


def clean(model):
    """[REF src/interface.jl:223-233]; the workspace pointer may be NULL."""
    ws, model.workspace = model.workspace, T.Workspace_p()
    model.isempty = True
    exitflag = model.lib.osqp_cleanup(ws)
    if exitflag != 0:
        raise OSQPError("Error in OSQP cleanup")


def dimensions(model):
    """[REF src/interface.jl:740-747]"""
    if not model.workspace:
        raise OSQPError("Workspace has not been setup yet")
    data = model.workspace.contents.data.contents
    return int(data.n), int(data.m)


def update_q(model, q):
    """[REF src/interface.jl:235-251]"""
    n, m = dimensions(model)
    q = _as_f64(q)
    if len(q) != n:
        raise OSQPError(f"q must have length n = {n}")
    if model.lib.osqp_update_lin_cost(model.workspace, _fptr(q)) != 0:
        raise OSQPError("Error updating q")


def update_l(model, l):
    """[REF src/interface.jl:253-269]"""
    n, m = dimensions(model)
    l = _as_f64(l)
    if len(l) != m:
        raise OSQPError(f"l must have length m = {m}")
    np.maximum(l, -OSQP_INFTY, out=model.lcache)
    if model.lib.osqp_update_lower_bound(model.workspace, _fptr(model.lcache)) != 0:
        raise OSQPError("Error updating l")


def update_u(model, u):
    """[REF src/interface.jl:271-287]"""
    n, m = dimensions(model)
    u = _as_f64(u)
    if len(u) != m:
        raise OSQPError(f"u must have length m = {m}")
    np.minimum(u, OSQP_INFTY, out=model.ucache)
    if model.lib.osqp_update_upper_bound(model.workspace, _fptr(model.ucache)) != 0:
        raise OSQPError("Error updating u")


def update_bounds(model, l, u):
    """[REF src/interface.jl:289-313]"""
    n, m = dimensions(model)
    l, u = _as_f64(l), _as_f64(u)
    if len(l) != m:
        raise OSQPError(f"l must have length m = {m}")
    if len(u) != m:
        raise OSQPError(f"u must have length m = {m}")
    np.maximum(l, -OSQP_INFTY, out=model.lcache)
    np.minimum(u, OSQP_INFTY, out=model.ucache)
    if model.lib.osqp_update_bounds(model.workspace, _fptr(model.lcache), _fptr(model.ucache)) != 0:
        raise OSQPError("Error updating bounds l and u")


def _prep_idx(idx, nvals, name):
    """[REF src/interface.jl:315-328]  Python callers pass 0-based positions already
    (numpy convention), so there is no in-place shift to undo."""
    if idx is None:
        return None, None
    idx = np.ascontiguousarray(np.asarray(idx, dtype=np.int64))
    if len(idx) != nvals:
        raise OSQPError(f"{name} and {name}_idx must have the same length")
    return idx, _iptr(idx)


def update_P(model, Px, Px_idx=None):
    """[REF src/interface.jl:330-349]"""
    Px = _as_f64(Px)
    keep, ptr = _prep_idx(Px_idx, len(Px), "P")
    if model.lib.osqp_update_P(model.workspace, _fptr(Px), ptr, len(Px)) != 0:
        raise OSQPError("Error updating P")


def update_A(model, Ax, Ax_idx=None):
    """[REF src/interface.jl:351-370]"""
    Ax = _as_f64(Ax)
    keep, ptr = _prep_idx(Ax_idx, len(Ax), "A")
    if model.lib.osqp_update_A(model.workspace, _fptr(Ax), ptr, len(Ax)) != 0:
        raise OSQPError("Error updating A")


def update_P_A(model, Px, Px_idx, Ax, Ax_idx):
    """[REF src/interface.jl:372-406]"""
    Px, Ax = _as_f64(Px), _as_f64(Ax)
    keepP, pP = _prep_idx(Px_idx, len(Px), "P")
    keepA, pA = _prep_idx(Ax_idx, len(Ax), "A")
    if model.lib.osqp_update_P_A(model.workspace, _fptr(Px), pP, len(Px), _fptr(Ax), pA, len(Ax)) != 0:
        raise OSQPError("Error updating P and A")


def update(model, q=None, l=None, u=None, Px=None, Px_idx=None, Ax=None, Ax_idx=None):
    """[REF src/interface.jl:408-440]"""
    if q is not None:
        update_q(model, q)
    if l is not None and u is not None:
        update_bounds(model, l, u)
    elif l is not None:
        update_l(model, l)
    elif u is not None:
        update_u(model, u)
    if Px is not None and Ax is not None:
        update_P_A(model, Px, Px_idx, Ax, Ax_idx)
    elif Px is not None:
        update_P(model, Px, Px_idx)
    elif Ax is not None:
        update_A(model, Ax, Ax_idx)


_INT_SETTINGS = ("max_iter", "polish", "polish_refine_iter", "verbose", "check_termination", "warm_start")
_FLOAT_SETTINGS = ("eps_abs", "eps_rel", "eps_prim_inf", "eps_dual_inf", "rho", "alpha", "delta", "time_limit")


def update_settings(model, **kwargs):
    """[REF src/interface.jl:442-670]  One single-value C call per setting.  As in
    the reference, ``scaled_termination`` is not in UPDATABLE_SETTINGS and the
    branch that would call osqp_update_scaled_termination is unreachable from
    here [REF src/interface.jl:468, 617-628]; the C symbol itself is exported."""
    if not kwargs:
        return
    for key in kwargs:
        if key not in UPDATABLE_SETTINGS:
            raise OSQPError(f"{key} cannot be updated or is not recognized")
    for key in UPDATABLE_SETTINGS:  # fixed order, like the reference's sequence of ifs
        if key not in kwargs or kwargs[key] is None:
            continue
        fn = getattr(model.lib, "osqp_update_" + key)
        v = kwargs[key]
        v = int(v) if key in _INT_SETTINGS else float(v)
        if fn(model.workspace, v) != 0:
            raise OSQPError(f"Error updating {key}")


def warm_start_x(model, x):
    """[REF src/interface.jl:672-684]"""
    n, m = dimensions(model)
    x = _as_f64(x)
    if len(x) != n:
        raise OSQPError("Wrong dimension for variable x")
    if model.lib.osqp_warm_start_x(model.workspace, _fptr(x)) != 0:
        raise OSQPError("Error in warm starting x")


def warm_start_y(model, y):
    """[REF src/interface.jl:686-698]"""
    n, m = dimensions(model)
    y = _as_f64(y)
    if len(y) != m:
        raise OSQPError("Wrong dimension for variable y")
    if model.lib.osqp_warm_start_y(model.workspace, _fptr(y)) != 0:
        raise OSQPError("Error in warm starting y")


def warm_start_x_y(model, x, y):
    """[REF src/interface.jl:700-718]"""
    n, m = dimensions(model)
    x, y = _as_f64(x), _as_f64(y)
    if len(x) != n:
        raise OSQPError("Wrong dimension for variable x")
    if len(y) != m:
        raise OSQPError("Wrong dimension for variable y")
    if model.lib.osqp_warm_start(model.workspace, _fptr(x), _fptr(y)) != 0:
        raise OSQPError("Error in warm starting x and y")


def warm_start(model, x=None, y=None):
    """[REF src/interface.jl:720-732]"""
    if x is not None and y is not None:
        warm_start_x_y(model, x, y)
    elif x is not None:
        warm_start_x(model, x)
    elif y is not None:
        warm_start_y(model, y)


def stats(model, count=26):
    """Extension: osqp_amd_get_stats as a list of floats."""
    out = np.zeros(count)
    k = model.lib.osqp_amd_get_stats(model.workspace, _fptr(out), count)
    return out[:k]
