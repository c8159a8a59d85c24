"""Mirror of the reference's MathOptInterface face [REF src/MOI_wrapper.jl:1-928] over the same host API
(`interface.py`, hence over whichever C-ABI library the `Model` was made with) -- SURVEY.md row N2.

The reference's wrapper is host bookkeeping only: it flattens a MathOptInterface model into (P, q, A, l, u)
[REF :151-168, 231-345, 347-482], keeps modification / warm-start caches between solves [REF src/modcaches.jl],
maps OSQP's status to MOI's termination / primal / dual statuses [REF :617-687] and flips the dual sign convention
[REF :509, 735, 868].  MathOptInterface itself is a Julia package, so this module carries the few model-side types it
needs (indices, affine / quadratic / vector-affine functions, the supported sets, a plain model container standing in for
`OSQPModel` [REF :914-926]); names and argument meaning follow MOI, 1-based `value`s included, so that the tests read like
test/MOI_wrapper.jl.  Row indices inside the optimizer are 0-based (numpy).
"""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import scipy.sparse as sp

from . import interface as oq
from .constants import SOLUTION_PRESENT, UPDATABLE_SETTINGS
from .modcaches import ProblemModificationCache, WarmStartCache

INF = float("inf")

# ---------------------------------------------------------------- model-side types (MathOptInterface's, reduced)
MIN_SENSE, MAX_SENSE, FEASIBILITY_SENSE = "MIN_SENSE", "MAX_SENSE", "FEASIBILITY_SENSE"

# termination / result statuses used by the wrapper [REF src/MOI_wrapper.jl:617-687]
OPTIMIZE_NOT_CALLED, INTERRUPTED, DUAL_INFEASIBLE, INFEASIBLE = "OPTIMIZE_NOT_CALLED", "INTERRUPTED", "DUAL_INFEASIBLE", "INFEASIBLE"
ITERATION_LIMIT, OPTIMAL, ALMOST_OPTIMAL, ALMOST_INFEASIBLE = "ITERATION_LIMIT", "OPTIMAL", "ALMOST_OPTIMAL", "ALMOST_INFEASIBLE"
INVALID_MODEL, TIME_LIMIT = "INVALID_MODEL", "TIME_LIMIT"
NO_SOLUTION, FEASIBLE_POINT, UNKNOWN_RESULT_STATUS = "NO_SOLUTION", "FEASIBLE_POINT", "UNKNOWN_RESULT_STATUS"
INFEASIBILITY_CERTIFICATE, NEARLY_INFEASIBILITY_CERTIFICATE = "INFEASIBILITY_CERTIFICATE", "NEARLY_INFEASIBILITY_CERTIFICATE"


class SetAttributeNotAllowed(Exception):
    pass


class ModifyObjectiveNotAllowed(Exception):
    pass


class UnsupportedConstraint(Exception):
    pass


class UnsupportedAttribute(Exception):
    pass


class InvalidIndex(Exception):
    pass


@dataclass(frozen=True)
class VariableIndex:
    value: int  # 1-based, as MOI.VariableIndex


@dataclass(frozen=True)
class ConstraintIndex:
    value: int  # 1-based
    F: type = object
    S: type = object


@dataclass
class ScalarAffineTerm:
    coefficient: float
    variable: VariableIndex


@dataclass
class ScalarQuadraticTerm:
    coefficient: float
    variable_1: VariableIndex
    variable_2: VariableIndex


@dataclass
class ScalarAffineFunction:
    terms: List[ScalarAffineTerm]
    constant: float = 0.0


@dataclass
class ScalarQuadraticFunction:
    """MOI's convention: the function is 1/2 x'Qx + a'x + c with `coefficient` the entry of the symmetric Q, i.e. a
    diagonal term (c, x, x) contributes c/2 x^2 and an off-diagonal (c, x, y) contributes c x y."""
    quadratic_terms: List[ScalarQuadraticTerm]
    affine_terms: List[ScalarAffineTerm]
    constant: float = 0.0


@dataclass
class VectorAffineTerm:
    output_index: int  # 1-based
    scalar_term: ScalarAffineTerm


@dataclass
class VectorAffineFunction:
    terms: List[VectorAffineTerm]
    constants: List[float]


@dataclass(frozen=True)
class Interval:
    lower: float
    upper: float


@dataclass(frozen=True)
class LessThan:
    upper: float


@dataclass(frozen=True)
class GreaterThan:
    lower: float


@dataclass(frozen=True)
class EqualTo:
    value: float


@dataclass(frozen=True)
class Zeros:
    dimension: int


@dataclass(frozen=True)
class Nonnegatives:
    dimension: int


@dataclass(frozen=True)
class Nonpositives:
    dimension: int


IntervalConvertible = (Interval, LessThan, GreaterThan, EqualTo)
SupportedVectorSets = (Zeros, Nonnegatives, Nonpositives)


def as_interval(s):
    """MOI.Interval(s) for the interval-convertible scalar sets."""
    if isinstance(s, Interval):
        return s
    if isinstance(s, LessThan):
        return Interval(-INF, s.upper)
    if isinstance(s, GreaterThan):
        return Interval(s.lower, INF)
    if isinstance(s, EqualTo):
        return Interval(s.value, s.value)
    raise UnsupportedConstraint(type(s).__name__)


def dimension(s):
    return s.dimension if isinstance(s, SupportedVectorSets) else 1


def lower(s, i):  # [REF src/MOI_wrapper.jl:38-43]
    return -INF if isinstance(s, Nonpositives) else 0.0


def upper(s, i):
    return INF if isinstance(s, Nonnegatives) else 0.0


@dataclass
class ScalarConstantChange:
    new_constant: float


@dataclass
class ScalarCoefficientChange:
    variable: VariableIndex
    new_coefficient: float


class Model:
    """The source model of `copy_to` -- what `OSQPModel{Float64}` [REF src/MOI_wrapper.jl:914-926] is on the Julia side: scalar
    affine constraints in interval-convertible sets, vector affine constraints in Zeros / Nonnegatives / Nonpositives, an
    affine or quadratic objective, primal / dual starts.  Constraints are listed per (function, set) type in order of first
    appearance, as `MOI.ListOfConstraintTypesPresent` does."""

    def __init__(self):
        self.empty()

    def empty(self):
        self.nvars = 0
        self.constraints: Dict[ConstraintIndex, Tuple[object, object]] = {}
        self.sense = FEASIBILITY_SENSE
        self.objective = None
        self.primal_start: Dict[VariableIndex, Optional[float]] = {}
        self.dual_start: Dict[ConstraintIndex, object] = {}
        self._next_ci = 0

    # -- building
    def add_variable(self):
        self.nvars += 1
        return VariableIndex(self.nvars)

    def add_variables(self, k):
        return [self.add_variable() for _ in range(k)]

    def add_constraint(self, f, s):
        if isinstance(f, ScalarAffineFunction) and isinstance(s, IntervalConvertible):
            pass
        elif isinstance(f, VectorAffineFunction) and isinstance(s, SupportedVectorSets):
            if len(f.constants) != s.dimension:
                raise ValueError("dimension mismatch between the function and the set")
        else:
            raise UnsupportedConstraint(f"{type(f).__name__}-in-{type(s).__name__}")
        self._next_ci += 1
        ci = ConstraintIndex(self._next_ci, type(f), type(s))
        self.constraints[ci] = (f, s)
        return ci

    # -- attributes (set / get / modify by name, as MOI.set / MOI.get / MOI.modify on a ModelLike)
    def set_objective_sense(self, sense):
        self.sense = sense

    def set_objective_function(self, f):
        if not isinstance(f, (ScalarAffineFunction, ScalarQuadraticFunction)):
            raise UnsupportedAttribute(type(f).__name__)
        self.objective = f

    def set_constraint_function(self, ci, f):
        self.constraints[ci] = (f, self.constraints[ci][1])

    def set_constraint_set(self, ci, s):
        self.constraints[ci] = (self.constraints[ci][0], s)

    def set_primal_start(self, vi, value):
        self.primal_start[vi] = value

    def set_dual_start(self, ci, value):
        self.dual_start[ci] = value

    def modify_objective(self, change):
        f = self.objective
        if isinstance(change, ScalarConstantChange):
            f.constant = change.new_constant
        else:
            terms = f.terms if isinstance(f, ScalarAffineFunction) else f.affine_terms
            terms[:] = [t for t in terms if t.variable != change.variable] + [ScalarAffineTerm(change.new_coefficient, change.variable)]

    def modify_constraint(self, ci, change):
        f, _ = self.constraints[ci]
        f.terms[:] = [t for t in f.terms if t.variable != change.variable] + [ScalarAffineTerm(change.new_coefficient, change.variable)]

    # -- queries used by copy_to
    def variable_indices(self):
        return [VariableIndex(i + 1) for i in range(self.nvars)]

    def constraint_types_present(self):
        seen = []
        for ci in self.constraints:
            if (ci.F, ci.S) not in seen:
                seen.append((ci.F, ci.S))
        return seen

    def constraint_indices(self, F, S):
        return [ci for ci in self.constraints if ci.F is F and ci.S is S]


@dataclass
class IndexMap:
    var_map: Dict[VariableIndex, VariableIndex] = field(default_factory=dict)
    con_map: Dict[ConstraintIndex, ConstraintIndex] = field(default_factory=dict)

    def __getitem__(self, idx):
        return self.var_map[idx] if isinstance(idx, VariableIndex) else self.con_map[idx]


def _camel_to_setting(name):
    """`OSQPSettings.EpsAbs` <-> :eps_abs [REF src/MOI_wrapper.jl:522-535]"""
    out = []
    for ch in name:
        if ch.isupper() and out:
            out.append("_")
        out.append(ch.lower())
    return "".join(out)


# ---------------------------------------------------------------- the optimizer
class Optimizer:
    """[REF src/MOI_wrapper.jl:54-96].  `lib` selects the C-ABI library (None: the HIP engine)."""

    def __init__(self, lib=None, **kwargs):
        self.lib = lib
        self.inner = oq.Model(lib)
        self.hasresults = False
        self.results = oq.Results()
        self.silent = False
        self.settings = {"verbose": True}  # preserved across empty!, as in the reference
        self.sense = MIN_SENSE
        self.objconstant = 0.0
        self.constrconstant = np.zeros(0)
        self.modcache = None
        self.warmstartcache = None
        self.rowranges: Dict[int, range] = {}
        for key, value in kwargs.items():
            self.set_raw(key, value)

    # ---- housekeeping [REF :98-149]
    solver_name = "OSQP"

    def set_silent(self, value):
        self.silent = bool(value)
        if not self.is_empty():
            oq.update_settings(self.inner, verbose=False if self.silent else self.settings["verbose"])

    def set_time_limit_sec(self, limit):
        if limit is None:
            self.settings.pop("time_limit", None)
            if not self.is_empty():
                oq.update_settings(self.inner, time_limit=0.0)
        else:
            self.set_raw("time_limit", limit)

    def get_time_limit_sec(self):
        return self.settings.get("time_limit")

    def empty(self):
        self.inner = oq.Model(self.lib)
        self.hasresults = False
        self.results = oq.Results()
        self.sense = MIN_SENSE
        self.objconstant = 0.0
        self.constrconstant = np.zeros(0)
        self.modcache = None
        self.warmstartcache = None
        self.rowranges = {}
        return self

    def is_empty(self):
        return self.inner.isempty

    # ---- settings [REF :537-565]
    def set_raw(self, name, value):
        setting = _camel_to_setting(name) if not name.islower() else name
        if not (setting in UPDATABLE_SETTINGS or self.is_empty()):
            raise SetAttributeNotAllowed(setting)
        self.settings[setting] = value
        if not self.is_empty():
            oq.update_settings(self.inner, **{setting: value})

    def get_raw(self, name):
        setting = _camel_to_setting(name) if not name.islower() else name
        return self.settings[setting]

    # ---- copy_to [REF :151-168]
    def supports_constraint(self, F, S):
        return (F is ScalarAffineFunction and issubclass(S, IntervalConvertible)) or \
               (F is VectorAffineFunction and issubclass(S, SupportedVectorSets))

    def copy_to(self, src: Model):
        self.empty()
        idxmap = self._index_map(src)
        self._assign_constraint_row_ranges(idxmap, src)
        self.sense, P, q, self.objconstant = processobjective(src, idxmap)
        A, l, u, self.constrconstant = processconstraints(src, idxmap, self.rowranges)
        settings = dict(self.settings)
        if self.silent:
            settings["verbose"] = False
        oq.setup(self.inner, P=P, q=q, A=A, l=l, u=u, **settings)
        self.modcache = ProblemModificationCache(P, q, A, l, u)
        self.warmstartcache = WarmStartCache(A.shape[1], A.shape[0])
        processprimalstart(self.warmstartcache.x, src, idxmap)
        processdualstart(self.warmstartcache.y, src, idxmap, self.rowranges)
        return idxmap

    def _index_map(self, src):  # [REF :173-190]
        idxmap = IndexMap()
        for i, vi in enumerate(src.variable_indices()):
            idxmap.var_map[vi] = VariableIndex(i + 1)
        i = 0
        for F, S in src.constraint_types_present():
            if not self.supports_constraint(F, S):
                raise UnsupportedConstraint(f"{F.__name__}-in-{S.__name__}")
            for ci in src.constraint_indices(F, S):
                i += 1
                idxmap.con_map[ci] = ConstraintIndex(i, F, S)
        return idxmap

    def _assign_constraint_row_ranges(self, idxmap, src):  # [REF :192-209]
        startrow = 0
        for F, S in src.constraint_types_present():
            for ci_src in src.constraint_indices(F, S):
                s = src.constraints[ci_src][1]
                endrow = startrow + dimension(s)
                self.rowranges[idxmap[ci_src].value] = range(startrow, endrow)
                startrow = endrow

    def constraint_rows(self, ci):  # [REF :211-229]
        rows = self.rowranges[ci.value]
        if issubclass(ci.S, SupportedVectorSets):
            return rows
        if len(rows) != 1:
            raise RuntimeError("scalar constraint with more than one row")
        return rows[0]

    # ---- standard attributes [REF :513-521, 578-616]
    def get_objective_sense(self):
        return self.sense

    def number_of_variables(self):
        return oq.dimensions(self.inner)[0]

    def list_of_variable_indices(self):
        return [VariableIndex(i + 1) for i in range(self.number_of_variables())]

    def is_valid(self, idx):
        if isinstance(idx, VariableIndex):
            return 1 <= idx.value <= self.number_of_variables()
        return (not self.is_empty()) and idx.value in self.rowranges

    def raw_solver(self):
        return self.inner

    def result_count(self):
        return 1 if self.hasresults else 0

    # ---- optimize! [REF :567-576]
    def optimize(self):
        self.modcache.processupdates(self.inner)
        self.warmstartcache.processupdates(self.inner)
        oq.solve(self.inner, self.results)
        self.hasresults = True
        # the solution becomes the next warm start without setting the dirty bit
        self.warmstartcache.x.data[:] = self.results.x
        self.warmstartcache.y.data[:] = self.results.y

    # ---- objective [REF :589-646, 876-910]
    def set_objective_function(self, obj):
        if self.is_empty():
            raise SetAttributeNotAllowed("ObjectiveFunction")
        cache = self.modcache
        cache.P[:] = 0
        if isinstance(obj, ScalarAffineFunction):
            processlinearterms(cache.q, obj.terms)
            self.objconstant = obj.constant
            return
        for term in obj.quadratic_terms:
            row, col = term.variable_1.value - 1, term.variable_2.value - 1
            if row > col:
                row, col = col, row  # upper triangle only
            if (row, col) not in cache.P.cartesian_indices_set:
                raise SetAttributeNotAllowed(
                    "This nonzero entry was not in the sparsity pattern of the objective function provided at `copy_to` and "
                    "OSQP does not support changing the sparsity pattern.")
            cache.P.modifications[(row, col)] = cache.P.modifications.get((row, col), 0.0) + term.coefficient
        processlinearterms(cache.q, obj.affine_terms)
        self.objconstant = obj.constant

    def modify_objective(self, change):
        if self.is_empty():
            raise ModifyObjectiveNotAllowed(change)
        if isinstance(change, ScalarConstantChange):
            constant = change.new_constant
            self.objconstant = -constant if self.sense == MAX_SENSE else constant
        else:
            coef = change.new_coefficient
            self.modcache.q[change.variable.value - 1] = -coef if self.sense == MAX_SENSE else coef

    def objective_value(self):
        self._check_result()
        rawobj = self.results.info.obj_val + self.objconstant
        return -rawobj if self.sense == MAX_SENSE else rawobj

    def solve_time_sec(self):
        self._check_has_results()
        return self.results.info.run_time

    def raw_status_string(self):
        return str(self.results.info.status)

    def _check_has_results(self):
        if not self.hasresults:
            raise RuntimeError("Problem is unsolved.")

    def _check_result(self, result_index=1):
        if result_index > self.result_count():
            raise IndexError("result index out of bounds")

    # ---- statuses [REF :617-687]
    def termination_status(self):
        if not self.hasresults:
            return OPTIMIZE_NOT_CALLED
        s = self.results.info.status
        table = {"Unsolved": OPTIMIZE_NOT_CALLED, "Interrupted": INTERRUPTED, "Dual_infeasible": DUAL_INFEASIBLE,
                 "Primal_infeasible": INFEASIBLE, "Max_iter_reached": ITERATION_LIMIT, "Solved": OPTIMAL,
                 "Solved_inaccurate": ALMOST_OPTIMAL, "Primal_infeasible_inaccurate": ALMOST_INFEASIBLE}
        if s in table:
            return table[s]
        if s != "Non_convex":  # the reference asserts here: anything else is outside its status table [REF :640-642]
            raise AssertionError(f"status {s} has no MOI termination status in the reference")
        return INVALID_MODEL

    def primal_status(self, result_index=1):
        if result_index > self.result_count():
            return NO_SOLUTION
        s = self.results.info.status
        return {"Solved": FEASIBLE_POINT, "Primal_infeasible_inaccurate": UNKNOWN_RESULT_STATUS,
                "Dual_infeasible": INFEASIBILITY_CERTIFICATE}.get(s, NO_SOLUTION)

    def dual_status(self, result_index=1):
        if result_index > self.result_count():
            return NO_SOLUTION
        s = self.results.info.status
        return {"Primal_infeasible": INFEASIBILITY_CERTIFICATE, "Primal_infeasible_inaccurate": NEARLY_INFEASIBILITY_CERTIFICATE,
                "Solved": FEASIBLE_POINT}.get(s, NO_SOLUTION)

    # ---- variables / constraints [REF :689-760, 862-874]
    def variable_primal(self, vi):
        self._check_result()
        x = self.results.x if self.results.info.status in SOLUTION_PRESENT else self.results.dual_inf_cert
        if isinstance(vi, (list, tuple)):
            return np.array([x[v.value - 1] for v in vi])
        return x[vi.value - 1]

    def set_primal_start(self, vi, value):
        if self.is_empty():
            raise SetAttributeNotAllowed("VariablePrimalStart")
        self.warmstartcache.x[vi.value - 1] = value

    def set_dual_start(self, ci, value):
        if self.is_empty():
            raise SetAttributeNotAllowed("ConstraintDualStart")
        rows = self.constraint_rows(ci)
        if isinstance(rows, range):
            for i, row in enumerate(rows):
                self.warmstartcache.y[row] = -value[i]  # opposite dual convention
        else:
            self.warmstartcache.y[rows] = -(value[0] if isinstance(value, (list, tuple, np.ndarray)) else value)

    def constraint_dual(self, ci):
        self._check_result()
        y = self.results.y if self.results.info.status in SOLUTION_PRESENT else self.results.prim_inf_cert
        rows = self.constraint_rows(ci)
        if isinstance(rows, range):
            return -y[rows.start:rows.stop]
        return -y[rows]

    # ---- modifications [REF :762-860]
    def set_constraint_function(self, ci, f):
        if not self.is_valid(ci):
            raise InvalidIndex(ci)
        cache = self.modcache
        if isinstance(f, ScalarAffineFunction):
            row = self.constraint_rows(ci)
            cache.A[row, :] = 0
            for term in f.terms:
                key = (row, term.variable.value - 1)
                if key not in cache.A.cartesian_indices_set:
                    raise ValueError("Changing the sparsity pattern is not allowed.")
                cache.A.modifications[key] = cache.A.modifications.get(key, 0.0) + term.coefficient
            dconstant = self.constrconstant[row] - f.constant
            self.constrconstant[row] = f.constant
            cache.l[row] = cache.l[row] + dconstant
            cache.u[row] = cache.u[row] + dconstant
            return
        rows = self.constraint_rows(ci)
        for row in rows:
            cache.A[row, :] = 0
        for term in f.terms:
            key = (rows[term.output_index - 1], term.scalar_term.variable.value - 1)
            if key not in cache.A.cartesian_indices_set:
                raise ValueError("Changing the sparsity pattern is not allowed.")
            cache.A.modifications[key] = cache.A.modifications.get(key, 0.0) + term.scalar_term.coefficient
        for i, row in enumerate(rows):
            dconstant = self.constrconstant[row] - f.constants[i]
            self.constrconstant[row] = f.constants[i]
            cache.l[row] = cache.l[row] + dconstant
            cache.u[row] = cache.u[row] + dconstant

    def set_constraint_set(self, ci, s):
        if not self.is_valid(ci):
            raise InvalidIndex(ci)
        if not isinstance(s, ci.S):
            raise TypeError("the set type of a constraint cannot change")
        cache = self.modcache
        if isinstance(s, IntervalConvertible):
            interval = as_interval(s)
            row = self.constraint_rows(ci)
            constant = self.constrconstant[row]
            cache.l[row] = interval.lower - constant
            cache.u[row] = interval.upper - constant
            return
        for i, row in enumerate(self.constraint_rows(ci)):
            constant = self.constrconstant[row]
            cache.l[row] = lower(s, i) - constant
            cache.u[row] = upper(s, i) - constant

    def modify_constraint(self, ci, change: ScalarCoefficientChange):
        if not self.is_valid(ci):
            raise InvalidIndex(ci)
        row = self.constraint_rows(ci)
        self.modcache.A[row, change.variable.value - 1] = change.new_coefficient


# ---------------------------------------------------------------- flattening [REF src/MOI_wrapper.jl:231-511]
def processlinearterms(q, terms, idxmap=None):
    """q <- the dense coefficient vector of the terms (duplicates add up); q may be a vector or a VectorModificationCache"""
    q[:] = 0
    for term in terms:
        var = idxmap[term.variable] if idxmap is not None else term.variable
        q[var.value - 1] = q[var.value - 1] + term.coefficient


def processobjective(src: Model, idxmap):
    """sense, P (upper triangle as given, duplicates summed), q, c such that the objective is 1/2 x'Px + q'x + c"""
    sense = src.sense
    n = src.nvars
    q = np.zeros(n)
    if sense != FEASIBILITY_SENSE:
        f = src.objective
        if isinstance(f, ScalarAffineFunction):
            P = sp.csc_matrix((n, n))
            processlinearterms(q, f.terms, idxmap)
            c = f.constant
        elif isinstance(f, ScalarQuadraticFunction):
            I = [idxmap[t.variable_1].value - 1 for t in f.quadratic_terms]
            J = [idxmap[t.variable_2].value - 1 for t in f.quadratic_terms]
            V = [t.coefficient for t in f.quadratic_terms]
            for k in range(len(V)):  # upper_triangularize!
                if I[k] > J[k]:
                    I[k], J[k] = J[k], I[k]
            # sparse(I, J, V) keeps explicitly stored zeros: the pattern of P is what the caller wrote down
            P = _sparse_keep_zeros(I, J, V, n, n)
            processlinearterms(q, f.affine_terms, idxmap)
            c = f.constant
        else:
            raise UnsupportedAttribute("ObjectiveFunction")
        if sense == MAX_SENSE:
            P = -P
            q = -q
            c = -c
    else:
        P = sp.csc_matrix((n, n))
        c = 0.0
    return sense, P, q, c


def _sparse_keep_zeros(I, J, V, m, n):
    """Julia's sparse(I, J, V, m, n): duplicates are summed and entries whose sum is zero stay in the pattern."""
    M = sp.coo_matrix((np.asarray(V, dtype=float), (np.asarray(I, dtype=int), np.asarray(J, dtype=int))), shape=(m, n)).tocsc()
    M.sum_duplicates()
    M.sort_indices()
    return M


def processconstraints(src: Model, idxmap, rowranges):
    m = sum(len(r) for r in rowranges.values())
    l = np.empty(m)
    u = np.empty(m)
    constant = np.empty(m)
    I, J, V = [], [], []
    for F, S in src.constraint_types_present():
        for ci in src.constraint_indices(F, S):
            f, s = src.constraints[ci]
            rows = rowranges[idxmap[ci].value]
            if isinstance(f, ScalarAffineFunction):
                row = rows[0]
                constant[row] = f.constant
                for term in f.terms:
                    I.append(row); J.append(idxmap[term.variable].value - 1); V.append(term.coefficient)
                interval = as_interval(s)
                l[row], u[row] = interval.lower, interval.upper
            else:
                for i, row in enumerate(rows):
                    constant[row] = f.constants[i]
                    l[row], u[row] = lower(s, i), upper(s, i)
                for term in f.terms:
                    I.append(rows[term.output_index - 1]); J.append(idxmap[term.scalar_term.variable].value - 1)
                    V.append(term.scalar_term.coefficient)
    l = l - constant
    u = u - constant
    A = _sparse_keep_zeros(I, J, V, m, src.nvars)
    return A, l, u, constant


def processprimalstart(x, src: Model, idxmap):
    for vi, value in src.primal_start.items():
        if value is not None:
            x[idxmap[vi].value - 1] = value


def processdualstart(y, src: Model, idxmap, rowranges):
    for ci, dual in src.dual_start.items():
        if dual is None:
            continue
        rows = rowranges[idxmap[ci].value]
        vals = dual if isinstance(dual, (list, tuple, np.ndarray)) else [dual]
        for i, row in enumerate(rows):
            y[row] = -vals[i]  # opposite dual convention
