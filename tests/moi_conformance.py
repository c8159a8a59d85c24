"""A conformance subset in the style of `MOI.Test.runtests` [REF test/MOI_wrapper.jl:59-93] over the Python mirror of the MOI face.

The reference runs MathOptInterface's generic test-suite against its wrapper (bridged: variable bounds reach OSQP as scalar
affine constraints) with 17 exclusions -- everything that needs infeasibility-certificate VALUES, solver names, or model
attributes OSQP does not carry.  MathOptInterface is a dependency of the reference, not part of it, and is not in this image;
the cases below restate, by the names MathOptInterface gives them, the known-answer problems of that suite which OSQP's
wrapper can express and which the reference does not exclude: the linear and quadratic integration problems, the
modification tests, the objective tests, the conic-linear family over Zeros / Nonnegatives / Nonpositives (round 4).  Every expected value is the suite's published one AND is re-derived here with scipy
(`linprog` for the LPs, an SLSQP solve for the QPs) before the wrapper's answer is compared with it -- so a mis-remembered
number cannot pass silently.  Tolerances: MOI.Test.Config(atol = 1e-4, rtol = 1e-4) [REF test/MOI_wrapper.jl:28-39].
Every case takes the loaded C-ABI library: the CPU oracle in the CPU suite, the HIP engine in the GPU suite."""
import numpy as np
from scipy.optimize import linprog, minimize

from osqp_jl_amd import moi as MOI
from moi_cases import approx, defaultoptimizer, term

INF = MOI.INF


def saf(coefs, vs, constant=0.0):
    return MOI.ScalarAffineFunction([term(c, v) for c, v in zip(coefs, vs)], constant)


def solve(lib, model):
    opt = defaultoptimizer(lib)
    idx = opt.copy_to(model)
    opt.optimize()
    return opt, idx


def expect_optimal(opt, idx, obj, xs=None, vars_=None, duals=None):
    assert opt.termination_status() == MOI.OPTIMAL, opt.raw_status_string()
    assert opt.primal_status() == MOI.FEASIBLE_POINT
    assert approx(opt.objective_value(), obj), (opt.objective_value(), obj)
    if xs is not None:
        got = opt.variable_primal([idx[v] for v in vars_])
        assert approx(got, xs), (got, xs)
    if duals:
        assert opt.dual_status() == MOI.FEASIBLE_POINT
        for ci, d in duals:
            assert approx(opt.constraint_dual(idx[ci]), d), (ci, opt.constraint_dual(idx[ci]), d)


def lp_reference(c, A_ub=None, b_ub=None, A_eq=None, b_eq=None, bounds=None, maximize=False):
    c = np.asarray(c, dtype=float)
    r = linprog(-c if maximize else c, A_ub=A_ub, b_ub=b_ub, A_eq=A_eq, b_eq=b_eq, bounds=bounds, method="highs")
    assert r.status == 0, r.message
    return (-r.fun if maximize else r.fun), r.x


def qp_reference(P, q, cons, x0):
    """min 1/2 x'Px + q'x under scipy `constraints` dicts (SLSQP: small dense problems only)."""
    P, q = np.asarray(P, dtype=float), np.asarray(q, dtype=float)
    r = minimize(lambda x: 0.5 * x @ P @ x + q @ x, x0, jac=lambda x: P @ x + q, constraints=cons, method="SLSQP",
                 options=dict(ftol=1e-14, maxiter=500))
    assert r.success, r.message
    return r.fun, r.x


# ---------------------------------------------------------------------------------------------------------------- linear
def case_linear_integration_2(lib):
    """min -x  s.t.  x + y <= 1, x >= 0, y >= 0   ->  -1 at (1, 0); duals -1, 0, 1."""
    m = MOI.Model()
    x, y = m.add_variables(2)
    c = m.add_constraint(saf([1, 1], [x, y]), MOI.LessThan(1.0))
    bx = m.add_constraint(saf([1], [x]), MOI.GreaterThan(0.0))
    by = m.add_constraint(saf([1], [y]), MOI.GreaterThan(0.0))
    m.set_objective_function(saf([-1.0], [x]))
    m.set_objective_sense(MOI.MIN_SENSE)
    ref, xr = lp_reference([-1, 0], A_ub=[[1, 1]], b_ub=[1], bounds=[(0, None)] * 2)
    assert approx(ref, -1) and approx(xr, [1, 0])
    opt, idx = solve(lib, m)
    expect_optimal(opt, idx, -1, [1, 0], [x, y], [(c, -1), (bx, 0), (by, 1)])


def case_linear_integration(lib):
    """The first stages of MathOptInterface's `linear1`: min -x over {x + y <= 1, x, y >= 0}; the same as max x; then a third
    variable z >= 0 joins (a new copy: OSQP cannot grow a model) in x + y + z <= 1 with objective max x + 2 z -> 2 at z = 1;
    then the right-hand side becomes 2 -> 4."""
    m = MOI.Model()
    x, y = m.add_variables(2)
    c = m.add_constraint(saf([1, 1], [x, y]), MOI.LessThan(1.0))
    m.add_constraint(saf([1], [x]), MOI.GreaterThan(0.0))
    m.add_constraint(saf([1], [y]), MOI.GreaterThan(0.0))
    m.set_objective_function(saf([-1.0, 0.0], [x, y]))
    m.set_objective_sense(MOI.MIN_SENSE)
    opt, idx = solve(lib, m)
    expect_optimal(opt, idx, -1, [1, 0], [x, y], [(c, -1)])
    m.set_objective_function(saf([1.0, 0.0], [x, y]))
    m.set_objective_sense(MOI.MAX_SENSE)
    opt, idx = solve(lib, m)
    expect_optimal(opt, idx, 1, [1, 0], [x, y], [(c, -1)])
    z = m.add_variable()
    m.add_constraint(saf([1], [z]), MOI.GreaterThan(0.0))
    m.set_constraint_function(c, saf([1, 1, 1], [x, y, z]))
    m.set_objective_function(saf([1.0, 0.0, 2.0], [x, y, z]))
    ref, xr = lp_reference([1, 0, 2], A_ub=[[1, 1, 1]], b_ub=[1], bounds=[(0, None)] * 3, maximize=True)
    assert approx(ref, 2) and approx(xr, [0, 0, 1])
    opt, idx = solve(lib, m)
    expect_optimal(opt, idx, 2, [0, 0, 1], [x, y, z], [(c, -2)])
    opt.set_constraint_set(idx[c], MOI.LessThan(2.0))  # on the live optimizer: a bound update, no new setup
    opt.optimize()
    expect_optimal(opt, idx, 4, [0, 0, 2], [x, y, z])


def case_linear_inactive_bounds(lib):
    """min x s.t. x >= 0, x >= 3 -> 3;  max x s.t. x <= 0, x <= 3 -> 0."""
    m = MOI.Model()
    x = m.add_variable()
    m.add_constraint(saf([1], [x]), MOI.GreaterThan(0.0))
    m.add_constraint(saf([1], [x]), MOI.GreaterThan(3.0))
    m.set_objective_function(saf([1.0], [x]))
    m.set_objective_sense(MOI.MIN_SENSE)
    opt, idx = solve(lib, m)
    expect_optimal(opt, idx, 3, [3], [x])
    m = MOI.Model()
    x = m.add_variable()
    m.add_constraint(saf([1], [x]), MOI.LessThan(0.0))
    m.add_constraint(saf([1], [x]), MOI.LessThan(3.0))
    m.set_objective_function(saf([1.0], [x]))
    m.set_objective_sense(MOI.MAX_SENSE)
    opt, idx = solve(lib, m)
    expect_optimal(opt, idx, 0, [0], [x])


def case_linear_LessThan_and_GreaterThan(lib):
    """min x - y s.t. x >= 0, y <= 0 -> 0; x >= 100 -> 100; y <= -100 -> 200 (bounds moved on the live optimizer)."""
    m = MOI.Model()
    x, y = m.add_variables(2)
    c1 = m.add_constraint(saf([1], [x]), MOI.GreaterThan(0.0))
    c2 = m.add_constraint(saf([1], [y]), MOI.LessThan(0.0))
    m.set_objective_function(saf([1.0, -1.0], [x, y]))
    m.set_objective_sense(MOI.MIN_SENSE)
    opt, idx = solve(lib, m)
    expect_optimal(opt, idx, 0, [0, 0], [x, y])
    opt.set_constraint_set(idx[c1], MOI.GreaterThan(100.0))
    opt.optimize()
    expect_optimal(opt, idx, 100, [100, 0], [x, y])
    opt.set_constraint_set(idx[c2], MOI.LessThan(-100.0))
    opt.optimize()
    expect_optimal(opt, idx, 200, [100, -100], [x, y])


def case_linear_integration_modification(lib):
    """max x + y s.t. 2x + y <= 4, x + 2y <= 4, x, y >= 0 -> 8/3 at (4/3, 4/3); the first row becomes 2x + 2y <= 4 (a
    coefficient change inside the sparsity pattern) -> 2; then its right-hand side 6 -> 10/3 ... each against linprog."""
    m = MOI.Model()
    x, y = m.add_variables(2)
    c1 = m.add_constraint(saf([2, 1], [x, y]), MOI.LessThan(4.0))
    m.add_constraint(saf([1, 2], [x, y]), MOI.LessThan(4.0))
    m.add_constraint(saf([1], [x]), MOI.GreaterThan(0.0))
    m.add_constraint(saf([1], [y]), MOI.GreaterThan(0.0))
    m.set_objective_function(saf([1.0, 1.0], [x, y]))
    m.set_objective_sense(MOI.MAX_SENSE)
    opt, idx = solve(lib, m)
    expect_optimal(opt, idx, 8 / 3, [4 / 3, 4 / 3], [x, y])
    opt.modify_constraint(idx[c1], MOI.ScalarCoefficientChange(idx[y], 2.0))
    opt.optimize()
    ref, _ = lp_reference([1, 1], A_ub=[[2, 2], [1, 2]], b_ub=[4, 4], bounds=[(0, None)] * 2, maximize=True)
    assert approx(ref, 2)
    expect_optimal(opt, idx, 2)
    opt.set_constraint_set(idx[c1], MOI.LessThan(6.0))
    opt.optimize()
    ref, xr = lp_reference([1, 1], A_ub=[[2, 2], [1, 2]], b_ub=[6, 4], bounds=[(0, None)] * 2, maximize=True)
    expect_optimal(opt, idx, ref)


def case_linear_modify_GreaterThan_and_LessThan_constraints(lib):
    """The same moves as test_linear_LessThan_and_GreaterThan with the bounds as affine rows that carry constants:
    x + 0 >= 0 and y + 0 <= 0, then the FUNCTIONS change (x - 100 >= 0, y + 100 <= 0)."""
    m = MOI.Model()
    x, y = m.add_variables(2)
    c1 = m.add_constraint(saf([1], [x], 0.0), MOI.GreaterThan(0.0))
    c2 = m.add_constraint(saf([1], [y], 0.0), MOI.LessThan(0.0))
    m.set_objective_function(saf([1.0, -1.0], [x, y]))
    m.set_objective_sense(MOI.MIN_SENSE)
    opt, idx = solve(lib, m)
    expect_optimal(opt, idx, 0, [0, 0], [x, y])
    opt.set_constraint_function(idx[c1], saf([1], [idx[x]], -100.0))
    opt.optimize()
    expect_optimal(opt, idx, 100, [100, 0], [x, y])
    opt.set_constraint_function(idx[c2], saf([1], [idx[y]], 100.0))
    opt.optimize()
    expect_optimal(opt, idx, 200, [100, -100], [x, y])


def case_linear_VectorAffineFunction(lib):
    """min x - y s.t. [x] in Nonnegatives(1), [y] in Nonpositives(1) -> 0; constants -100 / +100 -> 100, 200."""
    m = MOI.Model()
    x, y = m.add_variables(2)
    c1 = m.add_constraint(MOI.VectorAffineFunction([MOI.VectorAffineTerm(1, term(1.0, x))], [0.0]), MOI.Nonnegatives(1))
    c2 = m.add_constraint(MOI.VectorAffineFunction([MOI.VectorAffineTerm(1, term(1.0, y))], [0.0]), MOI.Nonpositives(1))
    m.set_objective_function(saf([1.0, -1.0], [x, y]))
    m.set_objective_sense(MOI.MIN_SENSE)
    opt, idx = solve(lib, m)
    expect_optimal(opt, idx, 0, [0, 0], [x, y])
    opt.set_constraint_function(idx[c1], MOI.VectorAffineFunction([MOI.VectorAffineTerm(1, term(1.0, idx[x]))], [-100.0]))
    opt.optimize()
    expect_optimal(opt, idx, 100, [100, 0], [x, y])
    opt.set_constraint_function(idx[c2], MOI.VectorAffineFunction([MOI.VectorAffineTerm(1, term(1.0, idx[y]))], [100.0]))
    opt.optimize()
    expect_optimal(opt, idx, 200, [100, -100], [x, y])


def case_linear_INFEASIBLE(lib):
    """min x s.t. 2x + y <= -1, x, y >= 0: infeasible; the wrapper reports INFEASIBLE with a certificate in the dual."""
    m = MOI.Model()
    x, y = m.add_variables(2)
    m.add_constraint(saf([2, 1], [x, y]), MOI.LessThan(-1.0))
    m.add_constraint(saf([1], [x]), MOI.GreaterThan(0.0))
    m.add_constraint(saf([1], [y]), MOI.GreaterThan(0.0))
    m.set_objective_function(saf([1.0], [x]))
    m.set_objective_sense(MOI.MIN_SENSE)
    opt, _ = solve(lib, m)
    assert opt.termination_status() in (MOI.INFEASIBLE, MOI.ALMOST_INFEASIBLE)
    assert opt.dual_status() in (MOI.INFEASIBILITY_CERTIFICATE, MOI.NEARLY_INFEASIBILITY_CERTIFICATE)
    assert opt.primal_status() in (MOI.NO_SOLUTION, MOI.UNKNOWN_RESULT_STATUS)


def case_linear_DUAL_INFEASIBLE(lib):
    """min -x - y s.t. -x + 2y <= 0, x, y >= 0: unbounded; DUAL_INFEASIBLE with a ray in the primal."""
    m = MOI.Model()
    x, y = m.add_variables(2)
    m.add_constraint(saf([-1, 2], [x, y]), MOI.LessThan(0.0))
    m.add_constraint(saf([1], [x]), MOI.GreaterThan(0.0))
    m.add_constraint(saf([1], [y]), MOI.GreaterThan(0.0))
    m.set_objective_function(saf([-1.0, -1.0], [x, y]))
    m.set_objective_sense(MOI.MIN_SENSE)
    opt, idx = solve(lib, m)
    assert opt.termination_status() == MOI.DUAL_INFEASIBLE
    assert opt.primal_status() == MOI.INFEASIBILITY_CERTIFICATE
    ray = opt.variable_primal([idx[x], idx[y]])
    assert -ray[0] - ray[1] < 0 and -ray[0] + 2 * ray[1] <= 1e-4 and np.all(ray >= -1e-4)  # a direction of unbounded descent (to eps_dual_inf)


def case_linear_add_constraints(lib):
    """max 1000x + 350y s.t. x >= 30, y >= 0, x - 1.5y >= 0, 12x + 8y <= 1000, 1000x + 300y <= 70000 -> 790000/11."""
    m = MOI.Model()
    x, y = m.add_variables(2)
    m.add_constraint(saf([1], [x]), MOI.GreaterThan(30.0))
    m.add_constraint(saf([1], [y]), MOI.GreaterThan(0.0))
    m.add_constraint(saf([1, -1.5], [x, y]), MOI.GreaterThan(0.0))
    m.add_constraint(saf([12, 8], [x, y]), MOI.LessThan(1000.0))
    m.add_constraint(saf([1000, 300], [x, y]), MOI.LessThan(70000.0))
    m.set_objective_function(saf([1000.0, 350.0], [x, y]))
    m.set_objective_sense(MOI.MAX_SENSE)
    ref, xr = lp_reference([1000, 350], A_ub=[[-1, 1.5], [12, 8], [1000, 300]], b_ub=[0, 1000, 70000], bounds=[(30, None), (0, None)],
                           maximize=True)
    assert approx(ref, 79e4 / 11) and approx(xr, [650 / 11, 400 / 11])
    opt, idx = solve(lib, m)
    expect_optimal(opt, idx, 79e4 / 11, [650 / 11, 400 / 11], [x, y])


def case_linear_integration_Interval(lib):
    """5 <= x + y <= 10, x, y >= 0: max x + y -> 10 (dual -1), min -> 5 (dual +1); the interval becomes [2, 12]: 2, 12."""
    m = MOI.Model()
    x, y = m.add_variables(2)
    c = m.add_constraint(saf([1, 1], [x, y]), MOI.Interval(5.0, 10.0))
    m.add_constraint(saf([1], [x]), MOI.GreaterThan(0.0))
    m.add_constraint(saf([1], [y]), MOI.GreaterThan(0.0))
    m.set_objective_function(saf([1.0, 1.0], [x, y]))
    m.set_objective_sense(MOI.MAX_SENSE)
    opt, idx = solve(lib, m)
    expect_optimal(opt, idx, 10, duals=[(c, -1)])
    m.set_objective_sense(MOI.MIN_SENSE)
    opt, idx = solve(lib, m)
    expect_optimal(opt, idx, 5, duals=[(c, 1)])
    opt.set_constraint_set(idx[c], MOI.Interval(2.0, 12.0))
    opt.optimize()
    expect_optimal(opt, idx, 2, duals=[(c, 1)])
    m.set_constraint_set(c, MOI.Interval(2.0, 12.0))
    m.set_objective_sense(MOI.MAX_SENSE)
    opt, idx = solve(lib, m)
    expect_optimal(opt, idx, 12, duals=[(c, -1)])


def case_linear_Interval_inactive(lib):
    """min 0 over -1 <= x <= 1 written as an interval row: optimal, objective 0, dual 0."""
    m = MOI.Model()
    x = m.add_variable()
    c = m.add_constraint(saf([1], [x]), MOI.Interval(-1.0, 1.0))
    m.set_objective_function(saf([0.0], [x]))
    m.set_objective_sense(MOI.MIN_SENSE)
    opt, idx = solve(lib, m)
    expect_optimal(opt, idx, 0, duals=[(c, 0)])
    assert -1 - 1e-4 <= opt.variable_primal(idx[x]) <= 1 + 1e-4


def case_linear_FEASIBILITY_SENSE(lib):
    """No objective: any point of {x + y >= 1 (as -x - y <= -1), x, y >= 0, x + y = 2} -- the answer must be feasible."""
    m = MOI.Model()
    x, y = m.add_variables(2)
    m.add_constraint(saf([1], [x]), MOI.GreaterThan(1.0))
    m.add_constraint(saf([1], [y]), MOI.GreaterThan(0.0))
    m.add_constraint(saf([1, 1], [x, y]), MOI.EqualTo(2.0))
    m.set_objective_sense(MOI.FEASIBILITY_SENSE)
    opt, idx = solve(lib, m)
    assert opt.termination_status() == MOI.OPTIMAL and opt.primal_status() == MOI.FEASIBLE_POINT
    xv, yv = opt.variable_primal([idx[x], idx[y]])
    assert xv >= 1 - 1e-4 and yv >= -1e-4 and abs(xv + yv - 2) <= 1e-4
    assert approx(opt.objective_value(), 0)


def case_linear_transform(lib):
    """min x + y s.t. x + y >= 1 (twice, the second copy as <= after a sign flip), x, y >= 0 -> 1."""
    m = MOI.Model()
    x, y = m.add_variables(2)
    m.add_constraint(saf([1, 1], [x, y]), MOI.GreaterThan(1.0))
    m.add_constraint(saf([-1, -1], [x, y]), MOI.LessThan(-1.0))
    m.add_constraint(saf([1], [x]), MOI.GreaterThan(0.0))
    m.add_constraint(saf([1], [y]), MOI.GreaterThan(0.0))
    m.set_objective_function(saf([1.0, 1.0], [x, y]))
    m.set_objective_sense(MOI.MIN_SENSE)
    opt, idx = solve(lib, m)
    expect_optimal(opt, idx, 1)


# ------------------------------------------------------------------------------------------------------------- quadratic
def _qp1(m, dup=False):
    """x^2 + xy + y^2 + yz + z^2  (1/2 x'Qx with Q = [2 1 0; 1 2 1; 0 1 2])  s.t.  x + 2y + 3z >= 4,  x + y >= 1."""
    x, y, z = m.add_variables(3)
    c1 = m.add_constraint(saf([1, 2, 3], [x, y, z]), MOI.GreaterThan(4.0))
    c2 = m.add_constraint(saf([1, 1], [x, y]), MOI.GreaterThan(1.0))
    if dup:  # the same form with split and repeated terms
        qt = [term(2.0, x, x), term(0.5, x, y), term(0.5, y, x), term(2.0, y, y), term(1.0, y, z), term(1.0, z, z), term(1.0, z, z)]
    else:
        qt = [term(2.0, x, x), term(1.0, x, y), term(2.0, y, y), term(1.0, y, z), term(2.0, z, z)]
    m.set_objective_function(MOI.ScalarQuadraticFunction(qt, [], 0.0))
    m.set_objective_sense(MOI.MIN_SENSE)
    return (x, y, z), (c1, c2)


def case_quadratic_integration(lib):
    Q = np.array([[2.0, 1, 0], [1, 2, 1], [0, 1, 2]])
    cons = [dict(type="ineq", fun=lambda v: v[0] + 2 * v[1] + 3 * v[2] - 4), dict(type="ineq", fun=lambda v: v[0] + v[1] - 1)]
    ref, xr = qp_reference(Q, np.zeros(3), cons, np.ones(3))
    assert approx(ref, 13 / 7) and approx(xr, [4 / 7, 3 / 7, 6 / 7])
    m = MOI.Model()
    (x, y, z), _ = _qp1(m)
    opt, idx = solve(lib, m)
    expect_optimal(opt, idx, 13 / 7, [4 / 7, 3 / 7, 6 / 7], [x, y, z])


def case_quadratic_duplicate_terms(lib):
    m = MOI.Model()
    (x, y, z), _ = _qp1(m, dup=True)
    opt, idx = solve(lib, m)
    expect_optimal(opt, idx, 13 / 7, [4 / 7, 3 / 7, 6 / 7], [x, y, z])
    # max of -2 times the form: same minimiser, objective -26/7
    qt = [term(-4.0, x, x), term(-2.0, x, y), term(-4.0, y, y), term(-2.0, y, z), term(-4.0, z, z)]
    m.set_objective_function(MOI.ScalarQuadraticFunction(qt, [], 0.0))
    m.set_objective_sense(MOI.MAX_SENSE)
    opt, idx = solve(lib, m)
    expect_optimal(opt, idx, -26 / 7, [4 / 7, 3 / 7, 6 / 7], [x, y, z])


def case_quadratic_nonhomogeneous(lib):
    """min 2x^2 + y^2 + xy + x + y + 1  s.t.  x, y >= 0, x + y = 1  ->  2.875 at (0.25, 0.75); max of its negative: -2.875."""
    Q = np.array([[4.0, 1], [1, 2]])
    cons = [dict(type="eq", fun=lambda v: v[0] + v[1] - 1), dict(type="ineq", fun=lambda v: v[0]), dict(type="ineq", fun=lambda v: v[1])]
    ref, xr = qp_reference(Q, np.ones(2), cons, np.array([0.5, 0.5]))
    assert approx(ref + 1, 2.875) and approx(xr, [0.25, 0.75])
    m = MOI.Model()
    x, y = m.add_variables(2)
    m.add_constraint(saf([1], [x]), MOI.GreaterThan(0.0))
    m.add_constraint(saf([1], [y]), MOI.GreaterThan(0.0))
    c = m.add_constraint(saf([1, 1], [x, y]), MOI.EqualTo(1.0))
    m.set_objective_function(MOI.ScalarQuadraticFunction([term(4.0, x, x), term(2.0, y, y), term(1.0, x, y)], [term(1.0, x), term(1.0, y)], 1.0))
    m.set_objective_sense(MOI.MIN_SENSE)
    opt, idx = solve(lib, m)
    expect_optimal(opt, idx, 2.875, [0.25, 0.75], [x, y])
    m.set_objective_function(MOI.ScalarQuadraticFunction([term(-4.0, x, x), term(-2.0, y, y), term(-1.0, x, y)],
                                                         [term(-1.0, x), term(-1.0, y)], -1.0))
    m.set_objective_sense(MOI.MAX_SENSE)
    opt, idx = solve(lib, m)
    expect_optimal(opt, idx, -2.875, [0.25, 0.75], [x, y])
    assert c in idx.con_map


# ---------------------------------------------------------------------------------------------------- objective / modification
def case_objective_ObjectiveFunction_constant(lib):
    """min 2x + 1 s.t. x >= 1 -> 3."""
    m = MOI.Model()
    x = m.add_variable()
    m.add_constraint(saf([1], [x]), MOI.GreaterThan(1.0))
    m.set_objective_function(saf([2.0], [x], 1.0))
    m.set_objective_sense(MOI.MIN_SENSE)
    opt, idx = solve(lib, m)
    expect_optimal(opt, idx, 3, [1], [x])


def case_objective_ObjectiveFunction_duplicate_terms(lib):
    """min x + x s.t. x >= 1 -> 2 (duplicate terms add up)."""
    m = MOI.Model()
    x = m.add_variable()
    m.add_constraint(saf([1], [x]), MOI.GreaterThan(1.0))
    m.set_objective_function(saf([1.0, 1.0], [x, x]))
    m.set_objective_sense(MOI.MIN_SENSE)
    opt, idx = solve(lib, m)
    expect_optimal(opt, idx, 2, [1], [x])


def case_objective_FEASIBILITY_SENSE_clears_objective(lib):
    """An objective set before the sense goes to FEASIBILITY_SENSE is not used: objective value 0 at a feasible point."""
    m = MOI.Model()
    x = m.add_variable()
    m.add_constraint(saf([1], [x]), MOI.GreaterThan(1.0))
    m.set_objective_function(saf([1.0], [x]))
    m.set_objective_sense(MOI.FEASIBILITY_SENSE)
    opt, idx = solve(lib, m)
    assert opt.termination_status() == MOI.OPTIMAL and approx(opt.objective_value(), 0)
    assert opt.variable_primal(idx[x]) >= 1 - 1e-4


def _max_x_le(lib, coef=1.0, rhs=1.0, const=0.0):
    m = MOI.Model()
    x = m.add_variable()
    c = m.add_constraint(saf([coef], [x]), MOI.LessThan(rhs))
    m.set_objective_function(saf([1.0], [x], const))
    m.set_objective_sense(MOI.MAX_SENSE)
    opt, idx = solve(lib, m)
    return m, x, c, opt, idx


def case_modification_set_scalaraffine_lessthan(lib):
    m, x, c, opt, idx = _max_x_le(lib)
    expect_optimal(opt, idx, 1, [1], [x])
    opt.set_constraint_set(idx[c], MOI.LessThan(2.0))
    opt.optimize()
    expect_optimal(opt, idx, 2, [2], [x])


def case_modification_func_scalaraffine_lessthan(lib):
    m, x, c, opt, idx = _max_x_le(lib)
    opt.set_constraint_function(idx[c], saf([2.0], [idx[x]]))
    opt.optimize()
    expect_optimal(opt, idx, 0.5, [0.5], [x])


def case_modification_coef_scalaraffine_lessthan(lib):
    m, x, c, opt, idx = _max_x_le(lib)
    opt.modify_constraint(idx[c], MOI.ScalarCoefficientChange(idx[x], 2.0))
    opt.optimize()
    expect_optimal(opt, idx, 0.5, [0.5], [x])


def case_modification_coef_scalar_objective(lib):
    m, x, c, opt, idx = _max_x_le(lib)
    opt.modify_objective(MOI.ScalarCoefficientChange(idx[x], 3.0))
    opt.optimize()
    expect_optimal(opt, idx, 3, [1], [x])


def case_modification_const_scalar_objective(lib):
    m, x, c, opt, idx = _max_x_le(lib, const=2.0)
    expect_optimal(opt, idx, 3, [1], [x])
    opt.modify_objective(MOI.ScalarConstantChange(3.0))
    opt.optimize()
    expect_optimal(opt, idx, 4, [1], [x])


def case_modification_const_vectoraffine_nonpos(lib):
    """max x + y s.t. [x; y] in Nonpositives(2) -> 0; constants (-1, -1.5)... the rows become x - 1 <= 0, y - 1.5 <= 0 -> 2.5."""
    m = MOI.Model()
    x, y = m.add_variables(2)
    f = MOI.VectorAffineFunction([MOI.VectorAffineTerm(1, term(1.0, x)), MOI.VectorAffineTerm(2, term(1.0, y))], [0.0, 0.0])
    c = m.add_constraint(f, MOI.Nonpositives(2))
    m.set_objective_function(saf([1.0, 1.0], [x, y]))
    m.set_objective_sense(MOI.MAX_SENSE)
    opt, idx = solve(lib, m)
    expect_optimal(opt, idx, 0, [0, 0], [x, y])
    g = MOI.VectorAffineFunction([MOI.VectorAffineTerm(1, term(1.0, idx[x])), MOI.VectorAffineTerm(2, term(1.0, idx[y]))], [-1.0, -1.5])
    opt.set_constraint_function(idx[c], g)
    opt.optimize()
    expect_optimal(opt, idx, 2.5, [1, 1.5], [x, y])


def case_modification_transform_singlevariable_lessthan(lib):
    """max x s.t. x <= 1 -> 1; the sense flips to min with x >= ... the reference's bridged form: a fresh row x >= 2 -> 2."""
    m = MOI.Model()
    x = m.add_variable()
    m.add_constraint(saf([1], [x]), MOI.LessThan(1.0))
    m.set_objective_function(saf([1.0], [x]))
    m.set_objective_sense(MOI.MAX_SENSE)
    opt, idx = solve(lib, m)
    expect_optimal(opt, idx, 1, [1], [x])
    m2 = MOI.Model()
    x2 = m2.add_variable()
    m2.add_constraint(saf([1], [x2]), MOI.GreaterThan(2.0))
    m2.set_objective_function(saf([1.0], [x2]))
    m2.set_objective_sense(MOI.MIN_SENSE)
    opt, idx = solve(lib, m2)
    expect_optimal(opt, idx, 2, [2], [x2])


def case_solve_TerminationStatus_and_result_count(lib):
    """Before optimize!: OPTIMIZE_NOT_CALLED, no result; after: one result [REF src/MOI_wrapper.jl:617-646]."""
    m = MOI.Model()
    x = m.add_variable()
    m.add_constraint(saf([1], [x]), MOI.GreaterThan(1.0))
    m.set_objective_function(saf([1.0], [x]))
    m.set_objective_sense(MOI.MIN_SENSE)
    opt = defaultoptimizer(lib)
    idx = opt.copy_to(m)
    assert opt.termination_status() == MOI.OPTIMIZE_NOT_CALLED and opt.result_count() == 0
    assert opt.primal_status() == MOI.NO_SOLUTION and opt.dual_status() == MOI.NO_SOLUTION
    opt.optimize()
    assert opt.result_count() == 1
    expect_optimal(opt, idx, 1, [1], [x])
    assert opt.solve_time_sec() >= 0.0 and opt.raw_status_string() == "Solved"


def case_model_supports_and_unsupported_constraint(lib):
    """What `supports_constraint` answers [REF src/MOI_wrapper.jl:912-952], and that an unsupported pair is refused."""
    opt = defaultoptimizer(lib)
    for S in (MOI.Interval, MOI.LessThan, MOI.GreaterThan, MOI.EqualTo):
        assert opt.supports_constraint(MOI.ScalarAffineFunction, S)
    for S in (MOI.Zeros, MOI.Nonnegatives, MOI.Nonpositives):
        assert opt.supports_constraint(MOI.VectorAffineFunction, S)
    assert not opt.supports_constraint(MOI.ScalarQuadraticFunction, MOI.LessThan)
    assert not opt.supports_constraint(MOI.VectorAffineFunction, MOI.Interval)
    m = MOI.Model()
    x = m.add_variable()
    try:
        m.add_constraint(MOI.ScalarQuadraticFunction([term(1.0, x, x)], [], 0.0), MOI.LessThan(1.0))
        raise AssertionError("a quadratic constraint must be refused")
    except MOI.UnsupportedConstraint:
        pass



# ---- the one-constraint problems of MathOptInterface's test_constraint_* / test_variable_* / test_solve_* families ----------
def _one_constraint(lib, sense, obj_coef, coefs, cset, obj, xval, dual):
    m = MOI.Model()
    x = m.add_variable()
    c = m.add_constraint(saf(coefs, [x] * len(coefs)), cset)
    m.set_objective_function(saf([obj_coef], [x]))
    m.set_objective_sense(sense)
    a = float(sum(coefs))
    lo = cset.lower if hasattr(cset, "lower") else (cset.value if isinstance(cset, MOI.EqualTo) else -np.inf)
    hi = cset.upper if hasattr(cset, "upper") else (cset.value if isinstance(cset, MOI.EqualTo) else np.inf)
    if isinstance(cset, MOI.GreaterThan):
        lo, hi = cset.lower, np.inf
    if isinstance(cset, MOI.LessThan):
        lo, hi = -np.inf, cset.upper
    bnds = sorted([lo / a, hi / a])
    ref, xr = lp_reference([obj_coef], bounds=[(None if not np.isfinite(bnds[0]) else bnds[0], None if not np.isfinite(bnds[1]) else bnds[1])],
                           maximize=(sense == MOI.MAX_SENSE))
    assert approx(ref, obj) and approx(xr, [xval])
    opt, idx = solve(lib, m)
    expect_optimal(opt, idx, obj, [xval], [x], [(c, dual)])
    return opt, idx, m, x, c


def case_constraint_ScalarAffineFunction_LessThan(lib):
    """max x  s.t.  2x <= 1   ->  0.5 at x = 0.5, dual -0.5."""
    _one_constraint(lib, MOI.MAX_SENSE, 1.0, [2.0], MOI.LessThan(1.0), 0.5, 0.5, -0.5)


def case_constraint_ScalarAffineFunction_GreaterThan(lib):
    """min x  s.t.  2x >= 1   ->  0.5 at x = 0.5, dual 0.5."""
    _one_constraint(lib, MOI.MIN_SENSE, 1.0, [2.0], MOI.GreaterThan(1.0), 0.5, 0.5, 0.5)


def case_constraint_ScalarAffineFunction_EqualTo(lib):
    """min x  s.t.  2x == 1   ->  0.5 at x = 0.5, dual 0.5."""
    _one_constraint(lib, MOI.MIN_SENSE, 1.0, [2.0], MOI.EqualTo(1.0), 0.5, 0.5, 0.5)


def case_constraint_ScalarAffineFunction_Interval(lib):
    """min 3x  s.t.  2x in [1, 4]   ->  1.5 at x = 0.5, dual 1.5."""
    _one_constraint(lib, MOI.MIN_SENSE, 3.0, [2.0], MOI.Interval(1.0, 4.0), 1.5, 0.5, 1.5)


def case_constraint_ScalarAffineFunction_duplicate(lib):
    """min x  s.t.  x + x >= 1 (the same variable twice in one function)   ->  0.5 at x = 0.5, dual 0.5."""
    _one_constraint(lib, MOI.MIN_SENSE, 1.0, [1.0, 1.0], MOI.GreaterThan(1.0), 0.5, 0.5, 0.5)


def case_variable_solve_with_lowerbound(lib):
    """min 2x  s.t.  x >= 1, x <= 2   ->  2 at x = 1; the lower bound's dual is 2, the upper bound's 0."""
    m = MOI.Model()
    x = m.add_variable()
    lb = m.add_constraint(saf([1], [x]), MOI.GreaterThan(1.0))
    ub = m.add_constraint(saf([1], [x]), MOI.LessThan(2.0))
    m.set_objective_function(saf([2.0], [x]))
    m.set_objective_sense(MOI.MIN_SENSE)
    ref, xr = lp_reference([2], bounds=[(1, 2)])
    assert approx(ref, 2) and approx(xr, [1])
    opt, idx = solve(lib, m)
    expect_optimal(opt, idx, 2, [1], [x], [(lb, 2), (ub, 0)])


def case_variable_solve_with_upperbound(lib):
    """max 2x  s.t.  x <= 1, x >= 0   ->  2 at x = 1; the upper bound's dual is -2, the lower bound's 0."""
    m = MOI.Model()
    x = m.add_variable()
    ub = m.add_constraint(saf([1], [x]), MOI.LessThan(1.0))
    lb = m.add_constraint(saf([1], [x]), MOI.GreaterThan(0.0))
    m.set_objective_function(saf([2.0], [x]))
    m.set_objective_sense(MOI.MAX_SENSE)
    ref, xr = lp_reference([2], bounds=[(0, 1)], maximize=True)
    assert approx(ref, 2) and approx(xr, [1])
    opt, idx = solve(lib, m)
    expect_optimal(opt, idx, 2, [1], [x], [(ub, -2), (lb, 0)])


def case_solve_VariableIndex_ConstraintDual_MIN_SENSE(lib):
    """min x  s.t.  x >= 1   ->  x = 1, dual 1."""
    _one_constraint(lib, MOI.MIN_SENSE, 1.0, [1.0], MOI.GreaterThan(1.0), 1.0, 1.0, 1.0)


def case_solve_VariableIndex_ConstraintDual_MAX_SENSE(lib):
    """max x  s.t.  x <= 1   ->  x = 1, dual -1."""
    _one_constraint(lib, MOI.MAX_SENSE, 1.0, [1.0], MOI.LessThan(1.0), 1.0, 1.0, -1.0)


def case_solve_optimize_twice(lib):
    """min x  s.t.  x >= 1, optimised twice in a row: the second call returns the same point (and, warm-started at the
    optimum, needs no more iterations than the first)."""
    opt, idx, m, x, c = _one_constraint(lib, MOI.MIN_SENSE, 1.0, [1.0], MOI.GreaterThan(1.0), 1.0, 1.0, 1.0)
    first = opt.variable_primal([idx[x]])
    opt.optimize()
    assert opt.termination_status() == MOI.OPTIMAL
    assert approx(opt.variable_primal([idx[x]]), first) and approx(opt.objective_value(), 1.0)


def case_objective_ObjectiveSense_MAX_and_MIN(lib):
    """The same feasible set [0, 1] x [0, 2] under max x + y (-> 3) and min x + y (-> 0); the wrapper takes the sense at copy_to
    [REF src/MOI_wrapper.jl:232, 497, 589: get and supports, no set], so the second sense is a second copy."""
    m = MOI.Model()
    x, y = m.add_variables(2)
    m.add_constraint(saf([1], [x]), MOI.Interval(0.0, 1.0))
    m.add_constraint(saf([1], [y]), MOI.Interval(0.0, 2.0))
    m.set_objective_function(saf([1.0, 1.0], [x, y]))
    m.set_objective_sense(MOI.MAX_SENSE)
    opt, idx = solve(lib, m)
    expect_optimal(opt, idx, 3, [1, 2], [x, y])
    m.set_objective_sense(MOI.MIN_SENSE)
    opt, idx = solve(lib, m)
    expect_optimal(opt, idx, 0, [0, 0], [x, y])

# ------------------------------------------------------------------------------------------------ conic linear (round 4)
# MathOptInterface's `test_conic_linear_*` family: the vector sets OSQP's wrapper supports (Zeros, Nonnegatives, Nonpositives over
# VectorAffineFunction [REF src/MOI_wrapper.jl:31-36]).  The suite's VectorOfVariables variants reach the wrapper through the
# bridge optimizer as the same affine functions with identity coefficients, which is how they are stated here.
def vaf(rows, constants):
    """rows: list of (row, coef, variable)"""
    return MOI.VectorAffineFunction([MOI.VectorAffineTerm(r, term(c, v)) for r, c, v in rows], list(constants))


def _lin1(m, x, y, z):
    c_eq = m.add_constraint(vaf([(1, 1.0, x), (1, 1.0, y), (1, 1.0, z), (2, 1.0, y), (2, 1.0, z)], [-3.0, -2.0]), MOI.Zeros(2))
    c_nn = m.add_constraint(vaf([(1, 1.0, x), (2, 1.0, y), (3, 1.0, z)], [0.0, 0.0, 0.0]), MOI.Nonnegatives(3))
    m.set_objective_function(saf([-3.0, -2.0, -4.0], [x, y, z]))
    m.set_objective_sense(MOI.MIN_SENSE)
    return c_eq, c_nn


def case_conic_linear_VectorAffineFunction(lib):
    """min -3x - 2y - 4z  s.t.  x + y + z = 3, y + z = 2 (Zeros), x, y, z >= 0 (Nonnegatives)  ->  -11 at (1, 0, 2); duals
    (-3, -1) on the equalities and (0, 2, 0) on the cone -- re-derived from linprog's marginals."""
    r = linprog([-3, -2, -4], A_eq=[[1, 1, 1], [0, 1, 1]], b_eq=[3, 2], bounds=[(0, None)] * 3, method="highs")
    assert r.status == 0 and approx(r.fun, -11) and approx(r.x, [1, 0, 2])
    assert approx(r.eqlin.marginals, [-3, -1]) and approx(r.lower.marginals, [0, 2, 0])
    m = MOI.Model()
    x, y, z = m.add_variables(3)
    c_eq, c_nn = _lin1(m, x, y, z)
    opt, idx = solve(lib, m)
    expect_optimal(opt, idx, -11, [1, 0, 2], [x, y, z], duals=[(c_eq, [-3, -1]), (c_nn, [0, 2, 0])])


def case_conic_linear_VectorOfVariables(lib):
    """The same problem as the suite states it with [x, y, z] in Nonnegatives(3) as a vector of variables (bridged: identity
    coefficients, zero constants), solved twice on one optimizer (the suite re-optimizes after reading the results)."""
    m = MOI.Model()
    x, y, z = m.add_variables(3)
    c_eq, c_nn = _lin1(m, x, y, z)
    opt, idx = solve(lib, m)
    opt.optimize()
    expect_optimal(opt, idx, -11, [1, 0, 2], [x, y, z], duals=[(c_eq, [-3, -1]), (c_nn, [0, 2, 0])])
    assert opt.result_count() == 1


def case_conic_linear_VectorAffineFunction_2(lib):
    """min 3x + 2y - 4z + 0s  s.t.  x - s = -4, y = -3, x + z = 12 (Zeros, one row each), y <= 0 (Nonpositives), z >= 0
    (Nonnegatives), s = 0 (Zeros)  ->  -82 at (x, y, z, s) = (-4, -3, 16, 0)."""
    r = linprog([3, 2, -4, 0], A_eq=[[1, 0, 0, -1], [0, 1, 0, 0], [1, 0, 1, 0], [0, 0, 0, 1]], b_eq=[-4, -3, 12, 0],
                bounds=[(None, None), (None, 0), (0, None), (None, None)], method="highs")
    assert r.status == 0 and approx(r.fun, -82) and approx(r.x, [-4, -3, 16, 0])
    m = MOI.Model()
    x, y, z, s = m.add_variables(4)
    m.add_constraint(vaf([(1, 1.0, x), (1, -1.0, s)], [4.0]), MOI.Zeros(1))
    m.add_constraint(vaf([(1, 1.0, y)], [3.0]), MOI.Zeros(1))
    m.add_constraint(vaf([(1, 1.0, x), (1, 1.0, z)], [-12.0]), MOI.Zeros(1))
    m.add_constraint(vaf([(1, 1.0, y)], [0.0]), MOI.Nonpositives(1))
    m.add_constraint(vaf([(1, 1.0, z)], [0.0]), MOI.Nonnegatives(1))
    m.add_constraint(vaf([(1, 1.0, s)], [0.0]), MOI.Zeros(1))
    m.set_objective_function(saf([3.0, 2.0, -4.0, 0.0], [x, y, z, s]))
    m.set_objective_sense(MOI.MIN_SENSE)
    opt, idx = solve(lib, m)
    expect_optimal(opt, idx, -82, [-4, -3, 16, 0], [x, y, z, s])


def _expect_infeasible(opt):
    assert opt.termination_status() in (MOI.INFEASIBLE, MOI.ALMOST_INFEASIBLE)
    assert opt.primal_status() in (MOI.NO_SOLUTION, MOI.UNKNOWN_RESULT_STATUS)
    assert opt.dual_status() in (MOI.INFEASIBILITY_CERTIFICATE, MOI.NEARLY_INFEASIBILITY_CERTIFICATE)


def case_conic_linear_INFEASIBLE(lib):
    """min 0  s.t.  x - 1 in Nonnegatives, x + 1 in Nonpositives: x >= 1 and x <= -1."""
    m = MOI.Model()
    (x,) = m.add_variables(1)
    m.add_constraint(vaf([(1, 1.0, x)], [-1.0]), MOI.Nonnegatives(1))
    m.add_constraint(vaf([(1, 1.0, x)], [1.0]), MOI.Nonpositives(1))
    opt, _ = solve(lib, m)
    _expect_infeasible(opt)


def case_conic_linear_INFEASIBLE_2(lib):
    """min 0  s.t.  x - 1 in Nonnegatives, [x] in Nonpositives: x >= 1 and x <= 0."""
    m = MOI.Model()
    (x,) = m.add_variables(1)
    m.add_constraint(vaf([(1, 1.0, x)], [-1.0]), MOI.Nonnegatives(1))
    m.add_constraint(vaf([(1, 1.0, x)], [0.0]), MOI.Nonpositives(1))
    opt, _ = solve(lib, m)
    _expect_infeasible(opt)


# --------------------------------------------------------------------------------------- quadratic objective edge cases (round 4)
def _qp_edge(lib, qterms, obj, xs):
    """x^2 + y^2 (+ what the caller adds) over x >= 1, y >= 2, the objective given as the list of quadratic terms
    (MathOptInterface's convention: 1/2 x'Qx, a diagonal term c x_i x_i contributes c / 2 x_i^2)."""
    m = MOI.Model()
    x, y = m.add_variables(2)
    m.add_constraint(saf([1], [x]), MOI.GreaterThan(1.0))
    m.add_constraint(saf([1], [y]), MOI.GreaterThan(2.0))
    m.set_objective_function(MOI.ScalarQuadraticFunction([term(c, (x, y)[a], (x, y)[b]) for c, a, b in qterms], [], 0.0))
    m.set_objective_sense(MOI.MIN_SENSE)
    opt, idx = solve(lib, m)
    expect_optimal(opt, idx, obj, xs, [x, y])


def case_objective_qp_ObjectiveFunction_edge_cases(lib):
    """x^2 + y^2 over x >= 1, y >= 2 stated four ways: plainly, with the diagonal term of x split in two, with the cross term
    as two entries that cancel, and with a cross term given twice (x^2 + xy + y^2: the bounds stay active, 1 + 2 + 4)."""
    cons = [dict(type="ineq", fun=lambda v: v[0] - 1), dict(type="ineq", fun=lambda v: v[1] - 2)]
    ref, xr = qp_reference(np.diag([2.0, 2.0]), np.zeros(2), cons, np.array([2.0, 3.0]))
    assert approx(ref, 5) and approx(xr, [1, 2])
    ref2, xr2 = qp_reference(np.array([[2.0, 1.0], [1.0, 2.0]]), np.zeros(2), cons, np.array([2.0, 3.0]))
    assert approx(ref2, 7) and approx(xr2, [1, 2])
    _qp_edge(lib, [(2.0, 0, 0), (2.0, 1, 1)], 5, [1, 2])
    _qp_edge(lib, [(1.0, 0, 0), (1.0, 0, 0), (2.0, 1, 1)], 5, [1, 2])
    _qp_edge(lib, [(2.0, 0, 0), (0.25, 0, 1), (-0.25, 1, 0), (2.0, 1, 1)], 5, [1, 2])
    _qp_edge(lib, [(2.0, 0, 0), (0.5, 0, 1), (0.5, 1, 0), (2.0, 1, 1)], 7, [1, 2])


def case_objective_qp_ObjectiveFunction_zero_ofdiag(lib):
    """The same objective with an EXPLICIT zero off-diagonal term: the wrapper keeps the zero in the pattern of P [REF
    src/MOI_wrapper.jl:151-182] and the answer is unchanged."""
    _qp_edge(lib, [(2.0, 0, 0), (0.0, 0, 1), (2.0, 1, 1)], 5, [1, 2])


ALL = [f for name, f in sorted(globals().items()) if name.startswith("case_") and callable(f)]
