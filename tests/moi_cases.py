"""The reference's MathOptInterface tests that pin results [REF test/MOI_wrapper.jl:280-790], restated over the Python mirror of
the MOI face (osqp.jl_amd/moi.py).  Every case takes the loaded C-ABI library: the CPU oracle in the CPU suite, the HIP
engine in the GPU suite.  `MOI.Test.runtests` itself (the generic conformance suite of the Julia package) has no counterpart
here; the cases below are the ones the reference wrote by hand against its own wrapper."""
import copy

import numpy as np

from osqp_jl_amd import moi as MOI

ATOL = RTOL = 1e-4  # const config = MOI.Test.Config(atol = 1e-4, rtol = 1e-4) [REF test/MOI_wrapper.jl:24]


def defaultoptimizer(lib):  # [REF test/MOI_wrapper.jl:41-49]
    opt = MOI.Optimizer(lib)
    opt.set_silent(True)
    opt.set_raw("EpsAbs", 1e-8)
    opt.set_raw("EpsRel", 1e-16)
    opt.set_raw("MaxIter", 10000)
    opt.set_raw("AdaptiveRhoInterval", 25)  # required for deterministic behaviour
    return opt


def approx(a, b, atol=ATOL, rtol=RTOL):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return np.all(np.abs(a - b) <= np.maximum(atol, rtol * np.maximum(np.abs(a), np.abs(b))))


def term(c, x, y=None):
    return MOI.ScalarAffineTerm(c, x) if y is None else MOI.ScalarQuadraticTerm(c, x, y)


class ModelFace:
    """The modification calls of the reference's `modfun(m)` closures, spelled once for both a Model and an Optimizer."""

    def __init__(self, target, idxmap):
        self.t, self.idxmap = target, idxmap
        self.is_opt = isinstance(target, MOI.Optimizer)

    def map(self, idx):  # mapfrommodel [REF test/MOI_wrapper.jl:345-357]
        return self.idxmap[idx] if self.is_opt else idx

    def set_objective(self, f):
        self.t.set_objective_function(f)

    def modify_objective(self, change):
        self.t.modify_objective(change)

    def set_constraint_function(self, ci, f):
        self.t.set_constraint_function(self.map(ci), f)

    def set_constraint_set(self, ci, s):
        self.t.set_constraint_set(self.map(ci), s)

    def modify_constraint(self, ci, change):
        self.t.modify_constraint(self.map(ci), change)


def check_modification(modfun, model, optimizer, idxmap, cleanoptimizer, atol=ATOL, rtol=RTOL):
    """_test_optimizer_modification [REF test/MOI_wrapper.jl:207-267]: apply modfun to the model and to the live optimizer,
    copy the model into a clean optimizer, solve both, compare."""
    modfun(ModelFace(model, idxmap))
    modfun(ModelFace(optimizer, idxmap))
    cleanidxmap = cleanoptimizer.copy_to(model)
    assert cleanidxmap.var_map == idxmap.var_map and cleanidxmap.con_map == idxmap.con_map
    optimizer.optimize()
    cleanoptimizer.optimize()
    assert optimizer.termination_status() == cleanoptimizer.termination_status()
    assert optimizer.primal_status() == cleanoptimizer.primal_status()
    assert approx(optimizer.objective_value(), cleanoptimizer.objective_value(), atol, rtol)
    if optimizer.primal_status() == MOI.FEASIBLE_POINT:
        for v in model.variable_indices():
            assert approx(optimizer.variable_primal(idxmap[v]), cleanoptimizer.variable_primal(idxmap[v]), atol, rtol)
    assert optimizer.dual_status() == cleanoptimizer.dual_status()
    if optimizer.dual_status() == MOI.FEASIBLE_POINT:
        for ci in model.constraints:
            assert approx(optimizer.constraint_dual(idxmap[ci]), cleanoptimizer.constraint_dual(idxmap[ci]), atol, rtol)


def case_problem_modification_after_copy_to(lib):  # [REF test/MOI_wrapper.jl:280-519]
    # min -x  s.t.  x + y <= 1,  x, y >= 0   (written with redundant zero terms, as the reference does)
    model = MOI.Model()
    v = model.add_variables(2)
    x, y = v
    cf = MOI.ScalarAffineFunction([term(0.0, x), term(0.0, y), term(1.0, x), term(1.0, y), term(0.0, x), term(0.0, y)], 0.0)
    c = model.add_constraint(cf, MOI.Interval(-MOI.INF, 1.0))
    vc1 = model.add_constraint(MOI.ScalarAffineFunction([term(1.0, x)], 0.0), MOI.Interval(0.0, MOI.INF))
    vc2 = model.add_constraint(MOI.ScalarAffineFunction([term(1.0, y)], 0.0), MOI.Interval(0.0, MOI.INF))
    objf = MOI.ScalarAffineFunction([term(0.0, x), term(0.0, y), term(-1.0, x), term(0.0, y), term(0.0, x), term(0.0, y)], 0.0)
    model.set_objective_function(objf)
    model.set_objective_sense(MOI.MIN_SENSE)

    optimizer = defaultoptimizer(lib)
    idxmap = optimizer.copy_to(model)
    assert optimizer.get_objective_sense() == MOI.MIN_SENSE
    assert optimizer.number_of_variables() == 2
    assert optimizer.list_of_variable_indices() == [MOI.VariableIndex(1), MOI.VariableIndex(2)]
    assert optimizer.is_valid(MOI.VariableIndex(2)) and not optimizer.is_valid(MOI.VariableIndex(3))
    optimizer.optimize()
    assert optimizer.termination_status() == MOI.OPTIMAL and optimizer.primal_status() == MOI.FEASIBLE_POINT
    assert approx(optimizer.objective_value(), -1)
    assert approx(optimizer.variable_primal([idxmap[x], idxmap[y]]), [1, 0])
    assert optimizer.dual_status() == MOI.FEASIBLE_POINT
    assert approx(optimizer.constraint_dual(idxmap[c]), -1)
    assert approx(optimizer.constraint_dual(idxmap[vc1]), 0)
    assert approx(optimizer.constraint_dual(idxmap[vc2]), 1)

    # default warm start
    itercold = optimizer.results.info.iter
    optimizer.optimize()
    assert optimizer.results.info.iter < itercold

    # solving again gives the same answer after zeroing the warm start
    for vi in idxmap.var_map.values():
        optimizer.set_primal_start(vi, 0.0)
    for ci in idxmap.con_map.values():
        optimizer.set_dual_start(ci, -0.0)
    check_modification(lambda m: None, model, optimizer, idxmap, defaultoptimizer(lib), atol=0.0, rtol=0.0)

    # change objective to min -2y
    check_modification(lambda m: m.set_objective(MOI.ScalarAffineFunction([term(-2.0, m.map(y))], 0.0)), model, optimizer, idxmap,
                       defaultoptimizer(lib))
    # add a constant to the objective
    before = optimizer.objective_value()
    check_modification(lambda m: m.modify_objective(MOI.ScalarConstantChange(1.5)), model, optimizer, idxmap, defaultoptimizer(lib))
    assert abs(optimizer.objective_value() - (before + 1.5)) <= 1e-8
    # change objective to min -y using ScalarCoefficientChange
    check_modification(lambda m: m.modify_objective(MOI.ScalarCoefficientChange(m.map(y), -1.0)), model, optimizer, idxmap,
                       defaultoptimizer(lib))
    assert abs(optimizer.objective_value() - (0.5 * before + 1.5)) <= 1e-8
    # change x + y <= 1 to x + 2 y + 0.5 <= 1
    check_modification(lambda m: m.set_constraint_function(
        c, MOI.ScalarAffineFunction([term(1.0, m.map(x)), term(1.0, m.map(x)), term(1.0, m.map(y))], 0.5)),
        model, optimizer, idxmap, defaultoptimizer(lib))
    # ... the reference writes term.([1, 1, 1], [x, x, y]) = 2x + y; back to x + y <= 1 with ScalarCoefficientChange on y only
    # leaves 2x + y + 0.5 <= 1: both sides apply the same change, which is what the test compares
    check_modification(lambda m: m.modify_constraint(c, MOI.ScalarCoefficientChange(m.map(y), 1.0)), model, optimizer, idxmap,
                       defaultoptimizer(lib))

    # flip the feasible set around and minimise +x
    def flip(m):
        m.set_objective(MOI.ScalarAffineFunction([term(1.0, m.map(x))], 0.0))
        m.set_constraint_function(c, MOI.ScalarAffineFunction([term(1.0, m.map(x)), term(1.0, m.map(y))], 0.0))
        m.set_constraint_set(c, MOI.Interval(-1.0, MOI.INF))
        m.set_constraint_set(vc1, MOI.Interval(-MOI.INF, 0.0))
        m.set_constraint_set(vc2, MOI.Interval(-MOI.INF, 0.0))
    check_modification(flip, model, optimizer, idxmap, defaultoptimizer(lib))

    def testflipped():
        assert optimizer.termination_status() == MOI.OPTIMAL and optimizer.primal_status() == MOI.FEASIBLE_POINT
        assert approx(optimizer.objective_value(), -1)
        assert approx(optimizer.variable_primal([idxmap[x], idxmap[y]]), [-1, 0])
        assert optimizer.dual_status() == MOI.FEASIBLE_POINT
        assert approx(optimizer.constraint_dual(idxmap[c]), 1)
        assert approx(optimizer.constraint_dual(idxmap[vc1]), 0)
        assert approx(optimizer.constraint_dual(idxmap[vc2]), -1)
    testflipped()
    # update settings
    assert optimizer.results.info.status_polish == 0
    optimizer.set_raw("Polish", True)
    optimizer.optimize()
    assert optimizer.results.info.status_polish == 1
    testflipped()


def case_vector_problem_modification_after_copy_to(lib, trials=25):  # [REF test/MOI_wrapper.jl:521-620] (basic.jl's QP)
    model = MOI.Model()
    x = model.add_variables(2)
    P11, q = 11.0, [3.0, 4.0]
    u = np.array([0.0, 0.0, -15, 100, 80])
    Adense = np.array([[-1.0, 0], [0, -1], [-1, -3], [2, 5], [3, 4]])
    I, J = np.nonzero(Adense.T)[1], np.nonzero(Adense.T)[0]  # findnz order: column by column
    coeffs = Adense[I, J]
    objf = MOI.ScalarQuadraticFunction([term(2 * P11, x[0], x[0]), term(0.0, x[0], x[1])], [term(q[0], x[0]), term(q[1], x[1])], 0.0)
    model.set_objective_function(objf)
    model.set_objective_sense(MOI.MIN_SENSE)

    def vfun(cs, consts):
        return MOI.VectorAffineFunction([MOI.VectorAffineTerm(int(i) + 1, term(float(cv), x[int(j)])) for i, j, cv in zip(I, J, cs)],
                                        [float(t) for t in consts])
    c = model.add_constraint(vfun(coeffs, -u), MOI.Nonpositives(len(u)))
    optimizer = defaultoptimizer(lib)
    idxmap = optimizer.copy_to(model)
    optimizer.optimize()
    assert optimizer.termination_status() == MOI.OPTIMAL and optimizer.primal_status() == MOI.FEASIBLE_POINT
    assert approx(optimizer.objective_value(), 20.0)
    assert approx(optimizer.variable_primal([idxmap[x[0]], idxmap[x[1]]]), [0.0, 5.0])
    assert optimizer.dual_status() == MOI.FEASIBLE_POINT
    assert approx(optimizer.constraint_dual(idxmap[c]), -np.array([1.666666666666, 0.0, 1.3333333, 0.0, 0.0]))
    # random modifications of the constraint function (Julia's MersenneTwister stream is not portable: numpy's here)
    rng = np.random.default_rng(1234)
    for _ in range(trials):
        newcoeffs = coeffs.copy()
        newcoeffs[rng.integers(len(newcoeffs))] = 0
        newconst = np.round(5 * (rng.random(len(u)) - 0.5), 2)
        check_modification(lambda m: m.set_constraint_function(c, vfun(newcoeffs, newconst)), model, optimizer, idxmap,
                           defaultoptimizer(lib), atol=np.inf, rtol=1e-4)


def case_warm_starting(lib):  # [REF test/MOI_wrapper.jl:622-692]
    l = [1.0, 0, 0]
    u = [1.0, 0.7, 0.7]
    model = MOI.Model()
    optimizer = defaultoptimizer(lib)
    x = model.add_variables(2)
    # 1.0 x1 + 1.0 x2 + 2.0 x1^2 + 1.0 x1 x2 + 1.0 x2^2  (MOI stores 2 * the coefficient of a squared term)
    model.set_objective_function(MOI.ScalarQuadraticFunction(
        [term(4.0, x[0], x[0]), term(1.0, x[0], x[1]), term(2.0, x[1], x[1])], [term(1.0, x[0]), term(1.0, x[1])], 0.0))
    model.set_objective_sense(MOI.MIN_SENSE)

    def rows(sign):
        return [MOI.VectorAffineTerm(1, term(sign, x[0])), MOI.VectorAffineTerm(1, term(sign, x[1])),
                MOI.VectorAffineTerm(2, term(sign, x[0])), MOI.VectorAffineTerm(3, term(sign, x[1]))]
    model.add_constraint(MOI.VectorAffineFunction(rows(-1.0), list(u)), MOI.Nonnegatives(3))
    con1 = model.add_constraint(MOI.VectorAffineFunction(rows(1.0), [-t for t in u]), MOI.Nonpositives(3))
    con2 = model.add_constraint(MOI.VectorAffineFunction(rows(1.0), [-t for t in l]), MOI.Nonnegatives(3))
    optimizer.empty()
    idxmap = optimizer.copy_to(model)
    optimizer.optimize()
    itercold = optimizer.results.info.iter
    x_sol = optimizer.variable_primal([idxmap[x[0]], idxmap[x[1]]])
    y_c1, y_c2 = optimizer.constraint_dual(idxmap[con1]), optimizer.constraint_dual(idxmap[con2])
    r1, r2 = optimizer.constraint_rows(idxmap[con1]), optimizer.constraint_rows(idxmap[con2])
    for vi, val in zip(x, x_sol):
        model.set_primal_start(vi, val)
    model.set_dual_start(con1, y_c1)
    model.set_dual_start(con2, y_c2)
    optimizer.empty()
    idxmap = optimizer.copy_to(model)
    assert np.array_equal(optimizer.warmstartcache.x.data, x_sol)
    assert np.array_equal(optimizer.warmstartcache.y.data[r1.start:r1.stop], -y_c1)
    assert np.array_equal(optimizer.warmstartcache.y.data[r2.start:r2.stop], -y_c2)
    # and the solve started from them needs fewer iterations than the cold one
    optimizer.optimize()
    assert optimizer.termination_status() == MOI.OPTIMAL and optimizer.results.info.iter < itercold


def case_vector_equality_constraint(lib, trials=10):  # [REF test/MOI_wrapper.jl:694-790]
    # minimise ||A x - b||^2 = x'A'A x - (2 A'b)'x + b'b  subject to  C x = d ; closed form through pseudo-inverses
    n, m = 8, 2
    rng = np.random.default_rng(1234)

    def data():
        A, b, C, d = rng.random((n, n)), rng.random(n), rng.random((m, n)), rng.random(m)
        Cp = np.linalg.pinv(C)
        Q = np.eye(n) - Cp @ C
        expected = Q @ (np.linalg.pinv(A @ Q) @ (b - A @ Cp @ d)) + Cp @ d
        assert np.allclose(C @ expected, d, atol=1e-10)
        return A, b, C, d, np.triu(A.T @ A), -2 * A.T @ b, float(b @ b), expected

    def objective(P, q, r, x):
        I, J = np.nonzero(P)
        quad = [term(2 * P[i, j], x[i], x[j]) for i, j in zip(I, J)]
        # the reference passes Symmetric(triu(A'A)) and doubles EVERY stored coefficient: an off-diagonal entry of the upper
        # triangle then stands for 2 * P_ij x_i x_j = P_ij x_i x_j + P_ji x_j x_i, a diagonal one for (2 P_ii) / 2 x_i^2
        return MOI.ScalarQuadraticFunction(quad, [term(float(q[i]), x[i]) for i in range(n)], r)

    def constraint(C, d, x):
        I, J = np.nonzero(C.T)[1], np.nonzero(C.T)[0]
        return MOI.VectorAffineFunction([MOI.VectorAffineTerm(int(i) + 1, term(float(C[i, j]), x[int(j)])) for i, j in zip(I, J)],
                                        [-float(t) for t in d])

    def check(optimizer, xs, A, b, expected):
        assert optimizer.termination_status() == MOI.OPTIMAL and optimizer.primal_status() == MOI.FEASIBLE_POINT
        assert np.allclose(optimizer.variable_primal(xs), expected, atol=1e-4)
        assert abs(optimizer.objective_value() - np.linalg.norm(A @ expected - b) ** 2) <= 1e-4

    A, b, C, d, P, q, r, expected = data()
    model = MOI.Model()
    x = model.add_variables(n)
    model.set_objective_function(objective(P, q, r, x))
    model.set_objective_sense(MOI.MIN_SENSE)
    c = model.add_constraint(constraint(C, d, x), MOI.Zeros(m))
    optimizer = defaultoptimizer(lib)
    idxmap = optimizer.copy_to(model)
    optimizer.optimize()
    xs = [idxmap[xi] for xi in x]
    check(optimizer, xs, A, b, expected)
    for _ in range(trials):
        A, b, C, d, P, q, r, expected = data()
        optimizer.set_objective_function(objective(P, q, r, xs))
        optimizer.set_constraint_function(idxmap[c], constraint(C, d, xs))
        optimizer.set_constraint_set(idxmap[c], MOI.Zeros(m))  # no-op, but allowed
        optimizer.optimize()
        check(optimizer, xs, A, b, expected)


def case_raw_solver_and_statuses(lib):  # [REF test/MOI_wrapper.jl:792-812, src/MOI_wrapper.jl:617-687]
    optimizer = defaultoptimizer(lib)
    assert optimizer.raw_solver().isempty and optimizer.solver_name == "OSQP"
    assert optimizer.termination_status() == MOI.OPTIMIZE_NOT_CALLED and optimizer.result_count() == 0
    model = MOI.Model()
    x = model.add_variable()
    model.add_constraint(MOI.ScalarAffineFunction([term(1.0, x)], 0.0), MOI.GreaterThan(2.0))
    model.set_objective_sense(MOI.MIN_SENSE)
    model.set_objective_function(MOI.ScalarAffineFunction([term(1.0, x)], 0.0))
    optimizer.copy_to(model)
    assert not optimizer.raw_solver().isempty
    optimizer.optimize()
    assert optimizer.termination_status() == MOI.OPTIMAL and approx(optimizer.objective_value(), 2.0)
    # maximisation flips P, q, the constant and the reported objective
    model.set_objective_sense(MOI.MAX_SENSE)
    model.set_objective_function(MOI.ScalarAffineFunction([term(-1.0, x)], 3.0))
    optimizer.copy_to(model)
    optimizer.optimize()
    assert approx(optimizer.objective_value(), 1.0) and approx(optimizer.variable_primal(MOI.VariableIndex(1)), 2.0)
    # an infeasible model: x >= 2 and x <= 1  -> INFEASIBLE with a certificate in the reference's (opposite) sign convention
    ci2 = model.add_constraint(MOI.ScalarAffineFunction([term(1.0, x)], 0.0), MOI.LessThan(1.0))
    idxmap = optimizer.copy_to(model)
    optimizer.optimize()
    assert optimizer.termination_status() == MOI.INFEASIBLE
    assert optimizer.primal_status() == MOI.NO_SOLUTION and optimizer.dual_status() == MOI.INFEASIBILITY_CERTIFICATE
    assert np.isfinite(optimizer.constraint_dual(idxmap[ci2]))
    # a setting that cannot be updated after copy_to is refused [REF src/MOI_wrapper.jl:545-555]
    try:
        optimizer.set_raw("Sigma", 1e-5)
        raise AssertionError("expected SetAttributeNotAllowed")
    except MOI.SetAttributeNotAllowed:
        pass
    # an entry outside the sparsity pattern given at copy_to is refused
    try:
        optimizer.set_objective_function(MOI.ScalarQuadraticFunction([term(1.0, MOI.VariableIndex(1), MOI.VariableIndex(1))], [], 0.0))
        raise AssertionError("expected SetAttributeNotAllowed")
    except MOI.SetAttributeNotAllowed:
        pass


ALL = [case_problem_modification_after_copy_to, case_vector_problem_modification_after_copy_to, case_warm_starting,
       case_vector_equality_constraint, case_raw_solver_and_statuses]
