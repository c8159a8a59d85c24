"""Mirror of the reference's constants [REF src/constants.jl:1-44]."""

QDLDL_SOLVER = 0  # [REF src/constants.jl:1]
MKL_PARDISO_SOLVER = 1  # [REF src/constants.jl:2]
AMD_PCG_SOLVER = 2  # extension (include/osqp_amd.h)
AMD_DIRECT_SOLVER = 3  # extension

OSQP_INFTY = 1e30  # [REF src/constants.jl:5]

# [REF src/constants.jl:9-21]
status_map = {
    4: "Dual_infeasible_inaccurate",
    3: "Primal_infeasible_inaccurate",
    2: "Solved_inaccurate",
    1: "Solved",
    -2: "Max_iter_reached",
    -3: "Primal_infeasible",
    -4: "Dual_infeasible",
    -5: "Interrupted",
    -6: "Time_limit_reached",
    -7: "Non_convex",
    -10: "Unsolved",
}

SOLUTION_PRESENT = ["Solved_inaccurate", "Solved", "Max_iter_reached"]  # [REF src/constants.jl:23]

UPDATABLE_DATA = ["q", "l", "u", "Px", "Px_idx", "Ax", "Ax_idx"]  # [REF src/constants.jl:26]

# [REF src/constants.jl:29-44] (scaled_termination is absent there too)
UPDATABLE_SETTINGS = [
    "max_iter",
    "eps_abs",
    "eps_rel",
    "eps_prim_inf",
    "eps_dual_inf",
    "time_limit",
    "rho",
    "alpha",
    "delta",
    "polish",
    "polish_refine_iter",
    "verbose",
    "check_termination",
    "warm_start",
]
