/*
 * oracle/oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, single thread) of the algorithm behind the 30
 * `osqp_*` symbols that osqp/OSQP.jl binds.  The arithmetic of that path lives
 * in the third-party binary OSQP_jll, pinned "=0.6.2, ~0.600.200"
 * [REF Project.toml:13,18] (libosqp v0.6.2 with its bundled QDLDL and AMD);
 * its source is NOT in /root/reference and not in this image.  This file set
 * restates the published algorithm (Stellato, Banjac, Goulart, Bemporad, Boyd,
 * "OSQP: an operator splitting solver for quadratic programs", Math. Prog.
 * Comp. 2020, sections 3-5) behind the reference's own call sites
 * [REF src/interface.jl:147..709, src/types.jl:139].
 *
 * PARITY STATUS: pinned against every known answer the reference's tests hold
 * for this path (tests/test_oracle_golden.py: test/basic.jl, polishing.jl incl.
 * the JLD2 fixture, non_convex.jl, dual_infeasibility.jl,
 * primal_infeasibility.jl, unconstrained.jl, warm_start.jl, feasibility.jl).
 * Bit-level / iteration-count parity with libosqp itself is UNPINNED: libosqp
 * cannot be built or run here.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product (osqp.jl_amd/csrc) never links or calls it.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include "../include/osqp_amd.h"
#include <stdint.h>

/* algorithm constants of libosqp v0.6.x (public constants.h; restated) */
#define RHO_MIN 1e-6
#define RHO_MAX 1e6
#define RHO_EQ_OVER_RHO_INEQ 1e3
#define RHO_TOL 1e-4
#define MIN_SCALING 1e-4
#define MAX_SCALING 1e4
#define ORACLE_NAN (0.0 / 0.0)

/* ---- sparse helpers (linalg.c) ---- */
csc  *csc_alloc(c_int m, c_int n, c_int nzmax);
csc  *csc_copy(const csc *A);
void  csc_free(csc *A);
void  mat_vec(const csc *A, const c_float *x, c_float *y, int plus_eq);            /* y (=|+=|-=) A x   */
void  mat_tpose_vec(const csc *A, const c_float *x, c_float *y, int plus_eq, int skip_diag); /* A' x */
c_float quad_form(const csc *P, const c_float *x);                                 /* 1/2 x' P x, P upper */
void  mat_inf_norm_cols(const csc *M, c_float *E);
void  mat_inf_norm_rows(const csc *M, c_float *E);
void  mat_inf_norm_cols_sym_triu(const csc *M, c_float *E);
void  mat_premult_diag(csc *A, const c_float *d);
void  mat_postmult_diag(csc *A, const c_float *d);
void  mat_mult_scalar(csc *A, c_float sc);
c_float vec_norm_inf(const c_float *v, c_int n);
c_float vec_scaled_norm_inf(const c_float *S, const c_float *v, c_int n);
c_float vec_prod(const c_float *a, const c_float *b, c_int n);

/* ---- direct KKT back-end (ldl.c) ---- */
typedef struct direct_solver direct_solver;
/* K = [P + sigma I, A'; A, -diag(rho_inv)]  (rho_inv == NULL: -sigma I, the polish form).
 * Returns NULL on failure; *err = 4 numeric failure, 5 wrong inertia (non-convex). */
direct_solver *direct_init(const csc *P, const csc *A, c_float sigma, const c_float *rho_inv, int polish, int *err);
void  direct_solve(direct_solver *s, c_float *b);       /* in place; non-polish form also does the z~ fix-up */
int   direct_update_matrices(direct_solver *s, const csc *P, const csc *A);
int   direct_update_rho(direct_solver *s, const c_float *rho_inv);
void  direct_free(direct_solver *s);
c_int direct_nnzL(const direct_solver *s);

/* ---- indirect KKT back-end (pcg.c) ---- */
typedef struct pcg_solver pcg_solver;
pcg_solver *pcg_init(const csc *P, const csc *A, c_float sigma, const c_float *rho_vec);
/* rhs/sol as for direct_solve (length n+m, in place).  tol_abs = absolute
 * inf-norm tolerance on the reduced-system residual.  Returns CG iterations, <0 if
 * negative curvature was met (P + sigma I + A' rho A not positive definite). */
c_int pcg_solve(pcg_solver *s, c_float *b, c_float tol_abs);
void  pcg_update_matrices(pcg_solver *s, const csc *P, const csc *A);
void  pcg_update_rho(pcg_solver *s, const c_float *rho_vec);
void  pcg_free(pcg_solver *s);
c_int pcg_total_iters(const pcg_solver *s);
void  pcg_set_guess(pcg_solver *s, const c_float *x);

/* ---- counter-based problem generators (gen.c) ---- */
uint64_t oracle_rnd(uint64_t seed, uint64_t stream, uint64_t idx);
double   oracle_u01(uint64_t seed, uint64_t stream, uint64_t idx);
double   oracle_gauss(uint64_t seed, uint64_t stream, uint64_t idx);

#endif
