/*
 * osqp_amd.h -- C ABI of the MI355X-native OSQP ADMM engine (libosqp_amd.so).
 *
 * This header is the drop-in boundary.  Part 1 declares exactly the symbols and
 * struct layouts that the reference wrapper osqp/OSQP.jl v0.8.1 binds with
 * `ccall` (every entry cites the reference call site as [REF file:line], paths
 * relative to the reference checkout).  An unmodified OSQP.jl pointed at
 * libosqp_amd.so instead of OSQP_jll's libosqp therefore keeps working
 * (see INTEGRATION.md).  Part 2 declares the extension entry points that have
 * no counterpart in the reference (device-resident problem generation, the
 * batched small-QP path, introspection for measurement).
 *
 * Conventions (same as the reference's FFI): plain pointers and sizes, no C++
 * or torch types; c_int is 64-bit [REF src/types.jl:5-9]; c_float is double;
 * return value 0 means success, anything else is an error that the Julia side
 * turns into `error(...)` [REF src/interface.jl:157-159].
 *
 * All pointers passed IN are borrowed for the duration of the call only (the
 * Julia side holds them under `@preserve` [REF src/interface.jl:132-155]); the
 * library deep-copies.  Everything reachable from an OSQPWorkspace* is owned
 * by the library and released by osqp_cleanup().
 */
#ifndef OSQP_AMD_H
#define OSQP_AMD_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef long long c_int;   /* [REF src/types.jl:5-9]  Cc_int = Clonglong */
typedef double    c_float; /* Cdouble everywhere in [REF src/types.jl]   */

/* ------------------------------------------------------------------------- */
/* Part 1a: constants  [REF src/constants.jl:1-21]                            */
/* ------------------------------------------------------------------------- */
#define OSQP_INFTY 1e30 /* [REF src/constants.jl:5] */

enum osqp_linsys_solver_type {
  QDLDL_SOLVER       = 0, /* [REF src/constants.jl:1]  direct LDL^T (default; "auto" here: falls back to PCG when the factor cannot fit) */
  MKL_PARDISO_SOLVER = 1, /* [REF src/constants.jl:2]  accepted, served by the same direct LDL^T back-end */
  AMD_PCG_SOLVER     = 2, /* extension: force the indirect (preconditioned CG) back-end */
  AMD_DIRECT_SOLVER  = 3  /* extension: force the direct back-end (never fall back) */
};

/* status_val codes [REF src/constants.jl:9-21] */
#define OSQP_DUAL_INFEASIBLE_INACCURATE    (4)
#define OSQP_PRIMAL_INFEASIBLE_INACCURATE  (3)
#define OSQP_SOLVED_INACCURATE             (2)
#define OSQP_SOLVED                        (1)
#define OSQP_MAX_ITER_REACHED             (-2)
#define OSQP_PRIMAL_INFEASIBLE            (-3)
#define OSQP_DUAL_INFEASIBLE              (-4)
#define OSQP_SIGINT                       (-5)
#define OSQP_TIME_LIMIT_REACHED           (-6)
#define OSQP_NON_CVX                      (-7)
#define OSQP_UNSOLVED                    (-10)

/* ------------------------------------------------------------------------- */
/* Part 1b: struct layouts read across the boundary                           */
/* ------------------------------------------------------------------------- */

/* Compressed-sparse-column matrix, 56 bytes. [REF src/types.jl:11-19]
 * nz == -1 marks compressed-column form [REF src/types.jl:46]; indices are
 * 0-based [REF src/types.jl:39-43]. */
typedef struct {
  c_int    nzmax;
  c_int    m;
  c_int    n;
  c_int   *p;
  c_int   *i;
  c_float *x;
  c_int    nz;
} csc;

/* 56 bytes. [REF src/types.jl:101-109] */
typedef struct {
  c_int    n;
  c_int    m;
  csc     *P; /* upper triangle only [REF src/interface.jl:102-104] */
  csc     *A;
  c_float *q;
  c_float *l;
  c_float *u;
} OSQPData;

/* 176 bytes, linsys_solver is a 32-bit enum followed by 4 bytes of padding.
 * [REF src/types.jl:111-134] */
typedef struct {
  c_float rho;
  c_float sigma;
  c_int   scaling;
  c_int   adaptive_rho;
  c_int   adaptive_rho_interval;
  c_float adaptive_rho_tolerance;
  c_float adaptive_rho_fraction;
  c_int   max_iter;
  c_float eps_abs;
  c_float eps_rel;
  c_float eps_prim_inf;
  c_float eps_dual_inf;
  c_float alpha;
  int     linsys_solver; /* enum osqp_linsys_solver_type */
  c_float delta;
  c_int   polish;
  c_int   polish_refine_iter;
  c_int   verbose;
  c_int   scaled_termination;
  c_int   check_termination;
  c_int   warm_start;
  c_float time_limit;
} OSQPSettings;

/* 136 bytes. [REF src/types.jl:81-99] */
typedef struct {
  c_int   iter;
  char    status[32];
  c_int   status_val;
  c_int   status_polish;
  c_float obj_val;
  c_float pri_res;
  c_float dua_res;
  c_float setup_time;
  c_float solve_time;
  c_float update_time;
  c_float polish_time;
  c_float run_time;
  c_int   rho_updates;
  c_float rho_estimate;
} OSQPInfo;

/* 16 bytes. [REF src/types.jl:74-77] */
typedef struct {
  c_float *x;
  c_float *y;
} OSQPSolution;

/* 30-field mirror. [REF src/types.jl:173-217]
 * The Julia side dereferences `data`, `solution`, `info`, `delta_y`, `delta_x`
 * as HOST pointers after osqp_solve [REF src/interface.jl:176-205, 744-746].
 * The iterates live in HBM; those five are host mirrors refreshed before
 * osqp_solve returns.  `rho_vec` .. `E_temp` other than delta_x/delta_y are
 * NULL in libosqp_amd.so (device-resident; no reference code reads them). */
typedef struct OSQPWorkspace {
  OSQPData     *data;          /*   0 */
  void         *linsys_solver; /*   8 */
  void         *pol;           /*  16 */
  c_float      *rho_vec;       /*  24 */
  c_float      *rho_inv_vec;   /*  32 */
  c_int        *constr_type;   /*  40 */
  c_float      *x;             /*  48 */
  c_float      *y;             /*  56 */
  c_float      *z;             /*  64 */
  c_float      *xz_tilde;      /*  72 */
  c_float      *x_prev;        /*  80 */
  c_float      *z_prev;        /*  88 */
  c_float      *Ax;            /*  96 */
  c_float      *Px;            /* 104 */
  c_float      *Aty;           /* 112 */
  c_float      *delta_y;       /* 120  host mirror: primal-infeasibility certificate */
  c_float      *Atdelta_y;     /* 128 */
  c_float      *delta_x;       /* 136  host mirror: dual-infeasibility certificate */
  c_float      *Pdelta_x;      /* 144 */
  c_float      *Adelta_x;      /* 152 */
  c_float      *D_temp;        /* 160 */
  c_float      *D_temp_A;      /* 168 */
  c_float      *E_temp;        /* 176 */
  OSQPSettings *settings;      /* 184 */
  void         *scaling;       /* 192 */
  OSQPSolution *solution;      /* 200  host mirror */
  OSQPInfo     *info;          /* 208  host mirror */
  void         *timer;         /* 216 */
  c_int         first_run;     /* 224 */
  c_int         summary_printed; /* 232 */
  void         *impl;          /* 240  library-private (engine handle) */
} OSQPWorkspace;

/* Layout contract, checked at compile time by every translation unit that includes this header (the library
 * itself and tests/c_harness.c): the byte offsets the Julia mirrors imply [REF src/types.jl:11-19, 74-77,
 * 81-99, 101-109, 111-134, 173-217] and that interface.jl reads through `unsafe_load`
 * [REF src/interface.jl:176-205, 744-746]. */
#if defined(__cplusplus)
#define OSQP_AMD_STATIC_ASSERT(c, msg) static_assert(c, msg)
#else
#define OSQP_AMD_STATIC_ASSERT(c, msg) _Static_assert(c, msg)
#endif
#define OSQP_AMD_OFFSET(T, f, o) OSQP_AMD_STATIC_ASSERT(offsetof(T, f) == (o), #T "." #f " must sit at byte " #o)
OSQP_AMD_STATIC_ASSERT(sizeof(c_int) == 8 && sizeof(c_float) == 8 && sizeof(void *) == 8, "64-bit c_int / c_float / pointers");
OSQP_AMD_STATIC_ASSERT(sizeof(csc) == 56, "csc is 56 bytes");
OSQP_AMD_OFFSET(csc, nzmax, 0); OSQP_AMD_OFFSET(csc, m, 8); OSQP_AMD_OFFSET(csc, n, 16); OSQP_AMD_OFFSET(csc, p, 24);
OSQP_AMD_OFFSET(csc, i, 32); OSQP_AMD_OFFSET(csc, x, 40); OSQP_AMD_OFFSET(csc, nz, 48);
OSQP_AMD_STATIC_ASSERT(sizeof(OSQPData) == 56, "OSQPData is 56 bytes");
OSQP_AMD_OFFSET(OSQPData, n, 0); OSQP_AMD_OFFSET(OSQPData, m, 8); OSQP_AMD_OFFSET(OSQPData, P, 16); OSQP_AMD_OFFSET(OSQPData, A, 24);
OSQP_AMD_OFFSET(OSQPData, q, 32); OSQP_AMD_OFFSET(OSQPData, l, 40); OSQP_AMD_OFFSET(OSQPData, u, 48);
OSQP_AMD_STATIC_ASSERT(sizeof(OSQPSettings) == 176, "OSQPSettings is 176 bytes");
OSQP_AMD_OFFSET(OSQPSettings, rho, 0); OSQP_AMD_OFFSET(OSQPSettings, sigma, 8); OSQP_AMD_OFFSET(OSQPSettings, scaling, 16);
OSQP_AMD_OFFSET(OSQPSettings, adaptive_rho, 24); OSQP_AMD_OFFSET(OSQPSettings, adaptive_rho_interval, 32);
OSQP_AMD_OFFSET(OSQPSettings, adaptive_rho_tolerance, 40); OSQP_AMD_OFFSET(OSQPSettings, adaptive_rho_fraction, 48);
OSQP_AMD_OFFSET(OSQPSettings, max_iter, 56); OSQP_AMD_OFFSET(OSQPSettings, eps_abs, 64); OSQP_AMD_OFFSET(OSQPSettings, eps_rel, 72);
OSQP_AMD_OFFSET(OSQPSettings, eps_prim_inf, 80); OSQP_AMD_OFFSET(OSQPSettings, eps_dual_inf, 88); OSQP_AMD_OFFSET(OSQPSettings, alpha, 96);
OSQP_AMD_OFFSET(OSQPSettings, linsys_solver, 104); OSQP_AMD_OFFSET(OSQPSettings, delta, 112); OSQP_AMD_OFFSET(OSQPSettings, polish, 120);
OSQP_AMD_OFFSET(OSQPSettings, polish_refine_iter, 128); OSQP_AMD_OFFSET(OSQPSettings, verbose, 136);
OSQP_AMD_OFFSET(OSQPSettings, scaled_termination, 144); OSQP_AMD_OFFSET(OSQPSettings, check_termination, 152);
OSQP_AMD_OFFSET(OSQPSettings, warm_start, 160); OSQP_AMD_OFFSET(OSQPSettings, time_limit, 168);
OSQP_AMD_STATIC_ASSERT(sizeof(OSQPInfo) == 136, "OSQPInfo is 136 bytes");
OSQP_AMD_OFFSET(OSQPInfo, iter, 0); OSQP_AMD_OFFSET(OSQPInfo, status, 8); OSQP_AMD_OFFSET(OSQPInfo, status_val, 40);
OSQP_AMD_OFFSET(OSQPInfo, status_polish, 48); OSQP_AMD_OFFSET(OSQPInfo, obj_val, 56); OSQP_AMD_OFFSET(OSQPInfo, pri_res, 64);
OSQP_AMD_OFFSET(OSQPInfo, dua_res, 72); OSQP_AMD_OFFSET(OSQPInfo, setup_time, 80); OSQP_AMD_OFFSET(OSQPInfo, solve_time, 88);
OSQP_AMD_OFFSET(OSQPInfo, update_time, 96); OSQP_AMD_OFFSET(OSQPInfo, polish_time, 104); OSQP_AMD_OFFSET(OSQPInfo, run_time, 112);
OSQP_AMD_OFFSET(OSQPInfo, rho_updates, 120); OSQP_AMD_OFFSET(OSQPInfo, rho_estimate, 128);
OSQP_AMD_STATIC_ASSERT(sizeof(OSQPSolution) == 16, "OSQPSolution is 16 bytes");
OSQP_AMD_OFFSET(OSQPSolution, x, 0); OSQP_AMD_OFFSET(OSQPSolution, y, 8);
OSQP_AMD_OFFSET(OSQPWorkspace, data, 0); OSQP_AMD_OFFSET(OSQPWorkspace, linsys_solver, 8); OSQP_AMD_OFFSET(OSQPWorkspace, pol, 16);
OSQP_AMD_OFFSET(OSQPWorkspace, rho_vec, 24); OSQP_AMD_OFFSET(OSQPWorkspace, rho_inv_vec, 32); OSQP_AMD_OFFSET(OSQPWorkspace, constr_type, 40);
OSQP_AMD_OFFSET(OSQPWorkspace, x, 48); OSQP_AMD_OFFSET(OSQPWorkspace, y, 56); OSQP_AMD_OFFSET(OSQPWorkspace, z, 64);
OSQP_AMD_OFFSET(OSQPWorkspace, xz_tilde, 72); OSQP_AMD_OFFSET(OSQPWorkspace, x_prev, 80); OSQP_AMD_OFFSET(OSQPWorkspace, z_prev, 88);
OSQP_AMD_OFFSET(OSQPWorkspace, Ax, 96); OSQP_AMD_OFFSET(OSQPWorkspace, Px, 104); OSQP_AMD_OFFSET(OSQPWorkspace, Aty, 112);
OSQP_AMD_OFFSET(OSQPWorkspace, delta_y, 120); OSQP_AMD_OFFSET(OSQPWorkspace, Atdelta_y, 128); OSQP_AMD_OFFSET(OSQPWorkspace, delta_x, 136);
OSQP_AMD_OFFSET(OSQPWorkspace, Pdelta_x, 144); OSQP_AMD_OFFSET(OSQPWorkspace, Adelta_x, 152); OSQP_AMD_OFFSET(OSQPWorkspace, D_temp, 160);
OSQP_AMD_OFFSET(OSQPWorkspace, D_temp_A, 168); OSQP_AMD_OFFSET(OSQPWorkspace, E_temp, 176); OSQP_AMD_OFFSET(OSQPWorkspace, settings, 184);
OSQP_AMD_OFFSET(OSQPWorkspace, scaling, 192); OSQP_AMD_OFFSET(OSQPWorkspace, solution, 200); OSQP_AMD_OFFSET(OSQPWorkspace, info, 208);
OSQP_AMD_OFFSET(OSQPWorkspace, timer, 216); OSQP_AMD_OFFSET(OSQPWorkspace, first_run, 224); OSQP_AMD_OFFSET(OSQPWorkspace, summary_printed, 232);
OSQP_AMD_OFFSET(OSQPWorkspace, impl, 240); /* past everything the reference mirror declares (240 bytes) */

/* ------------------------------------------------------------------------- */
/* Part 1c: the 30 symbols OSQP.jl binds                                      */
/* ------------------------------------------------------------------------- */

/* [REF src/types.jl:139] */
void osqp_set_default_settings(OSQPSettings *settings);

/* [REF src/interface.jl:147]  0 on success; non-zero (1 data, 2 settings,
 * 4 linsys init, 5 non-convex, 6 alloc) makes setup! throw.
 * data->A: the row indices inside every column must ascend without repeats -- what Julia's SparseMatrixCSC guarantees
 * and ManagedCcsc copies verbatim [REF src/types.jl:21-47]; the arrays serve as the CSR arrays of A' as they are.  A
 * caller that hands over unsorted columns gets exit flag 1, not a wrong answer.  data->P (upper triangle) may come in
 * any order inside its columns. */
c_int osqp_setup(OSQPWorkspace **workp, const OSQPData *data, const OSQPSettings *settings);

/* [REF src/interface.jl:171]  return value ignored by the caller; the outcome
 * is info->status_val. */
c_int osqp_solve(OSQPWorkspace *work);

/* [REF src/interface.jl:220] */
const char *osqp_version(void);

/* [REF src/interface.jl:225]  must accept NULL (finalizer of a never-set-up
 * Model [REF src/interface.jl:24-25]). */
c_int osqp_cleanup(OSQPWorkspace *work);

/* [REF src/interface.jl:241, 259, 277, 303] */
c_int osqp_update_lin_cost(OSQPWorkspace *work, const c_float *q_new);
c_int osqp_update_lower_bound(OSQPWorkspace *work, const c_float *l_new);
c_int osqp_update_upper_bound(OSQPWorkspace *work, const c_float *u_new);
c_int osqp_update_bounds(OSQPWorkspace *work, const c_float *l_new, const c_float *u_new);

/* [REF src/interface.jl:337, 358, 382]  idx are 0-based positions into the
 * setup-time nnz order (P: upper-triangular nnz order); NULL = all nnz. */
c_int osqp_update_P(OSQPWorkspace *work, const c_float *Px_new, const c_int *Px_new_idx, c_int P_new_n);
c_int osqp_update_A(OSQPWorkspace *work, const c_float *Ax_new, const c_int *Ax_new_idx, c_int A_new_n);
c_int osqp_update_P_A(OSQPWorkspace *work, const c_float *Px_new, const c_int *Px_new_idx, c_int P_new_n,
                      const c_float *Ax_new, const c_int *Ax_new_idx, c_int A_new_n);

/* [REF src/interface.jl:476, 580, 593, 606, 619, 632, 645] */
c_int osqp_update_max_iter(OSQPWorkspace *work, c_int max_iter_new);
c_int osqp_update_polish(OSQPWorkspace *work, c_int polish_new);
c_int osqp_update_polish_refine_iter(OSQPWorkspace *work, c_int polish_refine_iter_new);
c_int osqp_update_verbose(OSQPWorkspace *work, c_int verbose_new);
c_int osqp_update_scaled_termination(OSQPWorkspace *work, c_int scaled_termination_new);
c_int osqp_update_check_termination(OSQPWorkspace *work, c_int check_termination_new);
c_int osqp_update_warm_start(OSQPWorkspace *work, c_int warm_start_new);

/* [REF src/interface.jl:489, 502, 515, 528, 541, 554, 567, 658] */
c_int osqp_update_eps_abs(OSQPWorkspace *work, c_float eps_abs_new);
c_int osqp_update_eps_rel(OSQPWorkspace *work, c_float eps_rel_new);
c_int osqp_update_eps_prim_inf(OSQPWorkspace *work, c_float eps_prim_inf_new);
c_int osqp_update_eps_dual_inf(OSQPWorkspace *work, c_float eps_dual_inf_new);
c_int osqp_update_rho(OSQPWorkspace *work, c_float rho_new);
c_int osqp_update_alpha(OSQPWorkspace *work, c_float alpha_new);
c_int osqp_update_delta(OSQPWorkspace *work, c_float delta_new);
c_int osqp_update_time_limit(OSQPWorkspace *work, c_float time_limit_new);

/* [REF src/interface.jl:676, 690, 709] */
c_int osqp_warm_start_x(OSQPWorkspace *work, const c_float *x);
c_int osqp_warm_start_y(OSQPWorkspace *work, const c_float *y);
c_int osqp_warm_start(OSQPWorkspace *work, const c_float *x, const c_float *y);

/* ------------------------------------------------------------------------- */
/* Part 2: extensions (no counterpart in the reference)                       */
/* ------------------------------------------------------------------------- */

/* Synthetic problem families of SURVEY.md section 8d; generated by a
 * counter-based SplitMix64 stream keyed by (seed, stream, index), so the host
 * generator (oracle/gen.c) and the device generator produce identical bits. */
enum osqp_amd_problem_kind {
  OSQP_AMD_GEN_RANDOM_QP = 0, /* n=m, `per_row` nnz per row of A, P = M + M' + diag */
  OSQP_AMD_GEN_LASSO     = 1, /* P diagonal, A = [I; -I], m = 2n */
  OSQP_AMD_GEN_MPC       = 2  /* nx=6, nu=4, T=10: n=100, m=200 */
};

/* Build the problem directly in HBM (no host copy, no PCIe) and run setup on
 * it.  `n` is the number of variables, `per_row` the nnz per row of A (random
 * QP only), `seed` the generator key.  Same return codes as osqp_setup. */
c_int osqp_amd_setup_generated(OSQPWorkspace **workp, c_int kind, c_int n, c_int per_row,
                               unsigned long long seed, const OSQPSettings *settings);

/* Introspection for measurement (bench.py, tests).  Fills `out[0..count)`:
 *  0 back-end in use (0 direct, 2 pcg)      1 nnz(A)          2 nnz(P full symmetric)
 *  3 nnz(triu P)                             4 nnz(L) (direct) 5 levels of the trisolve schedule
 *  6 total CG iterations so far              7 total ADMM iterations so far
 *  8 numeric factorisations so far           9 device bytes allocated
 * 10 algorithmic bytes of one SpMV with A   11 algorithmic bytes of one forward+backward trisolve
 * 12 SpMV kernel used for A: 0 CSR (k_spmv), 2 LDS-staged column panels with sliced-ELL tiles (k_spmv_sell),
 *    3 the same tiles over wide panels gathered through L2 (n >> 1e6 at fixed nnz)
 * 13 ranks of the row partition (1: not sharded)   14 all-gathers issued so far   15 bytes received by them
 * 16, 17 rows of the local blocks (n, m)          18 compact mode (CSR column / value arrays released) 0 / 1
 * 19 levels of the supernode graph when the triangular solves run by supernodes (0: by the level schedule)
 * 20 high-water mark of the device bytes allocated by this process (a sharded setup stays near 1/ranks of the whole)
 * 21 solves that were run again from a cold start because a wait inside the one-launch supernodal solve timed out
 * 22 the numeric factorisation runs by supernodes (multifrontal, one launch per supernode level and size class) 0 / 1
 * 23 the factor's index arrays (CSC pattern of L, scatter maps, supernode lists) were built on the device from a lean host analysis 0 / 1
 * 24 bytes of device address space this process has reserved for mapped blocks and never handed back (ranges are not reused
 *    -- a ROCm 7 re-map defect, csrc/devmem.hip -- so a long-lived process grows this until reservations fail and blocks fall
 *    back to hipMalloc; 128 TiB per process)
 * 25 pivots of the dense top block of the direct back-end (its Schur complement is inverted explicitly: block sweeps on the
 *    fp64 matrix cores from 512 pivots on); 0: none
 * Returns the number of entries written (at most OSQP_AMD_STATS_COUNT). */
#define OSQP_AMD_STATS_COUNT 26
c_int osqp_amd_get_stats(const OSQPWorkspace *work, c_float *out, c_int count);

/* Time `reps` launches of one hot-path kernel with HIP events on the engine's
 * own stream; returns the mean milliseconds per launch, <0 on error.
 * which: 0 SpMV A*x, 1 SpMV A'*y, 2 SpMV P*x, 3 forward+backward trisolve,
 *        4 fused ADMM vector update, 5 one whole ADMM iteration of the back-end in use without the residual
 *        evaluation (advances the iterate), 7 one all-gather of an n-vector (sharded workspaces). */
c_float osqp_amd_time_kernel(OSQPWorkspace *work, c_int which, c_int reps);

/* ---- Row-sharded workspaces: ONE large QP over several GPUs, indirect back-end ----
 *
 * Rank r of R keeps rows [r*ceil(n/R), ...) of A', of the full symmetric P and of every n-vector, and rows
 * [r*ceil(m/R), ...) of A and of every m-vector.  Per product with A, P or A' the input vector is all-gathered
 * (n or m doubles); norms and dot products are all-gathered as scalars and combined in rank order, so all
 * ranks take identical decisions.  Every rank calls the same entry points in the same order with the same
 * arguments (full-length vectors; each rank reads its slice) and receives the full solution.  Setup walks the
 * problem column range by column range and keeps only the rank's row blocks, so the peak device memory of a rank is
 * about 1/R of the single-device workspace (stats[20]; the device generator never materialises the rest at all).  Not available on a
 * sharded workspace: the direct back-end, osqp_amd_apply.  (Round 4: polish runs in its iterative form; osqp_update_P / _A / _P_A
 * take the same full-length value arrays on every rank.)
 *
 * The communicator is one in-place all-gather of doubles; it must outlive the workspaces that use it.
 *   host:  `fn(ctx, host_buf, count)` is called with world*count doubles, chunk `rank` filled in, and fills in the
 *          other chunks (MPI_Allgather, gloo, ...); returns 0.
 *   rccl:  ncclAllGather on the engine's stream.  `unique_id` = the 128 bytes osqp_amd_comm_unique_id wrote on
 *          one rank, distributed by the caller; `librccl_path` (may be NULL) names the librccl to use when the
 *          process does not already hold one. */
typedef struct osqp_amd_comm osqp_amd_comm;
typedef int (*osqp_amd_allgather_fn)(void *ctx, double *host_buf, long long count);
c_int osqp_amd_comm_create_host(osqp_amd_comm **out, c_int rank, c_int world, osqp_amd_allgather_fn fn, void *ctx);
c_int osqp_amd_comm_unique_id(void *out128, const char *librccl_path);
c_int osqp_amd_comm_create_rccl(osqp_amd_comm **out, c_int rank, c_int world, const void *unique_id,
                                const char *librccl_path);
c_int osqp_amd_comm_destroy(osqp_amd_comm *comm);
/* The communicator's one primitive, exposed: in-place all-gather of `count` doubles per rank on a DEVICE buffer of
 * world*count doubles whose chunk `rank` is filled in; blocks until done. */
c_int osqp_amd_comm_all_gather(osqp_amd_comm *comm, c_float *dev_buf, c_int count);
/* rank / size as the communicator was created, and the size the transport itself reports (RCCL: ncclCommCount; the
 * host transport: the size it was created with; -1: the transport cannot say).  Any pointer may be NULL. */
c_int osqp_amd_comm_info(const osqp_amd_comm *comm, c_int *rank, c_int *world, c_int *transport_ranks);
/* as osqp_setup [REF src/interface.jl:147-162] / osqp_amd_setup_generated, keeping this rank's row block */
c_int osqp_amd_setup_sharded(OSQPWorkspace **workp, const OSQPData *data, const OSQPSettings *settings,
                             osqp_amd_comm *comm);
c_int osqp_amd_setup_generated_sharded(OSQPWorkspace **workp, c_int kind, c_int n, c_int per_row,
                                       unsigned long long seed, const OSQPSettings *settings, osqp_amd_comm *comm);

/* Run exactly `iters` ADMM iterations from the current iterate (no
 * termination test inside, residuals refreshed at the end); used by bench.py
 * to time K steps.  Returns 0 on success. */
c_int osqp_amd_iterate(OSQPWorkspace *work, c_int iters);

/* The current iterate in the caller's units (x = D x_scaled, y = E y_scaled / c -- what osqp_solve would store
 * [REF src/interface.jl:176-186]) copied to host buffers of n and m doubles, without touching the iterate: what the
 * headline-size parity record compares after osqp_amd_iterate on the engine and on the oracle.  Either pointer
 * may be NULL. */
c_int osqp_amd_get_iterate(OSQPWorkspace *work, c_float *x_out, c_float *y_out);

/* Element-wise kernel parity hooks (tests only): run one device kernel on
 * host-provided vectors and copy the result back.
 *  op 0: y = A*x (len n -> m)   op 1: y = A'*x (m -> n)   op 2: y = P*x (n -> n)
 *  op 3: y = K^{-1} x through the linear-system back-end (n+m -> n+m)        */
c_int osqp_amd_apply(OSQPWorkspace *work, c_int op, const c_float *in, c_float *out);

/* Which kernel the last batched solve of this process ran (tests, benchmarks): -1 the 512-thread kernel (one QP per eight
 * wavefronts, the factorisation through an n x n scratch in global memory), k >= 0 entry k of the table of instantiations of
 * the four-wavefront kernel (csrc/batch.hip DevicePattern::kQuadCfg; 0 = the MPC family with its shape compiled in), -2 none yet. */
c_int osqp_amd_batch_last_kernel(void);

/* Batched path: `count` independent QPs that
 * share one sparsity pattern.  P (upper triangle) and A are given once as
 * patterns; values are [count x nnz] row-major; q,l,u are [count x n|m].
 * Outputs: x [count x n], y [count x m], info [count] (host pointers).
 * The instances are solved one per workgroup with the reduced KKT system
 * factorised in LDS.  `device` selects the HIP device (one process per GPU). */
c_int osqp_amd_batch_solve(c_int count, c_int n, c_int m,
                           const c_int *Pp, const c_int *Pi, const c_float *Px_all,
                           const c_int *Ap, const c_int *Ai, const c_float *Ax_all,
                           const c_float *q_all, const c_float *l_all, const c_float *u_all,
                           const OSQPSettings *settings,
                           c_float *x_out, c_float *y_out, OSQPInfo *info_out, c_int device);

/* Generate `count` MPC instances [first, first+count) of the mpc-batch family
 * on the device and solve them there; outputs as above but DEVICE pointers
 * (so that the caller can hand them to an RCCL gather without a host hop).
 * info_out is [count x 4] doubles: iter, status_val, pri_res, dua_res. */
c_int osqp_amd_batch_solve_generated(c_int first, c_int count, unsigned long long seed,
                                     const OSQPSettings *settings,
                                     c_float *x_dev, c_float *y_dev, c_float *info_dev, c_int device);

/* The batched path over several GPUs (SURVEY.md 8e, rows K11 + K12): `total` MPC instances of the mpc-batch family
 * (seed `seed`) cut into contiguous equal blocks, instance i -> rank floor(i / (total / world)); one process per GPU.
 * create(): this rank's block is generated in HBM (`comm` NULL = a single rank owning everything; `total` must be
 * divisible by the communicator's size).  solve(): one workgroup per instance writes its row
 * [x (100) | y (200) | iter, status_val, pri_res, dua_res] straight into the packed DEVICE array
 * `packed_dev` [total x 304], then ONE in-place all-gather of the rank blocks over the communicator (RCCL over xGMI,
 * or the host callback) fills in the other ranks' rows; returns when the whole array is valid on this rank.  No other
 * communication. */
typedef struct osqp_amd_batch osqp_amd_batch;
c_int osqp_amd_batch_mpc_create(osqp_amd_batch **out, c_int total, unsigned long long seed, const OSQPSettings *settings,
                                osqp_amd_comm *comm, c_int device);
c_int osqp_amd_batch_mpc_solve(osqp_amd_batch *batch, c_float *packed_dev);
c_int osqp_amd_batch_destroy(osqp_amd_batch *batch);

/* Device memory for callers without an allocator of their own (the packed result array above): plain hipMalloc / hipFree /
 * hipMemcpy on `device`.  copy kind: 0 device -> host, 1 host -> device, 2 device -> device; blocking. */
void *osqp_amd_device_alloc(c_int bytes, c_int device);
c_int osqp_amd_device_free(void *ptr, c_int device);
c_int osqp_amd_device_copy(void *dst, const void *src, c_int bytes, c_int kind, c_int device);

/* Differences between the batched path and osqp_setup / osqp_solve: instances share one sparsity pattern, n <= 128,
 * fewer than 65536 rows and non-zeros, everything must fit 160 KB of LDS; `adaptive_rho_interval` = 0 (automatic) means
 * every 100 iterations (there is no per-instance clock); `polish`, `time_limit`, `warm_start`, `verbose` and
 * `linsys_solver` are ignored (always a cold start, the reduced KKT system factorised in LDS); data and settings are
 * validated as by osqp_setup (1 = data, 2 = settings). */

/* Select the HIP device for workspaces created afterwards by this process
 * (one process per GPU: pass LOCAL_RANK). */
c_int osqp_amd_set_device(c_int device);

/* Host-only: the symbolic analysis of the direct back-end on the pattern of a QP (triu(P) and A in CSC form), no
 * device needed -- what the CPU tests of the ordering / level schedule / supernode partition call.
 * ordering: 0 minimum degree, 1 nested dissection, 2 minimum degree with queued ties; smax: largest supernode.
 * out[0..12]: N, nnz(L), pivot levels, supernodes, supernode levels, entries outside the diagonal blocks,
 *            doubles in the inverted blocks, largest supernode, 1 if every structural invariant holds, nnz in blocks,
 *            modelled microseconds of a solve by levels / by supernodes, 1 if the engine would take supernodes;
 * out[13] (when count >= 14): depth of a breadth-first level structure of the KKT graph (>= 400 on a problem of >= 2e5
 *            pivots sends nested dissection first) */
c_int osqp_amd_symbolic_probe(c_int n, c_int m, const c_int *Pp, const c_int *Pi, const c_int *Ap, const c_int *Ai,
                              c_int ordering, c_int smax, c_float *out, c_int count);

/* Last error message of the calling thread ("" if none). */
const char *osqp_amd_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* OSQP_AMD_H */
