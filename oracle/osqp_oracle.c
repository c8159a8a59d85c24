/*
 * oracle/osqp_oracle.c -- TEST INFRASTRUCTURE (see oracle.h for status and
 * provenance).  CPU restatement of libosqp v0.6.2's ADMM engine behind the 30
 * C symbols that osqp/OSQP.jl binds.  Every exported function names the
 * reference call site it serves as [REF file:line]; the arithmetic follows the
 * published algorithm as laid out in SURVEY.md Appendix A.1-A.7.
 *
 * Exports the same symbol names and struct layouts as the product library so
 * that one host-side mirror (osqp.jl_amd/interface.py) drives both; the two are
 * only ever loaded RTLD_LOCAL, side by side, by tests/ and bench.py.
 */
#define _POSIX_C_SOURCE 200809L
#include "oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>
#include <signal.h>

/* ---------------------------------------------------------------- private */
typedef struct {
  c_float c, cinv;
  c_float *D, *Dinv, *E, *Einv;
} scaling_t;

typedef struct {
  csc *Ared;
  c_int n_low, n_upp;
  c_int *ind_low, *ind_upp, *A_to_Alow, *A_to_Aupp;
  c_float *x, *z, *y;
  c_float obj_val, pri_res, dua_res;
} polish_t;

typedef struct {
  int kind;              /* 0 direct, 2 pcg */
  direct_solver *direct;
  pcg_solver *pcg;
} linsys_t;

typedef struct {
  struct timespec tic;
  int clear_update_time;
  int rho_update_from_solve;
  /* scaled residuals of the last update_info (feed the PCG tolerance rule) */
  c_float sc_pri_res, sc_dua_res;
  int have_res;
  c_float g_seed; int have_seed; /* max(pri, dua) at the start point of the solve */
  c_float pcg_lambda0;  /* initial lambda of the PCG tolerance rule */
  c_float pcg_lambda;   /* current lambda (quartered whenever 25+ iterations gained < 2x) */
  c_float g_ref; c_int it_ref; int have_ref;
  c_int admm_iters_total;
} priv_t;

#define PRIV(w) ((priv_t *)(w)->impl)
#define SCAL(w) ((scaling_t *)(w)->scaling)
#define POL(w) ((polish_t *)(w)->pol)
#define LIN(w) ((linsys_t *)(w)->linsys_solver)

static void tic(OSQPWorkspace *w) { clock_gettime(CLOCK_MONOTONIC, &PRIV(w)->tic); }
static c_float toc(OSQPWorkspace *w) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (c_float)(t.tv_sec - PRIV(w)->tic.tv_sec) + 1e-9 * (c_float)(t.tv_nsec - PRIV(w)->tic.tv_nsec);
}

static c_float *vec_alloc(c_int n) { return (c_float *)calloc((size_t)(n > 0 ? n : 1), sizeof(c_float)); }
static c_float c_maxf(c_float a, c_float b) { return a > b ? a : b; }
static c_float c_minf(c_float a, c_float b) { return a < b ? a : b; }

static void update_status(OSQPInfo *info, c_int status_val) {
  const char *s = "unsolved";
  info->status_val = status_val;
  switch (status_val) {
  case OSQP_SOLVED: s = "solved"; break;
  case OSQP_SOLVED_INACCURATE: s = "solved inaccurate"; break;
  case OSQP_PRIMAL_INFEASIBLE: s = "primal infeasible"; break;
  case OSQP_PRIMAL_INFEASIBLE_INACCURATE: s = "primal infeasible inaccurate"; break;
  case OSQP_DUAL_INFEASIBLE: s = "dual infeasible"; break;
  case OSQP_DUAL_INFEASIBLE_INACCURATE: s = "dual infeasible inaccurate"; break;
  case OSQP_MAX_ITER_REACHED: s = "maximum iterations reached"; break;
  case OSQP_TIME_LIMIT_REACHED: s = "run time limit reached"; break;
  case OSQP_SIGINT: s = "interrupted"; break;
  case OSQP_NON_CVX: s = "problem non convex"; break;
  default: break;
  }
  memset(info->status, 0, sizeof(info->status));
  strncpy(info->status, s, sizeof(info->status) - 1);
}

static void reset_info(OSQPInfo *info) {
  info->solve_time = 0.0;
  info->polish_time = 0.0;
  update_status(info, OSQP_UNSOLVED);
  info->rho_updates = 0;
}

/* ---------------------------------------------------------------- defaults
 * [REF src/types.jl:136-145] fetches these through osqp_set_default_settings. */
void osqp_set_default_settings(OSQPSettings *s) {
  s->rho = 0.1; s->sigma = 1e-6; s->scaling = 10;
  s->adaptive_rho = 1; s->adaptive_rho_interval = 0;
  s->adaptive_rho_tolerance = 5.0; s->adaptive_rho_fraction = 0.4;
  s->max_iter = 4000; s->eps_abs = 1e-3; s->eps_rel = 1e-3;
  s->eps_prim_inf = 1e-4; s->eps_dual_inf = 1e-4; s->alpha = 1.6;
  s->linsys_solver = QDLDL_SOLVER; s->delta = 1e-6; s->polish = 0;
  s->polish_refine_iter = 3; s->verbose = 1; s->scaled_termination = 0;
  s->check_termination = 25; s->warm_start = 1; s->time_limit = 0.0;
}

const char *osqp_version(void) { return "0.6.2"; } /* [REF src/interface.jl:220] */

/* ---------------------------------------------------------------- validation (A.1.1) */
static int validate_data(const OSQPData *d) {
  c_int j, k;
  if (!d || !d->P || !d->A || !d->q) return 1;
  if (d->n <= 0 || d->m < 0) return 1;
  if (d->P->m != d->n || d->P->n != d->n) return 1;
  for (j = 0; j < d->n; j++)
    for (k = d->P->p[j]; k < d->P->p[j + 1]; k++)
      if (d->P->i[k] > j) return 1; /* P must be upper triangular */
  if (d->A->m != d->m || d->A->n != d->n) return 1;
  for (j = 0; j < d->m; j++)
    if (d->l[j] > d->u[j]) return 1;
  return 0;
}

static int validate_settings(const OSQPSettings *s) {
  if (!s) return 1;
  if (s->scaling < 0) return 1;
  if (s->adaptive_rho != 0 && s->adaptive_rho != 1) return 1;
  if (s->adaptive_rho_interval < 0) return 1;
  if (s->adaptive_rho_fraction <= 0) return 1;
  if (s->adaptive_rho_tolerance < 1.0) return 1;
  if (s->polish_refine_iter < 0) return 1;
  if (s->rho <= 0.0 || s->sigma <= 0.0 || s->delta <= 0.0) return 1;
  if (s->max_iter <= 0) return 1;
  if (s->eps_abs < 0.0 || s->eps_rel < 0.0) return 1;
  if (s->eps_abs == 0.0 && s->eps_rel == 0.0) return 1;
  if (s->eps_prim_inf <= 0.0 || s->eps_dual_inf <= 0.0) return 1;
  if (s->alpha <= 0.0 || s->alpha >= 2.0) return 1;
  if (s->linsys_solver < 0 || s->linsys_solver > 3) return 1;
  if (s->verbose != 0 && s->verbose != 1) return 1;
  if (s->scaled_termination != 0 && s->scaled_termination != 1) return 1;
  if (s->check_termination < 0) return 1;
  if (s->warm_start != 0 && s->warm_start != 1) return 1;
  if (s->time_limit < 0.0) return 1;
  return 0;
}

/* ---------------------------------------------------------------- scaling (K0, A.1.3) */
static void limit_scaling(c_float *D, c_int n) {
  c_int i;
  for (i = 0; i < n; i++) {
    D[i] = D[i] < MIN_SCALING ? 1.0 : D[i];
    D[i] = D[i] > MAX_SCALING ? MAX_SCALING : D[i];
  }
}

static void scale_data(OSQPWorkspace *w) {
  c_int n = w->data->n, m = w->data->m, i, it;
  scaling_t *sc = SCAL(w);
  csc *P = w->data->P, *A = w->data->A;
  sc->c = 1.0;
  for (i = 0; i < n; i++) { sc->D[i] = 1.0; sc->Dinv[i] = 1.0; }
  for (i = 0; i < m; i++) { sc->E[i] = 1.0; sc->Einv[i] = 1.0; }
  for (it = 0; it < w->settings->scaling; it++) {
    /* inf-norms of the columns of [P A'; A 0] */
    mat_inf_norm_cols_sym_triu(P, w->D_temp);
    mat_inf_norm_cols(A, w->D_temp_A);
    for (i = 0; i < n; i++) w->D_temp[i] = c_maxf(w->D_temp[i], w->D_temp_A[i]);
    mat_inf_norm_rows(A, w->E_temp);
    limit_scaling(w->D_temp, n);
    limit_scaling(w->E_temp, m);
    for (i = 0; i < n; i++) w->D_temp[i] = 1.0 / sqrt(w->D_temp[i]);
    for (i = 0; i < m; i++) w->E_temp[i] = 1.0 / sqrt(w->E_temp[i]);
    mat_premult_diag(P, w->D_temp);
    mat_postmult_diag(P, w->D_temp);
    mat_premult_diag(A, w->E_temp);
    mat_postmult_diag(A, w->D_temp);
    for (i = 0; i < n; i++) w->data->q[i] *= w->D_temp[i];
    for (i = 0; i < n; i++) sc->D[i] *= w->D_temp[i];
    for (i = 0; i < m; i++) sc->E[i] *= w->E_temp[i];
    /* cost scaling */
    mat_inf_norm_cols_sym_triu(P, w->D_temp);
    c_float c_temp = 0.0;
    for (i = 0; i < n; i++) c_temp += w->D_temp[i];
    c_temp /= (c_float)n;
    c_float inf_norm_q = vec_norm_inf(w->data->q, n);
    limit_scaling(&inf_norm_q, 1);
    c_temp = c_maxf(c_temp, inf_norm_q);
    limit_scaling(&c_temp, 1);
    c_temp = 1.0 / c_temp;
    mat_mult_scalar(P, c_temp);
    for (i = 0; i < n; i++) w->data->q[i] *= c_temp;
    sc->c *= c_temp;
  }
  sc->cinv = 1.0 / sc->c;
  for (i = 0; i < n; i++) sc->Dinv[i] = 1.0 / sc->D[i];
  for (i = 0; i < m; i++) sc->Einv[i] = 1.0 / sc->E[i];
  for (i = 0; i < m; i++) { w->data->l[i] *= sc->E[i]; w->data->u[i] *= sc->E[i]; }
}

static void unscale_data(OSQPWorkspace *w) {
  c_int n = w->data->n, m = w->data->m, i;
  scaling_t *sc = SCAL(w);
  mat_mult_scalar(w->data->P, sc->cinv);
  mat_premult_diag(w->data->P, sc->Dinv);
  mat_postmult_diag(w->data->P, sc->Dinv);
  for (i = 0; i < n; i++) w->data->q[i] *= sc->cinv * sc->Dinv[i];
  mat_premult_diag(w->data->A, sc->Einv);
  mat_postmult_diag(w->data->A, sc->Dinv);
  for (i = 0; i < m; i++) { w->data->l[i] *= sc->Einv[i]; w->data->u[i] *= sc->Einv[i]; }
}

/* ---------------------------------------------------------------- rho vector (K1, A.1.4) */
static void set_rho_vec(OSQPWorkspace *w) {
  c_int i, m = w->data->m;
  w->settings->rho = c_minf(c_maxf(w->settings->rho, RHO_MIN), RHO_MAX);
  for (i = 0; i < m; i++) {
    if (w->data->l[i] < -OSQP_INFTY * MIN_SCALING && w->data->u[i] > OSQP_INFTY * MIN_SCALING) {
      w->constr_type[i] = -1; w->rho_vec[i] = RHO_MIN;
    } else if (w->data->u[i] - w->data->l[i] < RHO_TOL) {
      w->constr_type[i] = 1; w->rho_vec[i] = RHO_EQ_OVER_RHO_INEQ * w->settings->rho;
    } else {
      w->constr_type[i] = 0; w->rho_vec[i] = w->settings->rho;
    }
    w->rho_inv_vec[i] = 1.0 / w->rho_vec[i];
  }
}

static int linsys_update_rho(OSQPWorkspace *w) {
  if (LIN(w)->kind == 0) return direct_update_rho(LIN(w)->direct, w->rho_inv_vec);
  pcg_update_rho(LIN(w)->pcg, w->rho_vec);
  return 0;
}

static int update_rho_vec(OSQPWorkspace *w) {
  c_int i, m = w->data->m, changed = 0;
  for (i = 0; i < m; i++) {
    if (w->data->l[i] < -OSQP_INFTY * MIN_SCALING && w->data->u[i] > OSQP_INFTY * MIN_SCALING) {
      if (w->constr_type[i] != -1) { w->constr_type[i] = -1; w->rho_vec[i] = RHO_MIN; w->rho_inv_vec[i] = 1.0 / RHO_MIN; changed = 1; }
    } else if (w->data->u[i] - w->data->l[i] < RHO_TOL) {
      if (w->constr_type[i] != 1) {
        w->constr_type[i] = 1; w->rho_vec[i] = RHO_EQ_OVER_RHO_INEQ * w->settings->rho;
        w->rho_inv_vec[i] = 1.0 / w->rho_vec[i]; changed = 1;
      }
    } else {
      if (w->constr_type[i] != 0) {
        w->constr_type[i] = 0; w->rho_vec[i] = w->settings->rho;
        w->rho_inv_vec[i] = 1.0 / w->settings->rho; changed = 1;
      }
    }
  }
  if (changed) return linsys_update_rho(w);
  return 0;
}

/* ---------------------------------------------------------------- iterates */
static void cold_start(OSQPWorkspace *w) {
  memset(w->x, 0, sizeof(c_float) * (size_t)w->data->n);
  memset(w->z, 0, sizeof(c_float) * (size_t)w->data->m);
  memset(w->y, 0, sizeof(c_float) * (size_t)w->data->m);
}

/* tolerance of the inexact KKT solve (indirect back-end only; DESIGN.md "PCG
 * tolerance rule"): lambda * sqrt(scaled pri_res * scaled dua_res) of the last
 * residual evaluation, clamped relative to the right-hand side by the solver. */
static c_float pcg_tolerance(OSQPWorkspace *w, c_float rhs_norm) {
  priv_t *pv = PRIV(w);
  c_float hi = 1e-2 * rhs_norm, lo = 1e-13 * rhs_norm + 1e-300;
  c_float t = hi;
  if (pv->have_res) t = pv->pcg_lambda * sqrt(pv->sc_pri_res * pv->sc_dua_res);
  else if (pv->have_seed) t = pv->pcg_lambda * pv->g_seed;
  if (!(t < hi)) t = hi;
  if (t < lo) t = lo;
  return t;
}

/* one KKT solve (K2-K4 or K9) on xz_tilde, in place */
static int kkt_solve(OSQPWorkspace *w) {
  if (LIN(w)->kind == 0) { direct_solve(LIN(w)->direct, w->xz_tilde); return 0; }
  c_int n = w->data->n, m = w->data->m, i;
  /* inf-norm of the reduced right-hand side r_x + A'(rho r_z), as the HIP path computes it */
  c_float *t = w->Adelta_x, *b1 = w->Pdelta_x;
  for (i = 0; i < m; i++) t[i] = w->rho_vec[i] * w->xz_tilde[n + i];
  for (i = 0; i < n; i++) b1[i] = w->xz_tilde[i];
  mat_tpose_vec(w->data->A, t, b1, 1, 0);
  c_float tol = pcg_tolerance(w, vec_norm_inf(b1, n));
  c_int it = pcg_solve(LIN(w)->pcg, w->xz_tilde, tol);
  return it < 0 ? 1 : 0;
}

static int admm_step(OSQPWorkspace *w) {
  c_int n = w->data->n, m = w->data->m, i;
  c_float alpha = w->settings->alpha, sigma = w->settings->sigma;
  /* (A.2) rhs, KKT solve */
  for (i = 0; i < n; i++) w->xz_tilde[i] = sigma * w->x_prev[i] - w->data->q[i];
  for (i = 0; i < m; i++) w->xz_tilde[n + i] = w->z_prev[i] - w->rho_inv_vec[i] * w->y[i];
  int bad = kkt_solve(w);
  /* x update */
  for (i = 0; i < n; i++) {
    w->x[i] = alpha * w->xz_tilde[i] + (1.0 - alpha) * w->x_prev[i];
    w->delta_x[i] = w->x[i] - w->x_prev[i];
  }
  /* z update and projection */
  for (i = 0; i < m; i++) {
    w->z[i] = alpha * w->xz_tilde[n + i] + (1.0 - alpha) * w->z_prev[i] + w->rho_inv_vec[i] * w->y[i];
    w->z[i] = c_minf(c_maxf(w->z[i], w->data->l[i]), w->data->u[i]);
  }
  /* y update */
  for (i = 0; i < m; i++) {
    w->delta_y[i] = w->rho_vec[i] * (alpha * w->xz_tilde[n + i] + (1.0 - alpha) * w->z_prev[i] - w->z[i]);
    w->y[i] += w->delta_y[i];
  }
  PRIV(w)->admm_iters_total++;
  return bad;
}

/* ---------------------------------------------------------------- residuals (K8, A.3) */
static c_float compute_obj_val(OSQPWorkspace *w, const c_float *x) {
  c_float obj = quad_form(w->data->P, x) + vec_prod(w->data->q, x, w->data->n);
  if (w->settings->scaling) obj *= SCAL(w)->cinv;
  return obj;
}

static c_float compute_pri_res(OSQPWorkspace *w, const c_float *x, const c_float *z) {
  c_int m = w->data->m, i;
  mat_vec(w->data->A, x, w->Ax, 0);
  for (i = 0; i < m; i++) w->z_prev[i] = w->Ax[i] - z[i];
  PRIV(w)->sc_pri_res = vec_norm_inf(w->z_prev, m);
  if (w->settings->scaling && !w->settings->scaled_termination)
    return vec_scaled_norm_inf(SCAL(w)->Einv, w->z_prev, m);
  return PRIV(w)->sc_pri_res;
}

static c_float compute_dua_res(OSQPWorkspace *w, const c_float *x, const c_float *y) {
  c_int n = w->data->n, i;
  for (i = 0; i < n; i++) w->x_prev[i] = w->data->q[i];
  mat_vec(w->data->P, x, w->Px, 0);
  mat_tpose_vec(w->data->P, x, w->Px, 1, 1);
  for (i = 0; i < n; i++) w->x_prev[i] += w->Px[i];
  if (w->data->m > 0) {
    mat_tpose_vec(w->data->A, y, w->Aty, 0, 0);
    for (i = 0; i < n; i++) w->x_prev[i] += w->Aty[i];
  }
  PRIV(w)->sc_dua_res = vec_norm_inf(w->x_prev, n);
  if (w->settings->scaling && !w->settings->scaled_termination)
    return SCAL(w)->cinv * vec_scaled_norm_inf(SCAL(w)->Dinv, w->x_prev, n);
  return PRIV(w)->sc_dua_res;
}

static void update_info(OSQPWorkspace *w, c_int iter, int compute_objective, int polish) {
  const c_float *x, *z, *y;
  c_float *obj_val, *pri_res, *dua_res, *run_time;
  if (polish) {
    x = POL(w)->x; z = POL(w)->z; y = POL(w)->y;
    obj_val = &POL(w)->obj_val; pri_res = &POL(w)->pri_res; dua_res = &POL(w)->dua_res;
    run_time = &w->info->polish_time;
  } else {
    x = w->x; z = w->z; y = w->y;
    obj_val = &w->info->obj_val; pri_res = &w->info->pri_res; dua_res = &w->info->dua_res;
    w->info->iter = iter;
    run_time = &w->info->solve_time;
  }
  if (compute_objective) *obj_val = compute_obj_val(w, x);
  if (w->data->m == 0) { *pri_res = 0.0; PRIV(w)->sc_pri_res = 0.0; }
  else *pri_res = compute_pri_res(w, x, z);
  *dua_res = compute_dua_res(w, x, y);
  if (!polish) {
    /* progress monitor of the PCG tolerance rule: if sqrt(pri*dua) has not
     * halved over a window of >= 25 iterations the inexact solves are holding ADMM back */
    priv_t *pv = PRIV(w);
    c_float g = sqrt(pv->sc_pri_res * pv->sc_dua_res);
    pv->have_res = 1;
    if (!pv->have_ref) { pv->g_ref = g; pv->it_ref = iter; pv->have_ref = 1; }
    else if (iter - pv->it_ref >= 25) {
      if (g > 0.5 * pv->g_ref) pv->pcg_lambda = c_maxf(0.25 * pv->pcg_lambda, 1e-6);
      pv->g_ref = g; pv->it_ref = iter;
    }
  }
  *run_time = toc(w);
}

static c_float compute_pri_tol(OSQPWorkspace *w, c_float eps_abs, c_float eps_rel) {
  c_int m = w->data->m;
  c_float mx;
  if (w->settings->scaling && !w->settings->scaled_termination)
    mx = c_maxf(vec_scaled_norm_inf(SCAL(w)->Einv, w->z, m), vec_scaled_norm_inf(SCAL(w)->Einv, w->Ax, m));
  else
    mx = c_maxf(vec_norm_inf(w->z, m), vec_norm_inf(w->Ax, m));
  return eps_abs + eps_rel * mx;
}

static c_float compute_dua_tol(OSQPWorkspace *w, c_float eps_abs, c_float eps_rel) {
  c_int n = w->data->n;
  c_float mx;
  if (w->settings->scaling && !w->settings->scaled_termination) {
    mx = vec_scaled_norm_inf(SCAL(w)->Dinv, w->data->q, n);
    mx = c_maxf(mx, vec_scaled_norm_inf(SCAL(w)->Dinv, w->Aty, n));
    mx = c_maxf(mx, vec_scaled_norm_inf(SCAL(w)->Dinv, w->Px, n));
    mx *= SCAL(w)->cinv;
  } else {
    mx = vec_norm_inf(w->data->q, n);
    mx = c_maxf(mx, vec_norm_inf(w->Aty, n));
    mx = c_maxf(mx, vec_norm_inf(w->Px, n));
  }
  return eps_abs + eps_rel * mx;
}

static int is_primal_infeasible(OSQPWorkspace *w, c_float eps_prim_inf) {
  c_int m = w->data->m, n = w->data->n, i;
  c_float norm_delta_y, ineq_lhs = 0.0;
  int unscale = w->settings->scaling && !w->settings->scaled_termination;
  /* project delta_y onto the polar of the recession cone of [l,u] */
  for (i = 0; i < m; i++) {
    if (w->data->u[i] > OSQP_INFTY * MIN_SCALING) {
      if (w->data->l[i] < -OSQP_INFTY * MIN_SCALING) w->delta_y[i] = 0.0;
      else w->delta_y[i] = c_minf(w->delta_y[i], 0.0);
    } else if (w->data->l[i] < -OSQP_INFTY * MIN_SCALING) {
      w->delta_y[i] = c_maxf(w->delta_y[i], 0.0);
    }
  }
  if (unscale) norm_delta_y = vec_scaled_norm_inf(SCAL(w)->E, w->delta_y, m);
  else norm_delta_y = vec_norm_inf(w->delta_y, m);
  if (norm_delta_y > eps_prim_inf) {
    for (i = 0; i < m; i++)
      ineq_lhs += w->data->u[i] * c_maxf(w->delta_y[i], 0.0) + w->data->l[i] * c_minf(w->delta_y[i], 0.0);
    if (ineq_lhs < -eps_prim_inf * norm_delta_y) {
      mat_tpose_vec(w->data->A, w->delta_y, w->Atdelta_y, 0, 0);
      if (unscale) for (i = 0; i < n; i++) w->Atdelta_y[i] *= SCAL(w)->Dinv[i];
      return vec_norm_inf(w->Atdelta_y, n) < eps_prim_inf * norm_delta_y;
    }
  }
  return 0;
}

static int is_dual_infeasible(OSQPWorkspace *w, c_float eps_dual_inf) {
  c_int m = w->data->m, n = w->data->n, i;
  c_float norm_delta_x, cost_scaling;
  int unscale = w->settings->scaling && !w->settings->scaled_termination;
  if (unscale) { norm_delta_x = vec_scaled_norm_inf(SCAL(w)->D, w->delta_x, n); cost_scaling = SCAL(w)->c; }
  else { norm_delta_x = vec_norm_inf(w->delta_x, n); cost_scaling = 1.0; }
  if (norm_delta_x > eps_dual_inf) {
    if (vec_prod(w->data->q, w->delta_x, n) < -cost_scaling * eps_dual_inf * norm_delta_x) {
      mat_vec(w->data->P, w->delta_x, w->Pdelta_x, 0);
      mat_tpose_vec(w->data->P, w->delta_x, w->Pdelta_x, 1, 1);
      if (unscale) for (i = 0; i < n; i++) w->Pdelta_x[i] *= SCAL(w)->Dinv[i];
      if (vec_norm_inf(w->Pdelta_x, n) < cost_scaling * eps_dual_inf * norm_delta_x) {
        mat_vec(w->data->A, w->delta_x, w->Adelta_x, 0);
        if (unscale) for (i = 0; i < m; i++) w->Adelta_x[i] *= SCAL(w)->Einv[i];
        for (i = 0; i < m; i++) {
          if ((w->data->u[i] < OSQP_INFTY * MIN_SCALING && w->Adelta_x[i] > eps_dual_inf * norm_delta_x) ||
              (w->data->l[i] > -OSQP_INFTY * MIN_SCALING && w->Adelta_x[i] < -eps_dual_inf * norm_delta_x))
            return 0;
        }
        return 1;
      }
    }
  }
  return 0;
}

static int check_termination(OSQPWorkspace *w, int approximate) {
  c_float eps_abs = w->settings->eps_abs, eps_rel = w->settings->eps_rel;
  c_float eps_prim_inf = w->settings->eps_prim_inf, eps_dual_inf = w->settings->eps_dual_inf;
  int exitflag = 0, prim_res_check = 0, dual_res_check = 0, prim_inf_check = 0, dual_inf_check = 0;
  c_int i;
  if (w->info->pri_res > OSQP_INFTY || w->info->dua_res > OSQP_INFTY ||
      w->info->pri_res != w->info->pri_res || w->info->dua_res != w->info->dua_res) {
    update_status(w->info, OSQP_NON_CVX);
    w->info->obj_val = ORACLE_NAN;
    return 1;
  }
  if (approximate) { eps_abs *= 10; eps_rel *= 10; eps_prim_inf *= 10; eps_dual_inf *= 10; }
  if (w->data->m == 0) prim_res_check = 1;
  else {
    c_float eps_prim = compute_pri_tol(w, eps_abs, eps_rel);
    if (w->info->pri_res < eps_prim) prim_res_check = 1;
    else prim_inf_check = is_primal_infeasible(w, eps_prim_inf);
  }
  c_float eps_dual = compute_dua_tol(w, eps_abs, eps_rel);
  if (w->info->dua_res < eps_dual) dual_res_check = 1;
  else dual_inf_check = is_dual_infeasible(w, eps_dual_inf);

  if (prim_res_check && dual_res_check) {
    update_status(w->info, approximate ? OSQP_SOLVED_INACCURATE : OSQP_SOLVED);
    exitflag = 1;
  } else if (prim_inf_check) {
    update_status(w->info, approximate ? OSQP_PRIMAL_INFEASIBLE_INACCURATE : OSQP_PRIMAL_INFEASIBLE);
    if (w->settings->scaling && !w->settings->scaled_termination)
      for (i = 0; i < w->data->m; i++) w->delta_y[i] *= SCAL(w)->E[i];
    w->info->obj_val = OSQP_INFTY;
    exitflag = 1;
  } else if (dual_inf_check) {
    update_status(w->info, approximate ? OSQP_DUAL_INFEASIBLE_INACCURATE : OSQP_DUAL_INFEASIBLE);
    if (w->settings->scaling && !w->settings->scaled_termination)
      for (i = 0; i < w->data->n; i++) w->delta_x[i] *= SCAL(w)->D[i];
    w->info->obj_val = -OSQP_INFTY;
    exitflag = 1;
  }
  return exitflag;
}

/* ---------------------------------------------------------------- adaptive rho (A.4) */
static c_float compute_rho_estimate(OSQPWorkspace *w) {
  c_int n = w->data->n, m = w->data->m;
  c_float pri_res = vec_norm_inf(w->z_prev, m); /* Ax - z, left there by compute_pri_res */
  c_float dua_res = vec_norm_inf(w->x_prev, n); /* Px + q + A'y, left by compute_dua_res */
  c_float pri_norm = c_maxf(vec_norm_inf(w->z, m), vec_norm_inf(w->Ax, m));
  pri_res /= (pri_norm + 1e-10);
  c_float dua_norm = c_maxf(vec_norm_inf(w->data->q, n), vec_norm_inf(w->Aty, n));
  dua_norm = c_maxf(dua_norm, vec_norm_inf(w->Px, n));
  dua_res /= (dua_norm + 1e-10);
  c_float est = w->settings->rho * sqrt(pri_res / (dua_res + 1e-10));
  return c_minf(c_maxf(est, RHO_MIN), RHO_MAX);
}

static int adapt_rho(OSQPWorkspace *w) {
  c_float rho_new = compute_rho_estimate(w);
  w->info->rho_estimate = rho_new;
  if (rho_new > w->settings->rho * w->settings->adaptive_rho_tolerance ||
      rho_new < w->settings->rho / w->settings->adaptive_rho_tolerance) {
    int e = (int)osqp_update_rho(w, rho_new);
    w->info->rho_updates += 1;
    return e;
  }
  return 0;
}

/* ---------------------------------------------------------------- solution store (A.5) */
static int has_solution(const OSQPInfo *info) {
  return info->status_val != OSQP_PRIMAL_INFEASIBLE && info->status_val != OSQP_PRIMAL_INFEASIBLE_INACCURATE &&
         info->status_val != OSQP_DUAL_INFEASIBLE && info->status_val != OSQP_DUAL_INFEASIBLE_INACCURATE &&
         info->status_val != OSQP_NON_CVX;
}

static void store_solution(OSQPWorkspace *w) {
  c_int n = w->data->n, m = w->data->m, i;
  if (has_solution(w->info)) {
    for (i = 0; i < n; i++) w->solution->x[i] = w->x[i];
    for (i = 0; i < m; i++) w->solution->y[i] = w->y[i];
    if (w->settings->scaling) {
      for (i = 0; i < n; i++) w->solution->x[i] *= SCAL(w)->D[i];
      for (i = 0; i < m; i++) w->solution->y[i] *= SCAL(w)->E[i] * SCAL(w)->cinv;
    }
  } else {
    for (i = 0; i < n; i++) w->solution->x[i] = ORACLE_NAN;
    for (i = 0; i < m; i++) w->solution->y[i] = ORACLE_NAN;
    if (w->info->status_val == OSQP_PRIMAL_INFEASIBLE || w->info->status_val == OSQP_PRIMAL_INFEASIBLE_INACCURATE) {
      c_float nrm = vec_norm_inf(w->delta_y, m);
      for (i = 0; i < m; i++) w->delta_y[i] /= nrm;
    }
    if (w->info->status_val == OSQP_DUAL_INFEASIBLE || w->info->status_val == OSQP_DUAL_INFEASIBLE_INACCURATE) {
      c_float nrm = vec_norm_inf(w->delta_x, n);
      for (i = 0; i < n; i++) w->delta_x[i] /= nrm;
    }
    cold_start(w);
  }
}

/* ---------------------------------------------------------------- polish (A.6, row N1) */
static c_int form_Ared(OSQPWorkspace *w) {
  polish_t *pol = POL(w);
  c_int m = w->data->m, n = w->data->n, i, j, k, nnz = 0;
  const csc *A = w->data->A;
  pol->n_low = 0; pol->n_upp = 0;
  for (i = 0; i < m; i++) {
    if (w->z[i] - w->data->l[i] < -w->y[i]) { pol->ind_low[pol->n_low] = i; pol->A_to_Alow[i] = pol->n_low++; }
    else pol->A_to_Alow[i] = -1;
  }
  for (i = 0; i < m; i++) {
    if (w->data->u[i] - w->z[i] < w->y[i]) { pol->ind_upp[pol->n_upp] = i; pol->A_to_Aupp[i] = pol->n_upp++; }
    else pol->A_to_Aupp[i] = -1;
  }
  c_int mred = pol->n_low + pol->n_upp;
  if (pol->Ared) { csc_free(pol->Ared); pol->Ared = NULL; }
  if (mred == 0) { pol->Ared = csc_alloc(0, n, 0); return 0; }
  for (k = 0; k < A->p[n]; k++)
    if (pol->A_to_Alow[A->i[k]] != -1 || pol->A_to_Aupp[A->i[k]] != -1) nnz++;
  pol->Ared = csc_alloc(mred, n, nnz);
  nnz = 0;
  for (j = 0; j < n; j++) {
    pol->Ared->p[j] = nnz;
    for (k = A->p[j]; k < A->p[j + 1]; k++) {
      i = A->i[k];
      if (pol->A_to_Alow[i] != -1) { pol->Ared->i[nnz] = pol->A_to_Alow[i]; pol->Ared->x[nnz++] = A->x[k]; }
      else if (pol->A_to_Aupp[i] != -1) { pol->Ared->i[nnz] = pol->A_to_Aupp[i] + pol->n_low; pol->Ared->x[nnz++] = A->x[k]; }
    }
  }
  pol->Ared->p[n] = nnz;
  return mred;
}

static int polish(OSQPWorkspace *w) {
  polish_t *pol = POL(w);
  c_int n = w->data->n, m = w->data->m, i, j, it;
  tic(w);
  c_int mred = form_Ared(w);
  if (mred < 0) { w->info->status_polish = -1; return -1; }
  int err = 0;
  direct_solver *plsh = direct_init(w->data->P, pol->Ared, w->settings->delta, NULL, 1, &err);
  if (!plsh) { w->info->status_polish = -1; w->info->polish_time = toc(w); return 1; }
  c_int nr = n + mred;
  c_float *rhs_red = vec_alloc(nr), *pol_sol = vec_alloc(nr), *rhs = vec_alloc(nr);
  for (i = 0; i < n; i++) rhs_red[i] = -w->data->q[i];
  for (i = 0; i < pol->n_low; i++) rhs_red[n + i] = w->data->l[pol->ind_low[i]];
  for (i = 0; i < pol->n_upp; i++) rhs_red[n + pol->n_low + i] = w->data->u[pol->ind_upp[i]];
  memcpy(pol_sol, rhs_red, sizeof(c_float) * (size_t)nr);
  direct_solve(plsh, pol_sol);
  /* iterative refinement against the unregularised matrix */
  for (it = 0; it < w->settings->polish_refine_iter; it++) {
    memcpy(rhs, rhs_red, sizeof(c_float) * (size_t)nr);
    mat_vec(w->data->P, pol_sol, rhs, -1);
    mat_tpose_vec(w->data->P, pol_sol, rhs, -1, 1);
    mat_tpose_vec(pol->Ared, pol_sol + n, rhs, -1, 0);
    mat_vec(pol->Ared, pol_sol, rhs + n, -1);
    direct_solve(plsh, rhs);
    for (j = 0; j < nr; j++) pol_sol[j] += rhs[j];
  }
  for (i = 0; i < n; i++) pol->x[i] = pol_sol[i];
  mat_vec(w->data->A, pol->x, pol->z, 0);
  for (i = 0; i < m; i++) {
    if (pol->A_to_Alow[i] != -1) pol->y[i] = pol_sol[n + pol->A_to_Alow[i]];
    else if (pol->A_to_Aupp[i] != -1) pol->y[i] = pol_sol[n + pol->n_low + pol->A_to_Aupp[i]];
    else pol->y[i] = 0.0;
  }
  /* (z, y) onto the normal cone of [l, u] */
  for (i = 0; i < m; i++) {
    c_float s = pol->z[i] + pol->y[i];
    pol->z[i] = c_minf(c_maxf(s, w->data->l[i]), w->data->u[i]);
    pol->y[i] = s - pol->z[i];
  }
  update_info(w, 0, 1, 1);
  int ok = (pol->pri_res < w->info->pri_res && pol->dua_res < w->info->dua_res) ||
           (pol->pri_res < w->info->pri_res && w->info->dua_res < 1e-10) ||
           (pol->dua_res < w->info->dua_res && w->info->pri_res < 1e-10);
  if (ok) {
    w->info->obj_val = pol->obj_val; w->info->pri_res = pol->pri_res; w->info->dua_res = pol->dua_res;
    w->info->status_polish = 1;
    memcpy(w->x, pol->x, sizeof(c_float) * (size_t)n);
    memcpy(w->z, pol->z, sizeof(c_float) * (size_t)m);
    memcpy(w->y, pol->y, sizeof(c_float) * (size_t)m);
  } else {
    w->info->status_polish = -1;
  }
  direct_free(plsh);
  free(rhs_red); free(pol_sol); free(rhs);
  return 0;
}

/* ---------------------------------------------------------------- setup
 * [REF src/interface.jl:147]; steps of SURVEY.md A.1. */
c_int osqp_setup(OSQPWorkspace **workp, const OSQPData *data, const OSQPSettings *settings) {
  if (validate_data(data)) return 1;
  if (validate_settings(settings)) return 2;
  OSQPWorkspace *w = (OSQPWorkspace *)calloc(1, sizeof(OSQPWorkspace));
  *workp = w;
  w->impl = calloc(1, sizeof(priv_t));
  PRIV(w)->pcg_lambda0 = 0.015;
  if (getenv("OSQP_ORACLE_PCG_LAMBDA")) PRIV(w)->pcg_lambda0 = atof(getenv("OSQP_ORACLE_PCG_LAMBDA"));
  PRIV(w)->pcg_lambda = PRIV(w)->pcg_lambda0;
  tic(w);
  c_int n = data->n, m = data->m;
  w->data = (OSQPData *)calloc(1, sizeof(OSQPData));
  w->data->n = n; w->data->m = m;
  w->data->P = csc_copy(data->P);
  w->data->A = csc_copy(data->A);
  w->data->q = vec_alloc(n); memcpy(w->data->q, data->q, sizeof(c_float) * (size_t)n);
  w->data->l = vec_alloc(m); w->data->u = vec_alloc(m);
  if (m > 0) { memcpy(w->data->l, data->l, sizeof(c_float) * (size_t)m); memcpy(w->data->u, data->u, sizeof(c_float) * (size_t)m); }
  w->rho_vec = vec_alloc(m); w->rho_inv_vec = vec_alloc(m);
  w->constr_type = (c_int *)calloc((size_t)(m > 0 ? m : 1), sizeof(c_int));
  w->x = vec_alloc(n); w->z = vec_alloc(m); w->xz_tilde = vec_alloc(n + m);
  w->x_prev = vec_alloc(n); w->z_prev = vec_alloc(m); w->y = vec_alloc(m);
  w->Ax = vec_alloc(m); w->Px = vec_alloc(n); w->Aty = vec_alloc(n);
  w->delta_y = vec_alloc(m); w->Atdelta_y = vec_alloc(n);
  w->delta_x = vec_alloc(n); w->Pdelta_x = vec_alloc(n); w->Adelta_x = vec_alloc(m);
  w->settings = (OSQPSettings *)malloc(sizeof(OSQPSettings));
  *w->settings = *settings;
  w->D_temp = vec_alloc(n); w->D_temp_A = vec_alloc(n); w->E_temp = vec_alloc(m);
  scaling_t *sc = (scaling_t *)calloc(1, sizeof(scaling_t));
  w->scaling = sc;
  sc->D = vec_alloc(n); sc->Dinv = vec_alloc(n); sc->E = vec_alloc(m); sc->Einv = vec_alloc(m);
  sc->c = 1.0; sc->cinv = 1.0;
  c_int i;
  for (i = 0; i < n; i++) { sc->D[i] = 1.0; sc->Dinv[i] = 1.0; }
  for (i = 0; i < m; i++) { sc->E[i] = 1.0; sc->Einv[i] = 1.0; }
  if (settings->scaling) scale_data(w);
  set_rho_vec(w);
  linsys_t *ls = (linsys_t *)calloc(1, sizeof(linsys_t));
  w->linsys_solver = ls;
  if (settings->linsys_solver == AMD_PCG_SOLVER) {
    ls->kind = 2;
    ls->pcg = pcg_init(w->data->P, w->data->A, w->settings->sigma, w->rho_vec);
  } else {
    int err = 0;
    ls->kind = 0;
    ls->direct = direct_init(w->data->P, w->data->A, w->settings->sigma, w->rho_inv_vec, 0, &err);
    if (!ls->direct) { osqp_cleanup(w); *workp = NULL; return err; }
  }
  polish_t *pol = (polish_t *)calloc(1, sizeof(polish_t));
  w->pol = pol;
  pol->ind_low = (c_int *)calloc((size_t)(m > 0 ? m : 1), sizeof(c_int));
  pol->ind_upp = (c_int *)calloc((size_t)(m > 0 ? m : 1), sizeof(c_int));
  pol->A_to_Alow = (c_int *)calloc((size_t)(m > 0 ? m : 1), sizeof(c_int));
  pol->A_to_Aupp = (c_int *)calloc((size_t)(m > 0 ? m : 1), sizeof(c_int));
  pol->x = vec_alloc(n); pol->z = vec_alloc(m); pol->y = vec_alloc(m);
  w->solution = (OSQPSolution *)calloc(1, sizeof(OSQPSolution));
  w->solution->x = vec_alloc(n); w->solution->y = vec_alloc(m);
  w->info = (OSQPInfo *)calloc(1, sizeof(OSQPInfo));
  w->info->status_polish = 0;
  update_status(w->info, OSQP_UNSOLVED);
  w->info->rho_estimate = w->settings->rho;
  w->first_run = 1;
  w->summary_printed = 0;
  PRIV(w)->clear_update_time = 0;
  PRIV(w)->rho_update_from_solve = 0;
  w->info->setup_time = toc(w);
  if (w->settings->verbose)
    printf("[osqp-oracle] n = %lld, m = %lld, nnz(P)+nnz(A) = %lld, linsys = %s\n", (long long)n, (long long)m,
           (long long)(w->data->P->p[n] + w->data->A->p[n]), ls->kind == 0 ? "direct LDL'" : "PCG");
  return 0;
}

/* ---------------------------------------------------------------- solve
 * [REF src/interface.jl:171]; loop of SURVEY.md A.2-A.5. */
/* ---------------------------------------------------------------- Ctrl-C
 * The published library listens for SIGINT while osqp_solve runs (its CTRLC build, the one the reference ships): the
 * handler only sets a flag, the loop tests it at the top of every iteration, the status becomes OSQP_SIGINT (-5)
 * [REF src/constants.jl:17 :Interrupted], osqp_solve returns 1 without storing a solution, and the handler that was
 * installed before the call is put back. */
static volatile sig_atomic_t oracle_sigint = 0;
static void oracle_on_sigint(int sig) { (void)sig; oracle_sigint = 1; }

c_int osqp_solve(OSQPWorkspace *w) {
  if (!w) return 7;
  struct sigaction sa_new, sa_old;
  memset(&sa_new, 0, sizeof sa_new);
  sa_new.sa_handler = oracle_on_sigint;
  sigemptyset(&sa_new.sa_mask);
  oracle_sigint = 0;
  sigaction(SIGINT, &sa_new, &sa_old);
  c_int iter, max_iter = w->settings->max_iter;
  int can_check_termination = 0, can_print = (int)w->settings->verbose;
  int compute_cost_function = (int)w->settings->verbose;
  c_float temp_run_time;
  if (PRIV(w)->clear_update_time == 1) w->info->update_time = 0.0;
  PRIV(w)->rho_update_from_solve = 1;
  tic(w);
  if (w->settings->verbose) printf("iter   objective    pri res    dua res    rho\n");
  if (!w->settings->warm_start) cold_start(w);
  if (LIN(w)->kind == 2) pcg_set_guess(LIN(w)->pcg, w->x);
  PRIV(w)->have_res = 0; PRIV(w)->have_ref = 0; PRIV(w)->pcg_lambda = PRIV(w)->pcg_lambda0;
  PRIV(w)->have_seed = 0;
  if (LIN(w)->kind == 2) { /* seed the PCG tolerance rule with the residuals of the start point */
    c_float p0 = w->data->m ? compute_pri_res(w, w->x, w->z) : 0.0;
    c_float d0 = compute_dua_res(w, w->x, w->y);
    (void)p0; (void)d0;
    PRIV(w)->g_seed = c_maxf(w->data->m ? PRIV(w)->sc_pri_res : 0.0, PRIV(w)->sc_dua_res);
    PRIV(w)->have_seed = 1;
  }

  for (iter = 1; iter <= max_iter; iter++) {
    if (oracle_sigint) { /* no solution is stored, the iterate stays as it is (a later solve warm-starts from it) */
      update_status(w->info, OSQP_SIGINT);
      w->info->solve_time = toc(w);
      PRIV(w)->rho_update_from_solve = 0;
      sigaction(SIGINT, &sa_old, NULL);
      for (c_int k = 0; k < w->data->n; k++) w->solution->x[k] = ORACLE_NAN;
      for (c_int k = 0; k < w->data->m; k++) w->solution->y[k] = ORACLE_NAN;
      return 1;
    }
    /* time limit (A.3 last bullet) */
    if (w->first_run) temp_run_time = w->info->setup_time + toc(w);
    else temp_run_time = w->info->update_time + toc(w);
    if (w->settings->time_limit && temp_run_time >= w->settings->time_limit) {
      update_status(w->info, OSQP_TIME_LIMIT_REACHED);
      can_check_termination = 0;
      break;
    }
    { c_float *t = w->x; w->x = w->x_prev; w->x_prev = t; t = w->z; w->z = w->z_prev; w->z_prev = t; }
    if (admm_step(w)) { /* negative curvature met by the indirect solve */
      update_status(w->info, OSQP_NON_CVX); w->info->obj_val = ORACLE_NAN; w->info->iter = iter;
      break;
    }
    can_check_termination = w->settings->check_termination && (iter % w->settings->check_termination == 0);
    can_print = w->settings->verbose && ((iter % 200 == 0) || iter == 1);
    if (can_check_termination || can_print) {
      update_info(w, iter, compute_cost_function, 0);
      if (can_print) printf("%4lld  %11.4e  %9.2e  %9.2e  %9.2e\n", (long long)iter, w->info->obj_val, w->info->pri_res, w->info->dua_res, w->settings->rho);
      if (can_check_termination && check_termination(w, 0)) break;
    }
    /* adaptive rho (A.4) */
    if (w->settings->adaptive_rho && !w->settings->adaptive_rho_interval) {
      if (toc(w) > w->settings->adaptive_rho_fraction * w->info->setup_time) {
        c_int base = w->settings->check_termination ? w->settings->check_termination : 25;
        c_int rounded = (c_int)(base * (c_int)floor((double)iter / (double)base + 0.5));
        if (rounded < base) rounded = base;
        w->settings->adaptive_rho_interval = rounded;
        if (w->settings->adaptive_rho_interval < w->settings->check_termination)
          w->settings->adaptive_rho_interval = w->settings->check_termination;
      }
    }
    if (w->settings->adaptive_rho && w->settings->adaptive_rho_interval &&
        (iter % w->settings->adaptive_rho_interval == 0)) {
      if (!can_check_termination && !can_print) update_info(w, iter, compute_cost_function, 0);
      if (adapt_rho(w)) { update_status(w->info, OSQP_NON_CVX); break; }
    }
  }

  /* residuals / termination test if the last pass through the loop did not do it
   * (max_iter reached off-cadence, check_termination disabled, time limit) */
  if (!can_check_termination && w->info->status_val != OSQP_NON_CVX) {
    if (!can_print) update_info(w, iter - 1, compute_cost_function, 0);
    check_termination(w, 0);
  }
  if (!compute_cost_function && has_solution(w->info)) w->info->obj_val = compute_obj_val(w, w->x);
  if (w->info->status_val == OSQP_UNSOLVED) {
    if (!check_termination(w, 1)) update_status(w->info, OSQP_MAX_ITER_REACHED);
  }
  if (w->info->status_val == OSQP_TIME_LIMIT_REACHED) {
    if (!check_termination(w, 1)) update_status(w->info, OSQP_TIME_LIMIT_REACHED);
  }
  w->info->rho_estimate = compute_rho_estimate(w);
  w->info->solve_time = toc(w);
  if (w->settings->polish && w->info->status_val == OSQP_SOLVED) polish(w);
  if (w->first_run) w->info->run_time = w->info->setup_time + w->info->solve_time + w->info->polish_time;
  else w->info->run_time = w->info->update_time + w->info->solve_time + w->info->polish_time;
  if (w->first_run) w->first_run = 0;
  PRIV(w)->clear_update_time = 1;
  PRIV(w)->rho_update_from_solve = 0;
  if (w->settings->verbose)
    printf("status: %s, iterations: %lld, objective: %.6e, run time: %.3es\n", w->info->status,
           (long long)w->info->iter, w->info->obj_val, w->info->run_time);
  store_solution(w);
  sigaction(SIGINT, &sa_old, NULL);
  return 0;
}

/* ---------------------------------------------------------------- cleanup
 * [REF src/interface.jl:225]; NULL accepted (finalizer on an empty Model). */
c_int osqp_cleanup(OSQPWorkspace *w) {
  if (!w) return 0;
  if (w->data) {
    csc_free(w->data->P); csc_free(w->data->A);
    free(w->data->q); free(w->data->l); free(w->data->u); free(w->data);
  }
  if (w->scaling) { scaling_t *sc = SCAL(w); free(sc->D); free(sc->Dinv); free(sc->E); free(sc->Einv); free(sc); }
  if (w->linsys_solver) { direct_free(LIN(w)->direct); pcg_free(LIN(w)->pcg); free(w->linsys_solver); }
  if (w->pol) {
    polish_t *p = POL(w);
    csc_free(p->Ared); free(p->ind_low); free(p->ind_upp); free(p->A_to_Alow); free(p->A_to_Aupp);
    free(p->x); free(p->z); free(p->y); free(p);
  }
  free(w->rho_vec); free(w->rho_inv_vec); free(w->constr_type);
  free(w->x); free(w->z); free(w->xz_tilde); free(w->x_prev); free(w->z_prev); free(w->y);
  free(w->Ax); free(w->Px); free(w->Aty); free(w->delta_y); free(w->Atdelta_y);
  free(w->delta_x); free(w->Pdelta_x); free(w->Adelta_x);
  free(w->D_temp); free(w->D_temp_A); free(w->E_temp);
  free(w->settings);
  if (w->solution) { free(w->solution->x); free(w->solution->y); free(w->solution); }
  free(w->info); free(w->impl);
  free(w);
  return 0;
}

/* ---------------------------------------------------------------- data updates (A.7) */
static void begin_update(OSQPWorkspace *w) {
  if (PRIV(w)->clear_update_time == 1) { PRIV(w)->clear_update_time = 0; w->info->update_time = 0.0; }
  tic(w);
}
static void end_update(OSQPWorkspace *w) { w->info->update_time += toc(w); }

/* [REF src/interface.jl:241] */
c_int osqp_update_lin_cost(OSQPWorkspace *w, const c_float *q_new) {
  if (!w) return 7;
  begin_update(w);
  c_int n = w->data->n, i;
  memcpy(w->data->q, q_new, sizeof(c_float) * (size_t)n);
  if (w->settings->scaling) for (i = 0; i < n; i++) { w->data->q[i] *= SCAL(w)->D[i]; w->data->q[i] *= SCAL(w)->c; }
  reset_info(w->info);
  end_update(w);
  return 0;
}

/* [REF src/interface.jl:303] */
c_int osqp_update_bounds(OSQPWorkspace *w, const c_float *l_new, const c_float *u_new) {
  if (!w) return 7;
  begin_update(w);
  c_int m = w->data->m, i;
  for (i = 0; i < m; i++) if (l_new[i] > u_new[i]) return 1;
  memcpy(w->data->l, l_new, sizeof(c_float) * (size_t)m);
  memcpy(w->data->u, u_new, sizeof(c_float) * (size_t)m);
  if (w->settings->scaling) for (i = 0; i < m; i++) { w->data->l[i] *= SCAL(w)->E[i]; w->data->u[i] *= SCAL(w)->E[i]; }
  reset_info(w->info);
  int e = update_rho_vec(w);
  end_update(w);
  return e;
}

/* [REF src/interface.jl:259] */
c_int osqp_update_lower_bound(OSQPWorkspace *w, const c_float *l_new) {
  if (!w) return 7;
  begin_update(w);
  c_int m = w->data->m, i;
  memcpy(w->data->l, l_new, sizeof(c_float) * (size_t)m);
  if (w->settings->scaling) for (i = 0; i < m; i++) w->data->l[i] *= SCAL(w)->E[i];
  for (i = 0; i < m; i++) if (w->data->l[i] > w->data->u[i]) return 1;
  reset_info(w->info);
  int e = update_rho_vec(w);
  end_update(w);
  return e;
}

/* [REF src/interface.jl:277] */
c_int osqp_update_upper_bound(OSQPWorkspace *w, const c_float *u_new) {
  if (!w) return 7;
  begin_update(w);
  c_int m = w->data->m, i;
  memcpy(w->data->u, u_new, sizeof(c_float) * (size_t)m);
  if (w->settings->scaling) for (i = 0; i < m; i++) w->data->u[i] *= SCAL(w)->E[i];
  for (i = 0; i < m; i++) if (w->data->l[i] > w->data->u[i]) return 1;
  reset_info(w->info);
  int e = update_rho_vec(w);
  end_update(w);
  return e;
}

static int linsys_update_matrices(OSQPWorkspace *w) {
  if (LIN(w)->kind == 0) return direct_update_matrices(LIN(w)->direct, w->data->P, w->data->A);
  pcg_update_matrices(LIN(w)->pcg, w->data->P, w->data->A);
  return 0;
}

static c_int update_PA(OSQPWorkspace *w, const c_float *Px_new, const c_int *Px_idx, c_int P_n,
                       const c_float *Ax_new, const c_int *Ax_idx, c_int A_n, int doP, int doA) {
  if (!w) return 7;
  begin_update(w);
  c_int nnzP = w->data->P->p[w->data->P->n], nnzA = w->data->A->p[w->data->A->n], i;
  if (doP) { if (Px_idx) { if (P_n > nnzP) return 1; } else if (P_n != nnzP && P_n != 0) return 1; }
  if (doA) { if (Ax_idx) { if (A_n > nnzA) return 2; } else if (A_n != nnzA && A_n != 0) return 2; }
  if (w->settings->scaling) unscale_data(w);
  if (doP) {
    if (Px_idx) for (i = 0; i < P_n; i++) w->data->P->x[Px_idx[i]] = Px_new[i];
    else for (i = 0; i < nnzP; i++) w->data->P->x[i] = Px_new[i];
  }
  if (doA) {
    if (Ax_idx) for (i = 0; i < A_n; i++) w->data->A->x[Ax_idx[i]] = Ax_new[i];
    else for (i = 0; i < nnzA; i++) w->data->A->x[i] = Ax_new[i];
  }
  if (w->settings->scaling) scale_data(w);
  int e = linsys_update_matrices(w);
  reset_info(w->info);
  end_update(w);
  return e;
}

/* [REF src/interface.jl:337, 358, 382] */
c_int osqp_update_P(OSQPWorkspace *w, const c_float *Px_new, const c_int *Px_new_idx, c_int P_new_n) {
  return update_PA(w, Px_new, Px_new_idx, P_new_n, NULL, NULL, 0, 1, 0);
}
c_int osqp_update_A(OSQPWorkspace *w, const c_float *Ax_new, const c_int *Ax_new_idx, c_int A_new_n) {
  return update_PA(w, NULL, NULL, 0, Ax_new, Ax_new_idx, A_new_n, 0, 1);
}
c_int osqp_update_P_A(OSQPWorkspace *w, const c_float *Px_new, const c_int *Px_new_idx, c_int P_new_n,
                      const c_float *Ax_new, const c_int *Ax_new_idx, c_int A_new_n) {
  return update_PA(w, Px_new, Px_new_idx, P_new_n, Ax_new, Ax_new_idx, A_new_n, 1, 1);
}

/* [REF src/interface.jl:541] */
c_int osqp_update_rho(OSQPWorkspace *w, c_float rho_new) {
  if (!w) return 7;
  if (rho_new <= 0) return 1;
  int from_solve = PRIV(w)->rho_update_from_solve;
  if (!from_solve) begin_update(w);
  c_int i, m = w->data->m;
  w->settings->rho = c_minf(c_maxf(rho_new, RHO_MIN), RHO_MAX);
  for (i = 0; i < m; i++) {
    if (w->constr_type[i] == 0) { w->rho_vec[i] = w->settings->rho; w->rho_inv_vec[i] = 1.0 / w->settings->rho; }
    else if (w->constr_type[i] == 1) { w->rho_vec[i] = RHO_EQ_OVER_RHO_INEQ * w->settings->rho; w->rho_inv_vec[i] = 1.0 / w->rho_vec[i]; }
  }
  int e = linsys_update_rho(w);
  if (!from_solve) end_update(w);
  return e;
}

/* ---------------------------------------------------------------- warm start (A.7)
 * [REF src/interface.jl:709, 676, 690]; the single-vector forms reset the other
 * block to zero, as the reference notes at [REF src/modcaches.jl:196]. */
c_int osqp_warm_start(OSQPWorkspace *w, const c_float *x, const c_float *y) {
  if (!w) return 7;
  c_int n = w->data->n, m = w->data->m, i;
  if (!w->settings->warm_start) w->settings->warm_start = 1;
  memcpy(w->x, x, sizeof(c_float) * (size_t)n);
  memcpy(w->y, y, sizeof(c_float) * (size_t)m);
  if (w->settings->scaling) {
    for (i = 0; i < n; i++) w->x[i] *= SCAL(w)->Dinv[i];
    for (i = 0; i < m; i++) { w->y[i] *= SCAL(w)->Einv[i]; w->y[i] *= SCAL(w)->c; }
  }
  mat_vec(w->data->A, w->x, w->z, 0);
  return 0;
}

c_int osqp_warm_start_x(OSQPWorkspace *w, const c_float *x) {
  if (!w) return 7;
  c_int n = w->data->n, m = w->data->m, i;
  if (!w->settings->warm_start) w->settings->warm_start = 1;
  memcpy(w->x, x, sizeof(c_float) * (size_t)n);
  if (w->settings->scaling) for (i = 0; i < n; i++) w->x[i] *= SCAL(w)->Dinv[i];
  mat_vec(w->data->A, w->x, w->z, 0);
  memset(w->y, 0, sizeof(c_float) * (size_t)m);
  return 0;
}

c_int osqp_warm_start_y(OSQPWorkspace *w, const c_float *y) {
  if (!w) return 7;
  c_int n = w->data->n, m = w->data->m, i;
  if (!w->settings->warm_start) w->settings->warm_start = 1;
  memcpy(w->y, y, sizeof(c_float) * (size_t)m);
  if (w->settings->scaling) for (i = 0; i < m; i++) { w->y[i] *= SCAL(w)->Einv[i]; w->y[i] *= SCAL(w)->c; }
  memset(w->x, 0, sizeof(c_float) * (size_t)n);
  memset(w->z, 0, sizeof(c_float) * (size_t)m);
  return 0;
}

/* ---------------------------------------------------------------- settings updates (A.7)
 * [REF src/interface.jl:476-658] */
c_int osqp_update_max_iter(OSQPWorkspace *w, c_int v) { if (!w) return 7; if (v <= 0) return 1; w->settings->max_iter = v; return 0; }
c_int osqp_update_eps_abs(OSQPWorkspace *w, c_float v) { if (!w) return 7; if (v < 0.) return 1; w->settings->eps_abs = v; return 0; }
c_int osqp_update_eps_rel(OSQPWorkspace *w, c_float v) { if (!w) return 7; if (v < 0.) return 1; w->settings->eps_rel = v; return 0; }
c_int osqp_update_eps_prim_inf(OSQPWorkspace *w, c_float v) { if (!w) return 7; if (v < 0.) return 1; w->settings->eps_prim_inf = v; return 0; }
c_int osqp_update_eps_dual_inf(OSQPWorkspace *w, c_float v) { if (!w) return 7; if (v < 0.) return 1; w->settings->eps_dual_inf = v; return 0; }
c_int osqp_update_alpha(OSQPWorkspace *w, c_float v) { if (!w) return 7; if (v <= 0. || v >= 2.) return 1; w->settings->alpha = v; return 0; }
c_int osqp_update_delta(OSQPWorkspace *w, c_float v) { if (!w) return 7; if (v <= 0.) return 1; w->settings->delta = v; return 0; }
c_int osqp_update_polish(OSQPWorkspace *w, c_int v) { if (!w) return 7; if (v != 0 && v != 1) return 1; w->settings->polish = v; w->info->polish_time = 0.0; return 0; }
c_int osqp_update_polish_refine_iter(OSQPWorkspace *w, c_int v) { if (!w) return 7; if (v < 0) return 1; w->settings->polish_refine_iter = v; return 0; }
c_int osqp_update_verbose(OSQPWorkspace *w, c_int v) { if (!w) return 7; if (v != 0 && v != 1) return 1; w->settings->verbose = v; return 0; }
c_int osqp_update_scaled_termination(OSQPWorkspace *w, c_int v) { if (!w) return 7; if (v != 0 && v != 1) return 1; w->settings->scaled_termination = v; return 0; }
c_int osqp_update_check_termination(OSQPWorkspace *w, c_int v) { if (!w) return 7; if (v < 0) return 1; w->settings->check_termination = v; return 0; }
c_int osqp_update_warm_start(OSQPWorkspace *w, c_int v) { if (!w) return 7; if (v != 0 && v != 1) return 1; w->settings->warm_start = v; return 0; }
c_int osqp_update_time_limit(OSQPWorkspace *w, c_float v) { if (!w) return 7; if (v < 0.) return 1; w->settings->time_limit = v; return 0; }

/* ---------------------------------------------------------------- measurement hooks
 * (same names as the product's extensions so that bench.py drives both) */
c_int osqp_amd_get_stats(const OSQPWorkspace *w, c_float *out, c_int count) {
  c_float v[12] = {0};
  if (!w) return 0;
  c_int n = w->data->n, k, nnzPfull = 0;
  for (k = 0; k < n; k++) {
    c_int p;
    for (p = w->data->P->p[k]; p < w->data->P->p[k + 1]; p++) nnzPfull += (w->data->P->i[p] == k) ? 1 : 2;
  }
  v[0] = (c_float)((linsys_t *)w->linsys_solver)->kind;
  v[1] = (c_float)w->data->A->p[n];
  v[2] = (c_float)nnzPfull;
  v[3] = (c_float)w->data->P->p[n];
  v[4] = ((linsys_t *)w->linsys_solver)->kind == 0 ? (c_float)direct_nnzL(((linsys_t *)w->linsys_solver)->direct) : 0.0;
  v[6] = ((linsys_t *)w->linsys_solver)->kind == 2 ? (c_float)pcg_total_iters(((linsys_t *)w->linsys_solver)->pcg) : 0.0;
  v[7] = (c_float)((priv_t *)w->impl)->admm_iters_total;
  for (k = 0; k < count && k < 12; k++) out[k] = v[k];
  return k;
}

/* run exactly `iters` ADMM iterations from the current iterate */
c_int osqp_amd_iterate(OSQPWorkspace *w, c_int iters) {
  c_int it;
  tic(w);
  if (LIN(w)->kind == 2) pcg_set_guess(LIN(w)->pcg, w->x);
  for (it = 0; it < iters; it++) {
    c_float *t = w->x; w->x = w->x_prev; w->x_prev = t; t = w->z; w->z = w->z_prev; w->z_prev = t;
    admm_step(w);
    if (w->settings->check_termination && ((it + 1) % w->settings->check_termination == 0)) update_info(w, it + 1, 0, 0);
  }
  update_info(w, iters, 1, 0);
  return 0;
}

/* the current iterate in the caller's units (what store_solution would write), the iterate untouched */
c_int osqp_amd_get_iterate(OSQPWorkspace *w, c_float *x_out, c_float *y_out) {
  c_int n, m, i;
  if (!w) return 7;
  n = w->data->n; m = w->data->m;
  if (x_out) for (i = 0; i < n; i++) x_out[i] = w->settings->scaling ? w->x[i] * SCAL(w)->D[i] : w->x[i];
  if (y_out) for (i = 0; i < m; i++) y_out[i] = w->settings->scaling ? w->y[i] * (SCAL(w)->E[i] * SCAL(w)->cinv) : w->y[i];
  return 0;
}

const char *osqp_amd_last_error(void) { return ""; }
