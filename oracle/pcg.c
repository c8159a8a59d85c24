/*
 * oracle/pcg.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * Indirect back-end of the KKT solve (row K9 of SURVEY.md section 8a; not part
 * of libosqp v0.6.2, needed because a direct factor of the random-sparsity
 * configs cannot fit in any memory -- SURVEY.md section 0.3).  Eliminating nu
 * from   [P + sigma I, A'; A, -diag(rho)^-1] [x~; nu] = [r_x; r_z]   gives
 *     (P + sigma I + A' diag(rho) A) x~ = r_x + A' (rho .* r_z),   z~ = A x~,
 * solved by Jacobi-preconditioned conjugate gradients.  Start vector: from the last
 * two solutions x1 (newest), x0 the point x1 + theta (x1 - x0) closest to the new
 * solution in the energy norm, theta = e'r / e'Me, e = x1 - x0, r = b - M x1 (clamped
 * to [-1, 4]; plain warm start x1 on the first solve after M changed or x~ was reset).
 * This is the CPU statement the HIP PCG path is checked against.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

struct pcg_solver {
  c_int n, m;
  c_float sigma;
  const csc *P, *A;       /* borrowed: the workspace's scaled matrices */
  c_float *rho;           /* copy of rho_vec */
  c_float *dinv;          /* inverse Jacobi diagonal */
  c_float *x, *r, *z, *p, *w, *t, *b1;
  c_float *x0, *w0;       /* previous solution and M times it */
  int have_prev;
  c_int total_iters, max_iter;
};

static void build_precond(pcg_solver *s) {
  c_int n = s->n, j, k;
  for (j = 0; j < n; j++) {
    c_float d = s->sigma;
    for (k = s->P->p[j]; k < s->P->p[j + 1]; k++) if (s->P->i[k] == j) d += s->P->x[k];
    for (k = s->A->p[j]; k < s->A->p[j + 1]; k++) d += s->rho[s->A->i[k]] * s->A->x[k] * s->A->x[k];
    s->dinv[j] = 1.0 / d;
  }
}

pcg_solver *pcg_init(const csc *P, const csc *A, c_float sigma, const c_float *rho_vec) {
  pcg_solver *s = (pcg_solver *)calloc(1, sizeof(pcg_solver));
  c_int n = P->n, m = A->m;
  size_t nn = (size_t)(n > 0 ? n : 1), mm = (size_t)(m > 0 ? m : 1);
  s->n = n; s->m = m; s->sigma = sigma; s->P = P; s->A = A;
  s->rho = (c_float *)malloc(sizeof(c_float) * mm);
  memcpy(s->rho, rho_vec, sizeof(c_float) * (size_t)m);
  s->dinv = (c_float *)malloc(sizeof(c_float) * nn);
  s->x = (c_float *)calloc(nn, sizeof(c_float));
  s->r = (c_float *)calloc(nn, sizeof(c_float));
  s->z = (c_float *)calloc(nn, sizeof(c_float));
  s->p = (c_float *)calloc(nn, sizeof(c_float));
  s->w = (c_float *)calloc(nn, sizeof(c_float));
  s->b1 = (c_float *)calloc(nn, sizeof(c_float));
  s->x0 = (c_float *)calloc(nn, sizeof(c_float));
  s->w0 = (c_float *)calloc(nn, sizeof(c_float));
  s->t = (c_float *)calloc(mm, sizeof(c_float));
  s->max_iter = 20000;
  build_precond(s);
  return s;
}

/* w = (P + sigma I + A' rho A) v */
static void apply_M(pcg_solver *s, const c_float *v, c_float *w) {
  c_int j;
  mat_vec(s->A, v, s->t, 0);
  for (j = 0; j < s->m; j++) s->t[j] *= s->rho[j];
  mat_vec(s->P, v, w, 0);
  mat_tpose_vec(s->P, v, w, 1, 1);
  for (j = 0; j < s->n; j++) w[j] += s->sigma * v[j];
  mat_tpose_vec(s->A, s->t, w, 1, 0);
}

c_int pcg_solve(pcg_solver *s, c_float *b, c_float tol_abs) {
  c_int n = s->n, m = s->m, j, it = 0;
  /* b1 = r_x + A' (rho .* r_z) */
  for (j = 0; j < m; j++) s->t[j] = s->rho[j] * b[n + j];
  for (j = 0; j < n; j++) s->b1[j] = b[j];
  mat_tpose_vec(s->A, s->t, s->b1, 1, 0);
  /* start vector and w = M x */
  apply_M(s, s->x, s->w);
  if (s->have_prev) {
    c_float num = 0.0, den = 0.0, theta;
    for (j = 0; j < n; j++) {
      c_float e = s->x[j] - s->x0[j];
      num += e * (s->b1[j] - s->w[j]);
      den += e * (s->w[j] - s->w0[j]);
    }
    theta = den > 0.0 ? num / den : 0.0;
    if (theta != theta) theta = 0.0;
    if (theta < -1.0) theta = -1.0;
    if (theta > 4.0) theta = 4.0;
    for (j = 0; j < n; j++) {
      c_float xc = s->x[j], wc = s->w[j];
      s->x[j] = xc + theta * (xc - s->x0[j]);
      s->w[j] = wc + theta * (wc - s->w0[j]);
      s->x0[j] = xc; s->w0[j] = wc;
    }
  } else {
    memcpy(s->x0, s->x, sizeof(c_float) * (size_t)n);
    memcpy(s->w0, s->w, sizeof(c_float) * (size_t)n);
    s->have_prev = 1;
  }
  /* r = b1 - M x_start */
  c_float rz = 0.0;
  for (j = 0; j < n; j++) {
    s->r[j] = s->b1[j] - s->w[j];
    s->z[j] = s->dinv[j] * s->r[j];
    s->p[j] = s->z[j];
    rz += s->r[j] * s->z[j];
  }
  c_int status = 0;
  /* EXPERIMENT (OSQP_ORACLE_PCG_SINGLE_REDUCTION=1; not the statement the HIP path is checked against): the recurrence of
   * Chronopoulos & Gear -- one product (of u = M^-1-preconditioned r, not of p) and ONE reduction point per iteration
   * (gamma = r'u and delta = w'u together) instead of two (p'w, then r'z).  On the device it would fold two vector
   * kernels of a CG iteration into one.  Used by tools/cg_recurrence_counts.py to answer what the recurrence costs in CG
   * iterations on the bench matrices (profiles/r04_cg_recurrence_counts.md). */
  {
    const char *sr = getenv("OSQP_ORACLE_PCG_SINGLE_REDUCTION");
    if (sr && atoi(sr) == 1) {
      c_float *u = s->z, *q = (c_float *)malloc(sizeof(c_float) * (size_t)(n > 0 ? n : 1));  /* q = M p, carried */
      c_float gamma = rz, gamma_old = 0.0, alpha = 0.0, delta;
      int first = 1;
      while (it < s->max_iter) {
        if (vec_norm_inf(s->r, n) <= tol_abs) break;
        apply_M(s, u, s->w);                       /* w = M u */
        delta = vec_prod(s->w, u, n);              /* the one reduction point: gamma (carried from the update) and delta */
        c_float beta = first ? 0.0 : gamma / gamma_old;
        c_float denom = first ? delta : delta - beta * gamma / alpha;
        if (!(denom > 0.0)) { status = -1; break; }
        alpha = gamma / denom;
        c_float gamma_new = 0.0;
        for (j = 0; j < n; j++) {
          s->p[j] = first ? u[j] : u[j] + beta * s->p[j];
          q[j] = first ? s->w[j] : s->w[j] + beta * q[j];
          s->x[j] += alpha * s->p[j];
          s->r[j] -= alpha * q[j];
        }
        for (j = 0; j < n; j++) { u[j] = s->dinv[j] * s->r[j]; gamma_new += s->r[j] * u[j]; }
        gamma_old = gamma; gamma = gamma_new; first = 0;
        it++;
      }
      free(q);
      s->total_iters += it;
      for (j = 0; j < n; j++) b[j] = s->x[j];
      mat_vec(s->A, s->x, b + n, 0);
      return status < 0 ? -1 - it : it;
    }
  }
  while (it < s->max_iter) {
    if (vec_norm_inf(s->r, n) <= tol_abs) break;
    apply_M(s, s->p, s->w);
    c_float pw = vec_prod(s->p, s->w, n);
    if (!(pw > 0.0)) { status = -1; break; }
    c_float alpha = rz / pw;
    c_float rz_new = 0.0;
    for (j = 0; j < n; j++) {
      s->x[j] += alpha * s->p[j];
      s->r[j] -= alpha * s->w[j];
      s->z[j] = s->dinv[j] * s->r[j];
      rz_new += s->r[j] * s->z[j];
    }
    c_float beta = rz_new / rz;
    rz = rz_new;
    for (j = 0; j < n; j++) s->p[j] = s->z[j] + beta * s->p[j];
    it++;
  }
  s->total_iters += it;
  for (j = 0; j < n; j++) b[j] = s->x[j];
  mat_vec(s->A, s->x, b + n, 0);
  return status < 0 ? -1 - it : it;
}

void pcg_update_matrices(pcg_solver *s, const csc *P, const csc *A) {
  s->P = P; s->A = A;
  s->have_prev = 0;
  build_precond(s);
}

void pcg_update_rho(pcg_solver *s, const c_float *rho_vec) {
  memcpy(s->rho, rho_vec, sizeof(c_float) * (size_t)s->m);
  s->have_prev = 0;
  build_precond(s);
}

c_int pcg_total_iters(const pcg_solver *s) { return s->total_iters; }

/* the CG start vector of the next solve (the ADMM loop sets it to the current x
 * at the start of every osqp_solve so that a solve depends on (x, z, y) only) */
void pcg_set_guess(pcg_solver *s, const c_float *x) {
  memcpy(s->x, x, sizeof(c_float) * (size_t)s->n);
  s->have_prev = 0;
}

void pcg_free(pcg_solver *s) {
  if (!s) return;
  free(s->rho); free(s->dinv); free(s->x); free(s->r); free(s->z); free(s->p); free(s->w); free(s->t); free(s->b1); free(s->x0); free(s->w0);
  free(s);
}
