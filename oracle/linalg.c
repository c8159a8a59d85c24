/*
 * oracle/linalg.c -- TEST INFRASTRUCTURE (see oracle.h).
 * Compressed-column kernels the ADMM restatement needs: the operations that
 * SURVEY.md section 8a lists as K6 (A x, A' y), K7 (P x from the upper
 * triangle), the norms of K0/K8.  Reached in the reference through
 * osqp_setup / osqp_solve [REF src/interface.jl:147, 171].
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

csc *csc_alloc(c_int m, c_int n, c_int nzmax) {
  csc *A = (csc *)calloc(1, sizeof(csc));
  A->m = m; A->n = n; A->nzmax = nzmax; A->nz = -1;
  A->p = (c_int *)calloc((size_t)n + 1, sizeof(c_int));
  A->i = (c_int *)calloc((size_t)(nzmax > 0 ? nzmax : 1), sizeof(c_int));
  A->x = (c_float *)calloc((size_t)(nzmax > 0 ? nzmax : 1), sizeof(c_float));
  return A;
}

csc *csc_copy(const csc *A) {
  c_int nnz = A->p[A->n];
  csc *B = csc_alloc(A->m, A->n, nnz);
  memcpy(B->p, A->p, sizeof(c_int) * ((size_t)A->n + 1));
  if (nnz > 0) {
    memcpy(B->i, A->i, sizeof(c_int) * (size_t)nnz);
    memcpy(B->x, A->x, sizeof(c_float) * (size_t)nnz);
  }
  return B;
}

void csc_free(csc *A) {
  if (!A) return;
  free(A->p); free(A->i); free(A->x); free(A);
}

/* y = A x (plus_eq 0), y += A x (1), y -= A x (-1) */
void mat_vec(const csc *A, const c_float *x, c_float *y, int plus_eq) {
  c_int j, k;
  if (!plus_eq) for (j = 0; j < A->m; j++) y[j] = 0.0;
  if (A->p[A->n] == 0) return;
  if (plus_eq == -1) {
    for (j = 0; j < A->n; j++)
      for (k = A->p[j]; k < A->p[j + 1]; k++) y[A->i[k]] -= A->x[k] * x[j];
  } else {
    for (j = 0; j < A->n; j++)
      for (k = A->p[j]; k < A->p[j + 1]; k++) y[A->i[k]] += A->x[k] * x[j];
  }
}

/* y = A' x; skip_diag drops entries with row == column (used for the strictly
 * lower part of a symmetric matrix stored as its upper triangle) */
void mat_tpose_vec(const csc *A, const c_float *x, c_float *y, int plus_eq, int skip_diag) {
  c_int j, k;
  if (!plus_eq) for (j = 0; j < A->n; j++) y[j] = 0.0;
  if (A->p[A->n] == 0) return;
  for (j = 0; j < A->n; j++) {
    c_float acc = 0.0;
    for (k = A->p[j]; k < A->p[j + 1]; k++) {
      if (skip_diag && A->i[k] == j) continue;
      acc += A->x[k] * x[A->i[k]];
    }
    if (plus_eq == -1) y[j] -= acc; else y[j] += acc;
  }
}

c_float quad_form(const csc *P, const c_float *x) {
  c_float q = 0.0;
  c_int j, k;
  for (j = 0; j < P->n; j++)
    for (k = P->p[j]; k < P->p[j + 1]; k++) {
      c_int i = P->i[k];
      if (i == j) q += 0.5 * P->x[k] * x[i] * x[i];
      else if (i < j) q += P->x[k] * x[i] * x[j];
    }
  return q;
}

void mat_inf_norm_cols(const csc *M, c_float *E) {
  c_int j, k;
  for (j = 0; j < M->n; j++) {
    E[j] = 0.0;
    for (k = M->p[j]; k < M->p[j + 1]; k++) E[j] = fmax(fabs(M->x[k]), E[j]);
  }
}

void mat_inf_norm_rows(const csc *M, c_float *E) {
  c_int j, k;
  for (j = 0; j < M->m; j++) E[j] = 0.0;
  for (j = 0; j < M->n; j++)
    for (k = M->p[j]; k < M->p[j + 1]; k++) {
      c_int i = M->i[k];
      E[i] = fmax(fabs(M->x[k]), E[i]);
    }
}

/* column norms of the full symmetric matrix given its upper triangle */
void mat_inf_norm_cols_sym_triu(const csc *M, c_float *E) {
  c_int j, k;
  for (j = 0; j < M->n; j++) E[j] = 0.0;
  for (j = 0; j < M->n; j++)
    for (k = M->p[j]; k < M->p[j + 1]; k++) {
      c_int i = M->i[k];
      c_float a = fabs(M->x[k]);
      E[j] = fmax(a, E[j]);
      if (i != j) E[i] = fmax(a, E[i]);
    }
}

void mat_premult_diag(csc *A, const c_float *d) {
  c_int j, k;
  for (j = 0; j < A->n; j++)
    for (k = A->p[j]; k < A->p[j + 1]; k++) A->x[k] *= d[A->i[k]];
}

void mat_postmult_diag(csc *A, const c_float *d) {
  c_int j, k;
  for (j = 0; j < A->n; j++)
    for (k = A->p[j]; k < A->p[j + 1]; k++) A->x[k] *= d[j];
}

void mat_mult_scalar(csc *A, c_float sc) {
  c_int k, nnz = A->p[A->n];
  for (k = 0; k < nnz; k++) A->x[k] *= sc;
}

c_float vec_norm_inf(const c_float *v, c_int n) {
  c_float mx = 0.0;
  c_int i;
  for (i = 0; i < n; i++) { c_float a = fabs(v[i]); if (a > mx) mx = a; }
  return mx;
}

c_float vec_scaled_norm_inf(const c_float *S, const c_float *v, c_int n) {
  c_float mx = 0.0;
  c_int i;
  for (i = 0; i < n; i++) { c_float a = fabs(S[i] * v[i]); if (a > mx) mx = a; }
  return mx;
}

c_float vec_prod(const c_float *a, const c_float *b, c_int n) {
  c_float p = 0.0;
  c_int i;
  for (i = 0; i < n; i++) p += a[i] * b[i];
  return p;
}
