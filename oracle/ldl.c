/*
 * oracle/ldl.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * Direct back-end of the KKT solve: rows K2 (assembly, fill-reducing ordering,
 * elimination tree, numeric LDL^T), K3 (forward solve) and K4 (diagonal +
 * backward solve) of SURVEY.md section 8a.  In the reference this is
 * OSQP_jll's bundled QDLDL + AMD, reached through osqp_setup / osqp_solve /
 * osqp_update_{P,A,rho} [REF src/interface.jl:147, 171, 337, 358, 541].
 * Restated from the published descriptions (quasi-definite LDL^T, Vanderbei
 * 1995; up-looking sparse LDL^T and elimination trees, Davis, "Direct Methods
 * for Sparse Linear Systems", ch. 4; approximate minimum degree, Amestoy,
 * Davis, Duff 1996) -- not from the libosqp sources, which are unavailable.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

struct direct_solver {
  c_int n, m, N;
  c_float sigma;
  int polish;
  csc *K;            /* permuted upper-triangular KKT matrix */
  c_int *PtoK, *AtoK, *rhotoK, *sigtoK; /* nz maps into K->x */
  char *Phasdiag;
  c_int *perm;       /* perm[k] = original index of pivot k */
  c_int *etree, *Lnz, *Lp, *Li;
  c_float *Lx, *D, *Dinv;
  c_int *iwork; c_float *fwork;
  c_float *bp, *sol, *rho_inv;
  c_int nnzP, nnzA;
};

/* ------------------------------------------------------------------------ */
/* approximate minimum degree on a quotient graph (no supervariables)        */
/* ------------------------------------------------------------------------ */
static void degree_list_remove(c_int i, c_int *head, c_int *next, c_int *prev, const c_int *deg) {
  if (prev[i] >= 0) next[prev[i]] = next[i]; else head[deg[i]] = next[i];
  if (next[i] >= 0) prev[next[i]] = prev[i];
  next[i] = prev[i] = -1;
}
static void degree_list_insert(c_int i, c_int *head, c_int *next, c_int *prev, const c_int *deg) {
  c_int d = deg[i];
  prev[i] = -1; next[i] = head[d];
  if (head[d] >= 0) prev[head[d]] = i;
  head[d] = i;
}

/* Kp/Ki: upper-triangular pattern (entries with row < col are edges). */
static void min_degree_order(c_int N, const c_int *Kp, const c_int *Ki, c_int *perm) {
  c_int i, j, k, p, e, t;
  c_int *cnt = (c_int *)calloc((size_t)N + 1, sizeof(c_int));
  for (j = 0; j < N; j++)
    for (p = Kp[j]; p < Kp[j + 1]; p++) { i = Ki[p]; if (i < j) { cnt[i]++; cnt[j]++; } }
  /* nodes of very high degree (dense constraint rows) stay out of the quotient graph and are eliminated last:
     pruning their adjacency at every step would make the ordering quadratic */
  char *dense = (char *)calloc((size_t)N, 1);
  c_int *extra = (c_int *)calloc((size_t)N, sizeof(c_int)); /* dense neighbours: a constant part of the degree */
  c_int ndense = 0;
  {
    double limit = 10.0 * sqrt((double)N);
    if (limit < 16.0) limit = 16.0;
    for (i = 0; i < N; i++) if ((double)cnt[i] > limit) { dense[i] = 1; ndense++; }
    if (ndense) {
      memset(cnt, 0, sizeof(c_int) * ((size_t)N + 1));
      for (j = 0; j < N; j++)
        for (p = Kp[j]; p < Kp[j + 1]; p++) { i = Ki[p]; if (i < j && !dense[i] && !dense[j]) { cnt[i]++; cnt[j]++; } }
    }
  }
  c_int *ap = (c_int *)malloc(sizeof(c_int) * ((size_t)N + 1));
  ap[0] = 0;
  for (i = 0; i < N; i++) ap[i + 1] = ap[i] + cnt[i];
  c_int *adj = (c_int *)malloc(sizeof(c_int) * (size_t)(ap[N] > 0 ? ap[N] : 1));
  c_int *nv = (c_int *)calloc((size_t)N, sizeof(c_int)); /* variable neighbours (stored first) */
  c_int *ne = (c_int *)calloc((size_t)N, sizeof(c_int)); /* element neighbours (stored after)  */
  for (j = 0; j < N; j++)
    for (p = Kp[j]; p < Kp[j + 1]; p++) {
      i = Ki[p];
      if (i >= j) continue;
      if (!dense[i] && !dense[j]) { adj[ap[i] + nv[i]++] = j; adj[ap[j] + nv[j]++] = i; }
      else { if (!dense[i]) extra[i]++; if (!dense[j]) extra[j]++; }
    }
  c_int **Le = (c_int **)calloc((size_t)N, sizeof(c_int *));
  c_int *Lsz = (c_int *)calloc((size_t)N, sizeof(c_int));
  char *status = (char *)calloc((size_t)N, 1); /* 0 variable, 1 element, 2 dead */
  c_int *deg = (c_int *)malloc(sizeof(c_int) * (size_t)N);
  c_int *head = (c_int *)malloc(sizeof(c_int) * ((size_t)N + 1));
  c_int *next = (c_int *)malloc(sizeof(c_int) * (size_t)N);
  c_int *prev = (c_int *)malloc(sizeof(c_int) * (size_t)N);
  c_int *mark = (c_int *)calloc((size_t)N, sizeof(c_int));
  c_int *wmark = (c_int *)calloc((size_t)N, sizeof(c_int));
  c_int *w = (c_int *)calloc((size_t)N, sizeof(c_int));
  c_int *Lp = (c_int *)malloc(sizeof(c_int) * (size_t)(N > 0 ? N : 1));
  for (i = 0; i <= N; i++) head[i] = -1;
  for (i = 0; i < N; i++) { next[i] = prev[i] = -1; deg[i] = nv[i] + extra[i]; if (deg[i] > N - 1) deg[i] = N - 1; }
  for (i = N - 1; i >= 0; i--) degree_list_insert(i, head, next, prev, deg);
  c_int mindeg = 0, tag = 0;
  for (k = 0; k < N; k++) {
    while (mindeg < N && head[mindeg] < 0) mindeg++;
    p = head[mindeg];
    degree_list_remove(p, head, next, prev, deg);
    perm[k] = p;
    tag++;
    mark[p] = tag;
    c_int len = 0;
    for (t = 0; t < nv[p]; t++) {
      i = adj[ap[p] + t];
      if (status[i] == 0 && mark[i] != tag) { mark[i] = tag; Lp[len++] = i; }
    }
    for (t = 0; t < ne[p]; t++) {
      e = adj[ap[p] + nv[p] + t];
      if (status[e] != 1) continue;
      for (j = 0; j < Lsz[e]; j++) {
        i = Le[e][j];
        if (status[i] == 0 && mark[i] != tag) { mark[i] = tag; Lp[len++] = i; }
      }
      status[e] = 2; free(Le[e]); Le[e] = NULL; /* absorbed into p */
    }
    status[p] = 1;
    Lsz[p] = len;
    if (len > 0) {
      Le[p] = (c_int *)malloc(sizeof(c_int) * (size_t)len);
      memcpy(Le[p], Lp, sizeof(c_int) * (size_t)len);
    }
    /* pass 1: w[e] = |Le \ Lp| for every live element touching Lp */
    for (j = 0; j < len; j++) {
      i = Lp[j];
      for (t = 0; t < ne[i]; t++) {
        e = adj[ap[i] + nv[i] + t];
        if (status[e] != 1) continue;
        if (wmark[e] != tag) { wmark[e] = tag; w[e] = Lsz[e]; }
        w[e]--;
      }
    }
    /* pass 2: prune the lists of every i in Lp, approximate its degree */
    for (j = 0; j < len; j++) {
      i = Lp[j];
      degree_list_remove(i, head, next, prev, deg);
      c_int base = ap[i], nvn = 0, nen = 0, d = 0;
      c_int old_nv = nv[i], old_ne = ne[i];
      /* elements first into a scratch position: gather kept elements after kept variables */
      for (t = 0; t < old_nv; t++) {
        c_int v = adj[base + t];
        if (status[v] == 0 && mark[v] != tag) adj[base + nvn++] = v;
      }
      for (t = 0; t < old_ne; t++) {
        e = adj[base + old_nv + t];
        if (status[e] != 1) continue;
        if (w[e] == 0) { status[e] = 2; free(Le[e]); Le[e] = NULL; continue; } /* Le subset of Lp */
        adj[base + nvn + nen++] = e;
        d += w[e];
      }
      adj[base + nvn + nen++] = p;
      nv[i] = nvn; ne[i] = nen;
      d += nvn + (len - 1) + extra[i];
      c_int bound = deg[i] + (len - 1);
      if (d > bound) d = bound;
      if (d > N - 1) d = N - 1;
      if (d < 0) d = 0;
      deg[i] = d;
      degree_list_insert(i, head, next, prev, deg);
      if (d < mindeg) mindeg = d;
    }
  }
  if (ndense) { /* isolated in the pruned graph, so their position is free: move them to the end (stable) */
    c_int *tmp = (c_int *)malloc(sizeof(c_int) * (size_t)N);
    c_int a = 0;
    for (k = 0; k < N; k++) if (!dense[perm[k]]) tmp[a++] = perm[k];
    for (k = 0; k < N; k++) if (dense[perm[k]]) tmp[a++] = perm[k];
    memcpy(perm, tmp, sizeof(c_int) * (size_t)N);
    free(tmp);
  }
  free(dense); free(extra);
  for (i = 0; i < N; i++) if (Le[i]) free(Le[i]);
  free(cnt); free(ap); free(adj); free(nv); free(ne); free(Le); free(Lsz); free(status);
  free(deg); free(head); free(next); free(prev); free(mark); free(wmark); free(w); free(Lp);
}

/* ------------------------------------------------------------------------ */
/* KKT assembly                                                              */
/* ------------------------------------------------------------------------ */
static csc *assemble_kkt(direct_solver *s, const csc *P, const csc *A) {
  c_int n = s->n, m = s->m, N = s->N, j, k;
  c_int nnzP = P->p[n], nnzA = A->p[n];
  c_int *colcnt = (c_int *)calloc((size_t)N + 1, sizeof(c_int));
  s->Phasdiag = (char *)calloc((size_t)(n > 0 ? n : 1), 1);
  for (j = 0; j < n; j++) {
    for (k = P->p[j]; k < P->p[j + 1]; k++) if (P->i[k] == j) s->Phasdiag[j] = 1;
    colcnt[j] = (P->p[j + 1] - P->p[j]) + (s->Phasdiag[j] ? 0 : 1);
  }
  for (j = 0; j < m; j++) colcnt[n + j] = 1;
  for (k = 0; k < nnzA; k++) colcnt[n + A->i[k]]++;
  c_int total = 0;
  for (j = 0; j < N; j++) total += colcnt[j];
  csc *K = csc_alloc(N, N, total);
  K->p[0] = 0;
  for (j = 0; j < N; j++) K->p[j + 1] = K->p[j] + colcnt[j];
  c_int *fill = (c_int *)malloc(sizeof(c_int) * ((size_t)N + 1));
  memcpy(fill, K->p, sizeof(c_int) * ((size_t)N + 1));
  s->PtoK = (c_int *)malloc(sizeof(c_int) * (size_t)(nnzP > 0 ? nnzP : 1));
  s->AtoK = (c_int *)malloc(sizeof(c_int) * (size_t)(nnzA > 0 ? nnzA : 1));
  s->rhotoK = (c_int *)malloc(sizeof(c_int) * (size_t)(m > 0 ? m : 1));
  s->sigtoK = (c_int *)malloc(sizeof(c_int) * (size_t)(n > 0 ? n : 1));
  for (j = 0; j < n; j++) {
    for (k = P->p[j]; k < P->p[j + 1]; k++) {
      c_int q = fill[j]++;
      K->i[q] = P->i[k];
      K->x[q] = P->x[k];
      s->PtoK[k] = q;
      if (P->i[k] == j) { K->x[q] += s->sigma; s->sigtoK[j] = q; }
    }
    if (!s->Phasdiag[j]) {
      c_int q = fill[j]++;
      K->i[q] = j; K->x[q] = s->sigma; s->sigtoK[j] = q;
    }
  }
  for (j = 0; j < n; j++)
    for (k = A->p[j]; k < A->p[j + 1]; k++) {
      c_int col = n + A->i[k];
      c_int q = fill[col]++;
      K->i[q] = j; K->x[q] = A->x[k];
      s->AtoK[k] = q;
    }
  for (j = 0; j < m; j++) {
    c_int q = fill[n + j]++;
    K->i[q] = n + j;
    K->x[q] = s->rho_inv ? -s->rho_inv[j] : -s->sigma;
    s->rhotoK[j] = q;
  }
  free(fill); free(colcnt);
  return K;
}

/* symmetric permutation of an upper-triangular matrix; map_out[old nz] = new nz */
static csc *sym_permute(const csc *K, const c_int *pinv, c_int *map_out) {
  c_int N = K->n, j, p;
  c_int nnz = K->p[N];
  csc *C = csc_alloc(N, N, nnz);
  c_int *cnt = (c_int *)calloc((size_t)N + 1, sizeof(c_int));
  for (j = 0; j < N; j++)
    for (p = K->p[j]; p < K->p[j + 1]; p++) {
      c_int i2 = pinv[K->i[p]], j2 = pinv[j];
      cnt[i2 > j2 ? i2 : j2]++;
    }
  C->p[0] = 0;
  for (j = 0; j < N; j++) C->p[j + 1] = C->p[j] + cnt[j];
  memcpy(cnt, C->p, sizeof(c_int) * (size_t)N);
  for (j = 0; j < N; j++)
    for (p = K->p[j]; p < K->p[j + 1]; p++) {
      c_int i2 = pinv[K->i[p]], j2 = pinv[j];
      c_int col = i2 > j2 ? i2 : j2, row = i2 > j2 ? j2 : i2;
      c_int q = cnt[col]++;
      C->i[q] = row; C->x[q] = K->x[p];
      map_out[p] = q;
    }
  free(cnt);
  return C;
}

/* elimination tree and column counts of L */
static void symbolic(direct_solver *s) {
  c_int N = s->N, k, p, i;
  const csc *K = s->K;
  c_int *flag = s->iwork;
  for (k = 0; k < N; k++) {
    s->etree[k] = -1; flag[k] = k; s->Lnz[k] = 0;
    for (p = K->p[k]; p < K->p[k + 1]; p++) {
      i = K->i[p];
      if (i >= k) continue;
      for (; flag[i] != k; i = s->etree[i]) {
        if (s->etree[i] == -1) s->etree[i] = k;
        s->Lnz[i]++;
        flag[i] = k;
      }
    }
  }
  s->Lp[0] = 0;
  for (k = 0; k < N; k++) s->Lp[k + 1] = s->Lp[k] + s->Lnz[k];
}

/* up-looking numeric LDL^T; returns number of positive pivots, -1 on a zero pivot */
static c_int numeric(direct_solver *s) {
  c_int N = s->N, k, p, i, len, top, npos = 0;
  const csc *K = s->K;
  c_int *flag = s->iwork, *pattern = s->iwork + N, *lfill = s->iwork + 2 * N;
  c_float *y = s->fwork;
  for (k = 0; k < N; k++) { y[k] = 0.0; lfill[k] = 0; }
  for (k = 0; k < N; k++) {
    top = N; flag[k] = k;
    for (p = K->p[k]; p < K->p[k + 1]; p++) {
      i = K->i[p];
      if (i > k) continue;
      y[i] += K->x[p];
      for (len = 0; flag[i] != k; i = s->etree[i]) { pattern[len++] = i; flag[i] = k; }
      while (len > 0) pattern[--top] = pattern[--len];
    }
    c_float dk = y[k];
    y[k] = 0.0;
    for (; top < N; top++) {
      i = pattern[top];
      c_float yi = y[i];
      y[i] = 0.0;
      c_int p2 = s->Lp[i] + lfill[i];
      for (p = s->Lp[i]; p < p2; p++) y[s->Li[p]] -= s->Lx[p] * yi;
      c_float lki = yi * s->Dinv[i];
      dk -= lki * yi;
      s->Li[p2] = k; s->Lx[p2] = lki;
      lfill[i]++;
    }
    if (dk == 0.0 || dk != dk) return -1;
    s->D[k] = dk; s->Dinv[k] = 1.0 / dk;
    if (dk > 0.0) npos++;
  }
  return npos;
}

direct_solver *direct_init(const csc *P, const csc *A, c_float sigma, const c_float *rho_inv, int polish, int *err) {
  direct_solver *s = (direct_solver *)calloc(1, sizeof(direct_solver));
  c_int n = P->n, m = A->m, N = n + m, k;
  *err = 0;
  s->n = n; s->m = m; s->N = N; s->sigma = sigma; s->polish = polish;
  s->nnzP = P->p[n]; s->nnzA = A->p[n];
  if (rho_inv) {
    s->rho_inv = (c_float *)malloc(sizeof(c_float) * (size_t)(m > 0 ? m : 1));
    memcpy(s->rho_inv, rho_inv, sizeof(c_float) * (size_t)m);
  }
  csc *K0 = assemble_kkt(s, P, A);
  s->perm = (c_int *)malloc(sizeof(c_int) * (size_t)N);
  min_degree_order(N, K0->p, K0->i, s->perm);
  c_int *pinv = (c_int *)malloc(sizeof(c_int) * (size_t)N);
  for (k = 0; k < N; k++) pinv[s->perm[k]] = k;
  c_int *map = (c_int *)malloc(sizeof(c_int) * (size_t)(K0->p[N] > 0 ? K0->p[N] : 1));
  s->K = sym_permute(K0, pinv, map);
  for (k = 0; k < s->nnzP; k++) s->PtoK[k] = map[s->PtoK[k]];
  for (k = 0; k < s->nnzA; k++) s->AtoK[k] = map[s->AtoK[k]];
  for (k = 0; k < m; k++) s->rhotoK[k] = map[s->rhotoK[k]];
  for (k = 0; k < n; k++) s->sigtoK[k] = map[s->sigtoK[k]];
  free(map); free(pinv); csc_free(K0);
  s->etree = (c_int *)malloc(sizeof(c_int) * (size_t)N);
  s->Lnz = (c_int *)malloc(sizeof(c_int) * (size_t)N);
  s->Lp = (c_int *)malloc(sizeof(c_int) * ((size_t)N + 1));
  s->iwork = (c_int *)malloc(sizeof(c_int) * 3 * (size_t)N);
  s->fwork = (c_float *)malloc(sizeof(c_float) * (size_t)N);
  s->D = (c_float *)malloc(sizeof(c_float) * (size_t)N);
  s->Dinv = (c_float *)malloc(sizeof(c_float) * (size_t)N);
  s->bp = (c_float *)malloc(sizeof(c_float) * (size_t)N);
  s->sol = (c_float *)malloc(sizeof(c_float) * (size_t)N);
  symbolic(s);
  c_int nnzL = s->Lp[N];
  s->Li = (c_int *)malloc(sizeof(c_int) * (size_t)(nnzL > 0 ? nnzL : 1));
  s->Lx = (c_float *)malloc(sizeof(c_float) * (size_t)(nnzL > 0 ? nnzL : 1));
  c_int npos = numeric(s);
  if (npos < 0) { *err = 4; direct_free(s); return NULL; }
  if (npos != n) { *err = 5; direct_free(s); return NULL; }
  return s;
}

c_int direct_nnzL(const direct_solver *s) { return s->Lp[s->N]; }

static void ldl_solve(const direct_solver *s, const c_float *b, c_float *x) {
  c_int N = s->N, j, p;
  c_float *bp = s->bp;
  for (j = 0; j < N; j++) bp[j] = b[s->perm[j]];
  for (j = 0; j < N; j++) {
    c_float v = bp[j];
    for (p = s->Lp[j]; p < s->Lp[j + 1]; p++) bp[s->Li[p]] -= s->Lx[p] * v;
  }
  for (j = 0; j < N; j++) bp[j] *= s->Dinv[j];
  for (j = N - 1; j >= 0; j--) {
    c_float v = bp[j];
    for (p = s->Lp[j]; p < s->Lp[j + 1]; p++) v -= s->Lx[p] * bp[s->Li[p]];
    bp[j] = v;
  }
  for (j = 0; j < N; j++) x[s->perm[j]] = bp[j];
}

/* In place.  ADMM form (polish == 0): on entry b = [sigma x_prev - q ; z_prev - rho^-1 y],
 * on exit b = [x~ ; z~] with z~ = b_z + rho^-1 nu  (SURVEY.md A.2). */
void direct_solve(direct_solver *s, c_float *b) {
  c_int j;
  if (s->polish) { ldl_solve(s, b, b); return; }
  ldl_solve(s, b, s->sol);
  for (j = 0; j < s->n; j++) b[j] = s->sol[j];
  for (j = 0; j < s->m; j++) b[s->n + j] += s->rho_inv[j] * s->sol[s->n + j];
}

int direct_update_matrices(direct_solver *s, const csc *P, const csc *A) {
  c_int k;
  for (k = 0; k < s->nnzP; k++) s->K->x[s->PtoK[k]] = P->x[k];
  for (k = 0; k < s->n; k++) {
    if (s->Phasdiag[k]) s->K->x[s->sigtoK[k]] += s->sigma;
    else s->K->x[s->sigtoK[k]] = s->sigma;
  }
  for (k = 0; k < s->nnzA; k++) s->K->x[s->AtoK[k]] = A->x[k];
  c_int npos = numeric(s);
  if (npos < 0) return 4;
  if (npos != s->n) return 5;
  return 0;
}

int direct_update_rho(direct_solver *s, const c_float *rho_inv) {
  c_int k;
  for (k = 0; k < s->m; k++) { s->rho_inv[k] = rho_inv[k]; s->K->x[s->rhotoK[k]] = -rho_inv[k]; }
  c_int npos = numeric(s);
  if (npos < 0) return 4;
  if (npos != s->n) return 5;
  return 0;
}

void direct_free(direct_solver *s) {
  if (!s) return;
  csc_free(s->K);
  free(s->PtoK); free(s->AtoK); free(s->rhotoK); free(s->sigtoK); free(s->Phasdiag);
  free(s->perm); free(s->etree); free(s->Lnz); free(s->Lp); free(s->Li); free(s->Lx);
  free(s->D); free(s->Dinv); free(s->iwork); free(s->fwork); free(s->bp); free(s->sol); free(s->rho_inv);
  free(s);
}
