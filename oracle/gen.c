/*
 * oracle/gen.c -- TEST INFRASTRUCTURE (see oracle.h).
 *
 * Host statement of the counter-based synthetic problem families of
 * SURVEY.md section 8d (BASELINE.json configs 2-5).  Every number is a pure
 * function of (seed, stream, index) through a SplitMix64 finaliser, integer
 * arithmetic and at most one floating-point multiply/add, so the device
 * generator (osqp.jl_amd/csrc/gen.hip) reproduces the same bits; the GPU tests
 * compare the two.  The reference's own random tests use Julia's RNG stream,
 * which is not portable even across Julia versions [REF test/update_matrices.jl:53-55].
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

enum { S_AROW = 1, S_AVAL = 2, S_UROW = 3, S_UVAL = 4, S_Q = 5, S_L = 6, S_U = 7, S_PDIAG = 8,
       S_MPC_A = 9, S_MPC_B = 10, S_MPC_X0 = 11, S_MPC_REF = 12 };

static inline uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

uint64_t oracle_rnd(uint64_t seed, uint64_t stream, uint64_t idx) {
  uint64_t k = mix64(seed + 0x9E3779B97F4A7C15ULL * (stream + 1));
  return mix64(k ^ (idx * 0xD1B54A32D192ED03ULL + 0x8CB92BA72F3D8DD7ULL));
}

double oracle_u01(uint64_t seed, uint64_t stream, uint64_t idx) {
  return ((double)(oracle_rnd(seed, stream, idx) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}

/* "Gaussian" = centred sum of four 16-bit uniforms, unit variance, as an exact
 * integer multiple of GAUSS_K (so that sums of magnitudes are order-independent). */
#define GAUSS_K (1.7320508075688772 / 65536.0)
static inline int64_t gauss_int(uint64_t r) {
  return (int64_t)((r & 0xFFFF) + ((r >> 16) & 0xFFFF) + ((r >> 32) & 0xFFFF) + ((r >> 48) & 0xFFFF)) - 131070;
}
double oracle_gauss(uint64_t seed, uint64_t stream, uint64_t idx) {
  return (double)gauss_int(oracle_rnd(seed, stream, idx)) * GAUSS_K;
}

static OSQPData *data_alloc(c_int n, c_int m, c_int nnzP, c_int nnzA) {
  OSQPData *d = (OSQPData *)calloc(1, sizeof(OSQPData));
  d->n = n; d->m = m;
  d->P = csc_alloc(n, n, nnzP);
  d->A = csc_alloc(m, n, nnzA);
  d->q = (c_float *)calloc((size_t)n, sizeof(c_float));
  d->l = (c_float *)calloc((size_t)(m > 0 ? m : 1), sizeof(c_float));
  d->u = (c_float *)calloc((size_t)(m > 0 ? m : 1), sizeof(c_float));
  return d;
}

/* ---- random sparse QP: n = m, A has `k` entries per column, P = U + U' + D ---- */
static OSQPData *gen_random_qp(c_int n, c_int k, uint64_t seed) {
  c_int m = n, j, t;
  c_int kp = k / 2 > 0 ? k / 2 : 1;
  if (k > m) k = m;
  /* nnz(triu P) = sum_j (min(kp, j) + 1) */
  c_int nnzP = 0;
  for (j = 0; j < n; j++) nnzP += (j < kp ? j : kp) + 1;
  OSQPData *d = data_alloc(n, m, nnzP, n * k);
  /* A */
  for (j = 0; j < n; j++) {
    d->A->p[j] = j * k;
    for (t = 0; t < k; t++) {
      c_int lo = (t * m) / k, hi = ((t + 1) * m) / k;
      uint64_t idx = (uint64_t)j * (uint64_t)k + (uint64_t)t;
      d->A->i[j * k + t] = lo + (c_int)(oracle_rnd(seed, S_AROW, idx) % (uint64_t)(hi - lo));
      d->A->x[j * k + t] = oracle_gauss(seed, S_AVAL, idx);
    }
  }
  d->A->p[n] = n * k;
  /* U and the integer magnitude sums for the dominant diagonal */
  int64_t *S = (int64_t *)calloc((size_t)n, sizeof(int64_t));
  c_int pos = 0;
  for (j = 0; j < n; j++) {
    d->P->p[j] = pos;
    c_int cnt = j < kp ? j : kp;
    for (t = 0; t < cnt; t++) {
      c_int row;
      if (j <= kp) row = t;
      else {
        c_int lo = (t * j) / kp, hi = ((t + 1) * j) / kp;
        row = lo + (c_int)(oracle_rnd(seed, S_UROW, (uint64_t)j * (uint64_t)kp + (uint64_t)t) % (uint64_t)(hi - lo));
      }
      int64_t I = gauss_int(oracle_rnd(seed, S_UVAL, (uint64_t)j * (uint64_t)kp + (uint64_t)t));
      d->P->i[pos] = row;
      d->P->x[pos] = (double)I * GAUSS_K;
      int64_t a = I < 0 ? -I : I;
      S[row] += a; S[j] += a;
      pos++;
    }
    d->P->i[pos] = j; /* diagonal, filled below */
    pos++;
  }
  d->P->p[n] = pos;
  for (j = 0; j < n; j++) d->P->x[d->P->p[j + 1] - 1] = 1.0 + (double)S[j] * GAUSS_K;
  free(S);
  for (j = 0; j < n; j++) d->q[j] = oracle_gauss(seed, S_Q, (uint64_t)j);
  for (j = 0; j < m; j++) {
    d->l[j] = -2.0 * oracle_u01(seed, S_L, (uint64_t)j);
    d->u[j] = 2.0 * oracle_u01(seed, S_U, (uint64_t)j);
  }
  return d;
}

/* ---- Lasso-as-QP: P diagonal, A = [I; -I], l = -inf, u = 0.1 ---- */
static OSQPData *gen_lasso(c_int n, uint64_t seed) {
  c_int m = 2 * n, j;
  OSQPData *d = data_alloc(n, m, n, 2 * n);
  for (j = 0; j < n; j++) {
    d->P->p[j] = j; d->P->i[j] = j; d->P->x[j] = 0.5 + oracle_u01(seed, S_PDIAG, (uint64_t)j);
    d->A->p[j] = 2 * j;
    d->A->i[2 * j] = j; d->A->x[2 * j] = 1.0;
    d->A->i[2 * j + 1] = n + j; d->A->x[2 * j + 1] = -1.0;
    d->q[j] = oracle_gauss(seed, S_Q, (uint64_t)j);
  }
  d->P->p[n] = n; d->A->p[n] = 2 * n;
  for (j = 0; j < m; j++) { d->l[j] = -OSQP_INFTY; d->u[j] = 0.1; }
  return d;
}

/* ---- MPC instance `inst`: nx=6, nu=4, T=10 -> n=100, m=200 ---- */
#define MPC_NX 6
#define MPC_NU 4
#define MPC_T 10
static OSQPData *gen_mpc(c_int inst, uint64_t seed) {
  const c_int nx = MPC_NX, nu = MPC_NU, T = MPC_T, ns = nx + nu;
  const c_int n = ns * T, m = nx * T + n + nu * T;
  const c_int row_box = nx * T, row_rate = nx * T + n;
  double Ad[MPC_NX][MPC_NX], Bd[MPC_NX][MPC_NU], x0[MPC_NX], xref[MPC_NX];
  c_int r, c, t;
  for (r = 0; r < nx; r++) {
    for (c = 0; c < nx; c++) {
      double base = (r == c ? 0.9 : 0.0) + ((r - c == 1 || c - r == 1) ? 0.05 : 0.0);
      Ad[r][c] = base + 0.02 * oracle_gauss(seed, S_MPC_A, (uint64_t)(inst * 36 + r * 6 + c));
    }
    for (c = 0; c < nu; c++)
      Bd[r][c] = ((r % 4) == c ? 0.5 : 0.0) + 0.1 * oracle_gauss(seed, S_MPC_B, (uint64_t)(inst * 24 + r * 4 + c));
    x0[r] = oracle_gauss(seed, S_MPC_X0, (uint64_t)(inst * 6 + r));
    xref[r] = 0.5 * oracle_gauss(seed, S_MPC_REF, (uint64_t)(inst * 6 + r));
  }
  /* nnz(A): state cols 2 + (t+1<T ? nx : 0); input cols nx + 2 + (t+1<T ? 1 : 0) */
  c_int nnzA = 0;
  for (t = 0; t < T; t++) nnzA += nx * (2 + (t + 1 < T ? nx : 0)) + nu * (nx + 2 + (t + 1 < T ? 1 : 0));
  OSQPData *d = data_alloc(n, m, n, nnzA);
  c_int pos = 0, j = 0;
  for (t = 0; t < T; t++) {
    for (r = 0; r < nx; r++, j++) { /* state x_{t+1}[r] */
      d->P->p[j] = j; d->P->i[j] = j; d->P->x[j] = 1.0 + 0.1 * (double)r;
      d->q[j] = -(1.0 + 0.1 * (double)r) * xref[r];
      d->A->p[j] = pos;
      d->A->i[pos] = nx * t + r; d->A->x[pos++] = 1.0;
      if (t + 1 < T) for (c = 0; c < nx; c++) { d->A->i[pos] = nx * (t + 1) + c; d->A->x[pos++] = -Ad[c][r]; }
      d->A->i[pos] = row_box + j; d->A->x[pos++] = 1.0;
    }
    for (c = 0; c < nu; c++, j++) { /* input u_t[c] */
      d->P->p[j] = j; d->P->i[j] = j; d->P->x[j] = 0.1;
      d->q[j] = 0.0;
      d->A->p[j] = pos;
      for (r = 0; r < nx; r++) { d->A->i[pos] = nx * t + r; d->A->x[pos++] = -Bd[r][c]; }
      d->A->i[pos] = row_box + j; d->A->x[pos++] = 1.0;
      d->A->i[pos] = row_rate + nu * t + c; d->A->x[pos++] = 1.0;
      if (t + 1 < T) { d->A->i[pos] = row_rate + nu * (t + 1) + c; d->A->x[pos++] = -1.0; }
    }
  }
  d->P->p[n] = n; d->A->p[n] = pos;
  for (r = 0; r < m; r++) { d->l[r] = 0.0; d->u[r] = 0.0; }
  for (r = 0; r < nx; r++) {
    double s = 0.0;
    for (c = 0; c < nx; c++) s += Ad[r][c] * x0[c];
    d->l[r] = s; d->u[r] = s;
  }
  for (j = 0; j < n; j++) {
    double b = (j % ns) < nx ? 20.0 : 1.0;
    d->l[row_box + j] = -b; d->u[row_box + j] = b;
  }
  for (r = 0; r < nu * T; r++) { d->l[row_rate + r] = -0.5; d->u[row_rate + r] = 0.5; }
  return d;
}

/* kind as enum osqp_amd_problem_kind; for MPC `per_row` is the instance index */
OSQPData *oracle_generate(c_int kind, c_int n, c_int per_row, unsigned long long seed) {
  if (kind == OSQP_AMD_GEN_RANDOM_QP) return gen_random_qp(n, per_row, seed);
  if (kind == OSQP_AMD_GEN_LASSO) return gen_lasso(n, seed);
  if (kind == OSQP_AMD_GEN_MPC) return gen_mpc(per_row, seed);
  return NULL;
}

void oracle_data_free(OSQPData *d) {
  if (!d) return;
  csc_free(d->P); csc_free(d->A); free(d->q); free(d->l); free(d->u); free(d);
}

c_int osqp_amd_setup_generated(OSQPWorkspace **workp, c_int kind, c_int n, c_int per_row,
                               unsigned long long seed, const OSQPSettings *settings) {
  OSQPData *d = oracle_generate(kind, n, per_row, seed);
  if (!d) return 1;
  c_int e = osqp_setup(workp, d, settings);
  oracle_data_free(d);
  return e;
}
