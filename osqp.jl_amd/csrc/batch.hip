// batch.hip -- batched small-QP path (rows K11/K12 of SURVEY.md section 8a,
// BASELINE.json config 5: 4096 independent MPC QPs, n = 100, m = 200).
//
// One workgroup solves one QP from start to finish with everything in LDS:
//   * the instance's values of A (shared sparsity pattern, CSC order) and of the
//     full symmetric P, q, l, u, the Ruiz scalings, all ADMM iterates;
//   * the reduced KKT matrix M = P + sigma I + A' diag(rho) A (n x n, 80 KB at
//     n = 100; the 300 x 300 KKT matrix would not fit), inverted in place by
//     Gauss-Jordan sweeps and rebuilt whenever adaptive rho changes rho.
// Per iteration: b = sigma x - q + A'(rho z - y); x~ = M^-1 b (dense product,
// lane = row); z~ = A x~ consumed row by row by the x/z/y update; every `check_termination`
// iterations the same residual / infeasibility tests as the large-problem path.
// Same algorithm as oracle/osqp_oracle.c with the KKT system in reduced form.
// There is no communication between instances: the multi-GPU path shards the
// instance range over ranks and gathers the packed results once (batch.py).
#include <algorithm>
#include <cmath>

#include "engine.hpp"
#include "rng.hpp"

namespace oq {
namespace {

#ifndef OQ_BATCH_NT
#define OQ_BATCH_NT 512
#endif
constexpr int NT = OQ_BATCH_NT;  // threads per workgroup = per QP
constexpr int NW = NT / 64;
#define B_RHO_MIN 1e-6
#define B_RHO_MAX 1e6
#define B_MIN_SCALING 1e-4
#define B_MAX_SCALING 1e4
#define B_INF (OSQP_INFTY * B_MIN_SCALING)

struct Pattern {        // shared by all instances; device pointers
  int n, m, nnzA, nnzP, nnzF;
  const int *Ap, *Ai;               // A, CSC
  const int *Rp, *Rc, *Rmap;        // A, CSR; Rmap -> position in the CSC value array
  const int *Fp, *Fc, *Fmap;        // full symmetric P, CSR; Fmap -> position in the triu(P) value array
  // structure of A' diag(rho) A (lower triangle), pre-computed once for the shared pattern: non-zero pair t is
  // entry (Ti[t], Tj[t]) = sum over terms q in [Tp[t], Tp[t+1]) of rho[Tr[q]] * Av[Ta[q]] * Av[Tb[q]]
  int npair;
  const int *Tp;
  const unsigned short *Ti, *Tj, *Tr, *Ta, *Tb;
};

__device__ __forceinline__ double nmax(double a, double b) { return (a > b || a != a) ? a : b; }
__device__ __forceinline__ double lim(double v) { v = v < B_MIN_SCALING ? 1.0 : v; return v > B_MAX_SCALING ? B_MAX_SCALING : v; }

// K simultaneous block reductions (max for op 0, sum for op 1); result broadcast to every thread
template <int K>
__device__ __forceinline__ void block_reduce(double *v, int op, double *red) {
#pragma unroll
  for (int k = 0; k < K; k++) {
    double a = v[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { double b = __shfl_xor(a, o, 64); a = op ? a + b : nmax(a, b); }
    v[k] = a;
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0)
    for (int k = 0; k < K; k++) red[(threadIdx.x >> 6) * K + k] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; k++) {
    double a = red[k];
#pragma unroll
    for (int w = 1; w < NW; w++) a = op ? a + red[w * K + k] : nmax(a, red[w * K + k]);
    v[k] = a;
  }
}

struct Lds {
  double *M, *Av, *Pv, *q, *l, *u, *D, *E, *rho, *rhoi, *x, *z, *y, *xp, *zp, *xt, *zt, *dx, *dy, *Ax, *Px, *Aty, *tn, *tm, *red, *ldinv;
  int *ctype;
  unsigned short *Ap, *Ai, *Rp, *Rc, *Rmap, *Fp, *Fc;  // shared pattern, staged into LDS as 16-bit indices
  int ld;
};
__host__ __device__ inline size_t lds_doubles(int n, int m, int nnzA, int nnzF) {
  return (size_t)n * (n + 1) + (NT / 128) * (size_t)n + nnzA + nnzF + 10 * (size_t)n + 12 * (size_t)m + 16 * NW;
}
__host__ __device__ inline size_t lds_shorts(int n, int m, int nnzA, int nnzF) {
  return 2 * ((size_t)n + 1) + ((size_t)m + 1) + 3 * (size_t)nnzA + (size_t)nnzF + 8;
}
__host__ __device__ inline size_t lds_bytes(int n, int m, int nnzA, int nnzF) {
  return lds_doubles(n, m, nnzA, nnzF) * 8 + (((size_t)m * 4 + 15) / 16) * 16 + lds_shorts(n, m, nnzA, nnzF) * 2 + 16;
}
__device__ inline Lds carve(double *base, const Pattern &P) {
  Lds s;
  const int n = P.n, m = P.m;
  s.ld = n + 1;
  double *p = base;
  s.M = p; p += (size_t)n * s.ld + (NT / 128) * (size_t)n;  // the array + the partial sums of the dense product
  s.Av = p; p += P.nnzA; s.Pv = p; p += P.nnzF;
  s.q = p; p += n; s.D = p; p += n; s.x = p; p += n; s.xp = p; p += n; s.xt = p; p += n; s.dx = p; p += n;
  s.Px = p; p += n; s.Aty = p; p += n; s.tn = p; p += n; s.ldinv = p; p += n;
  s.l = p; p += m; s.u = p; p += m; s.E = p; p += m; s.rho = p; p += m; s.rhoi = p; p += m; s.z = p; p += m; s.y = p; p += m;
  s.zp = p; p += m; s.zt = p; p += m; s.dy = p; p += m; s.Ax = p; p += m; s.tm = p; p += m;
  s.red = p; p += 16 * NW;  // NW * K doubles of block_reduce, K <= 14
  s.ctype = (int *)p;
  unsigned short *h = (unsigned short *)((char *)p + (((size_t)m * 4 + 15) / 16) * 16);
  s.Ap = h; h += n + 1; s.Fp = h; h += n + 1; s.Rp = h; h += m + 1;
  s.Ai = h; h += P.nnzA; s.Rc = h; h += P.nnzA; s.Rmap = h; h += P.nnzA; s.Fc = h;
  return s;
}

// y = A x (CSR), y = A' x (CSC), y = P x (full symmetric CSR); no barriers inside.  L lanes share a row (the index ->
// value -> operand chain of LDS reads is latency-bound: 8 entries walked by one lane cost 8 round trips, by 4 lanes 2)
// and add up with xor shuffles, so every thread of the workgroup reaches the shuffles whether it has a row or not.
// finish(r, sum) runs on one lane per row.
template <int L, typename F, typename G>
__device__ __forceinline__ void rows_dot(int rows, const unsigned short *__restrict__ ptr, F term, G finish) {
  const int lane = threadIdx.x & (L - 1);
  for (int base = 0; base < rows; base += NT / L) {
    const int r = base + threadIdx.x / L;
    double a = 0.0;
    if (r < rows)
      for (int q = ptr[r] + lane; q < ptr[r + 1]; q += L) a += term(q);
#pragma unroll
    for (int o = L >> 1; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if (lane == 0 && r < rows) finish(r, a);
  }
}
__device__ __forceinline__ void mul_A(const Pattern &P, const Lds &s, const double *x, double *y) {
  rows_dot<2>(P.m, s.Rp, [&](int q) { return s.Av[s.Rmap[q]] * x[s.Rc[q]]; }, [&](int r, double a) { y[r] = a; });
}
__device__ __forceinline__ void mul_At(const Pattern &P, const Lds &s, const double *x, double *y) {
  rows_dot<4>(P.n, s.Ap, [&](int k) { return s.Av[k] * x[s.Ai[k]]; }, [&](int r, double a) { y[r] = a; });
}
__device__ __forceinline__ void mul_P(const Pattern &P, const Lds &s, const double *x, double *y) {
  rows_dot<4>(P.n, s.Fp, [&](int q) { return s.Pv[q] * x[s.Fc[q]]; }, [&](int r, double a) { y[r] = a; });
}

__device__ void set_rho(const Pattern &P, const Lds &s, double rho, bool classify) {
  for (int i = threadIdx.x; i < P.m; i += NT) {
    int t;
    if (classify) {
      if (s.l[i] < -B_INF && s.u[i] > B_INF) t = -1;
      else if (s.u[i] - s.l[i] < 1e-4) t = 1;
      else t = 0;
      s.ctype[i] = t;
    } else t = s.ctype[i];
    double r = t == -1 ? B_RHO_MIN : (t == 1 ? 1e3 * rho : rho);
    s.rho[i] = r; s.rhoi[i] = 1.0 / r;
  }
  __syncthreads();
}

// M = P + sigma I + A' diag(rho) A, then M <- M^-1 in place.  Returns false if M is not positive definite.
//
// The inverse is applied to thousands of right-hand sides (one per ADMM iteration) between two rho updates,
// and a dense product M^-1 b keeps every row independent while a triangular solve is a chain of 2n dependent
// steps; so the factorisation step is an explicit symmetric inversion by n Gauss-Jordan sweeps (Goodnight's
// sweep operator): every sweep is one rank-1 update of the whole n x n array, spread over the workgroup, with
// two barriers -- no serial column loop anywhere.  The pivots are the Schur complements of M, so the
// positive-definiteness test of the Cholesky factorisation carries over unchanged.
__device__ bool build_and_factor(const Pattern &P, const Lds &s, double sigma) {
  const int n = P.n, ld = s.ld;
  // A' rho A from the pre-computed term lists (the intersections of the columns of A do not depend on the instance)
  for (int e = threadIdx.x; e < n * ld; e += NT) s.M[e] = 0.0;
  __syncthreads();
  for (int t = threadIdx.x; t < P.npair; t += NT) {
    double acc = 0.0;
    for (int q = P.Tp[t]; q < P.Tp[t + 1]; q++) acc += s.rho[P.Tr[q]] * s.Av[P.Ta[q]] * s.Av[P.Tb[q]];
    const int i = P.Ti[t], j = P.Tj[t];
    s.M[i + j * ld] = acc;
    s.M[j + i * ld] = acc;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += NT) s.M[i + i * ld] += sigma;
  __syncthreads();
  for (int r = threadIdx.x; r < n; r += NT)
    for (int q = s.Fp[r]; q < s.Fp[r + 1]; q++) s.M[r + s.Fc[q] * ld] += s.Pv[q];  // full symmetric P: both triangles
  __syncthreads();
  bool ok = true;
  // Only the lower triangle (i >= j) is swept; the rest of the array is filled in by symmetry at the end.  Thread
  // layout: rows r and n-1-r form a pair with n+1 lower-triangle elements together, so every pair is the same
  // amount of work: pair = tid % PAIRS_PAD, position inside the pair p = tid / PAIRS_PAD, + PS, ...
  //   p <= r: element (r, p)          p > r: element (n-1-r, n-p)    (column independent of r: no bank conflicts)
  const int npairs = (n + 1) >> 1;  // for odd n the middle row pairs with itself and is taken once (p <= r only)
  constexpr int PP = 64;            // pairs handled side by side (lanes of a wavefront = consecutive pairs)
  constexpr int PS = NT / PP;       // positions handled side by side
  const int tr = threadIdx.x % PP, tp = threadIdx.x / PP;
  // Two pivots per pass (a block sweep on {k, k+1}: the same result as two single sweeps, with half the passes over
  // the array and half the barriers):  M_ij -= [c0_i c1_i] B^-1 [c0_j; c1_j],  columns k, k+1 <- [c0 c1] B^-1,
  // pivot block <- -B^-1, with B = [a b; b c] the 2 x 2 pivot block and c0, c1 the two columns.
  double *c0 = s.tn, *c1 = s.Px;  // scratch: Px is only live inside a residual evaluation
  int k = 0;
  for (; k + 1 < n; k += 2) {
    for (int i = threadIdx.x; i < n; i += NT) {
      c0[i] = i >= k ? s.M[i + k * ld] : s.M[k + i * ld];
      c1[i] = i >= k + 1 ? s.M[i + (k + 1) * ld] : s.M[k + 1 + i * ld];
    }
    __syncthreads();
    const double pa = c0[k], pb = c0[k + 1], pc = c1[k + 1];
    const double det = pa * pc - pb * pb;
    if (!(pa > 0.0) || !(det > 0.0)) ok = false;  // both pivots (a and c - b^2 / a) positive
    const double idet = 1.0 / det;
    for (int r = tr; r < npairs; r += PP) {
      const int r2 = n - 1 - r;
      const double g0a = (c0[r] * pc - c1[r] * pb) * idet, g1a = (c1[r] * pa - c0[r] * pb) * idet;
      const double g0b = (c0[r2] * pc - c1[r2] * pb) * idet, g1b = (c1[r2] * pa - c0[r2] * pb) * idet;
      const int last = (r2 == r) ? r : n;
      int q = tp;
      for (; q + 3 * PS <= last; q += 4 * PS) {  // 4 independent read-modify-writes in flight
        int idx[4];
        double mv[4], fv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int qq = q + u * PS;
          const bool first = qq <= r;
          const int i = first ? r : r2, j = first ? qq : n - qq;
          idx[u] = i + j * ld;
          mv[u] = s.M[idx[u]];
          fv[u] = (first ? g0a : g0b) * c0[j] + (first ? g1a : g1b) * c1[j];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) s.M[idx[u]] = mv[u] - fv[u];
      }
      for (; q <= last; q += PS) {
        const bool first = q <= r;
        const int i = first ? r : r2, j = first ? q : n - q;
        s.M[i + j * ld] -= (first ? g0a : g0b) * c0[j] + (first ? g1a : g1b) * c1[j];
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += NT) {
      if (i == k) {
        s.M[k + k * ld] = -pc * idet;
        s.M[k + 1 + k * ld] = pb * idet;
        s.M[k + 1 + (k + 1) * ld] = -pa * idet;
      } else if (i != k + 1) {
        const double v0 = (c0[i] * pc - c1[i] * pb) * idet, v1 = (c1[i] * pa - c0[i] * pb) * idet;
        if (i > k) { s.M[i + k * ld] = v0; s.M[i + (k + 1) * ld] = v1; }
        else { s.M[k + i * ld] = v0; s.M[k + 1 + i * ld] = v1; }
      }
    }
    __syncthreads();
  }
  for (; k < n; k++) {  // odd n: the last pivot alone
    // column k of the symmetric matrix: below the diagonal from column k, above it from row k
    for (int i = threadIdx.x; i < n; i += NT) s.tn[i] = i >= k ? s.M[i + k * ld] : s.M[k + i * ld];
    __syncthreads();
    const double d = s.tn[k];
    if (!(d > 0.0)) ok = false;
    const double p = 1.0 / d;
    // rank-1 update of the lower triangle (row and column k are overwritten right after)
    for (int r = tr; r < npairs; r += PP) {
      const int r2 = n - 1 - r;
      const double f1 = s.tn[r] * p, f2 = s.tn[r2] * p;
      const int last = (r2 == r) ? r : n;  // positions 0..last
      int q = tp;
      for (; q + 3 * PS <= last; q += 4 * PS) {  // 4 independent read-modify-writes in flight
        int idx[4];
        double mv[4], fv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int qq = q + u * PS;
          const bool first = qq <= r;
          const int i = first ? r : r2, j = first ? qq : n - qq;
          idx[u] = i + j * ld;
          mv[u] = s.M[idx[u]];
          fv[u] = (first ? f1 : f2) * s.tn[j];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) s.M[idx[u]] = mv[u] - fv[u];
      }
      for (; q <= last; q += PS) {
        const bool first = q <= r;
        const int i = first ? r : r2, j = first ? q : n - q;
        s.M[i + j * ld] -= (first ? f1 : f2) * s.tn[j];
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += NT) {
      const double v = (i == k) ? -p : s.tn[i] * p;
      if (i >= k) s.M[i + k * ld] = v; else s.M[k + i * ld] = v;
    }
    __syncthreads();
  }
  // negate and mirror: M^-1 as a full array for the row-wise products of the iterations
  for (int r = tr; r < npairs; r += PP) {
    const int r2 = n - 1 - r;
    const int last = (r2 == r) ? r : n;
    for (int q = tp; q <= last; q += PS) {
      const bool first = q <= r;
      const int i = first ? r : r2, j = first ? q : n - q;
      const double v = -s.M[i + j * ld];
      s.M[i + j * ld] = v;
      s.M[j + i * ld] = v;
    }
  }
  __syncthreads();
  return ok;
}


#ifdef OQ_BATCH_PROFILE
#define PROF_DECL long long pt0 = clock64(), pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PROF(k) { long long pt1 = clock64(); pacc[k] += pt1 - pt0; pt0 = pt1; }
#define PROF_PRINT if (inst == 0 && tid == 0) printf("cycles: load %lld scale %lld factor %lld rhs %lld solve %lld mulA+upd %lld check %lld rho %lld iters %d\n", pacc[0], pacc[1], pacc[2], pacc[3], pacc[4], pacc[5], pacc[6], pacc[7], iter);
#else
#define PROF_DECL
#define PROF(k)
#define PROF_PRINT
#endif

__global__ __launch_bounds__(NT) void k_batch_solve(Pattern P, OSQPSettings st, int count, const double *__restrict__ Px_all,
                                                    const double *__restrict__ Ax_all, const double *__restrict__ q_all,
                                                    const double *__restrict__ l_all, const double *__restrict__ u_all,
                                                    double *__restrict__ x_out, double *__restrict__ y_out,
                                                    double *__restrict__ info_out, int x_stride, int y_stride, int info_stride,
                                                    int info_cols) {
  extern __shared__ __attribute__((aligned(16))) double lds_raw[];
  const int inst = blockIdx.x;
  if (inst >= count) return;
  const int n = P.n, m = P.m, tid = threadIdx.x;
  Lds s = carve(lds_raw, P);
  PROF_DECL
  // ---- stage the shared pattern (16-bit) and load the instance -----------------
  for (int k = tid; k <= n; k += NT) { s.Ap[k] = (unsigned short)P.Ap[k]; s.Fp[k] = (unsigned short)P.Fp[k]; }
  for (int k = tid; k <= m; k += NT) s.Rp[k] = (unsigned short)P.Rp[k];
  for (int k = tid; k < P.nnzA; k += NT) { s.Ai[k] = (unsigned short)P.Ai[k]; s.Rc[k] = (unsigned short)P.Rc[k]; s.Rmap[k] = (unsigned short)P.Rmap[k]; }
  for (int k = tid; k < P.nnzF; k += NT) s.Fc[k] = (unsigned short)P.Fc[k];
  for (int k = tid; k < P.nnzA; k += NT) s.Av[k] = Ax_all[(size_t)inst * P.nnzA + k];
  for (int k = tid; k < P.nnzF; k += NT) s.Pv[k] = Px_all[(size_t)inst * P.nnzP + P.Fmap[k]];
  for (int j = tid; j < n; j += NT) { s.q[j] = q_all[(size_t)inst * n + j]; s.D[j] = 1.0; s.x[j] = 0.0; s.xp[j] = 0.0; s.dx[j] = 0.0; }
  for (int i = tid; i < m; i += NT) {
    s.l[i] = fmax(l_all[(size_t)inst * m + i], -OSQP_INFTY); s.u[i] = fmin(u_all[(size_t)inst * m + i], OSQP_INFTY);
    s.E[i] = 1.0; s.z[i] = 0.0; s.y[i] = 0.0; s.zp[i] = 0.0; s.dy[i] = 0.0;
  }
  __syncthreads();
  PROF(0)
  // ---- K0: Ruiz equilibration + cost scaling --------------------------------
  double c = 1.0;
  for (int it = 0; it < st.scaling; it++) {
    for (int j = tid; j < n; j += NT) {
      double mx = 0.0;
      for (int q = s.Fp[j]; q < s.Fp[j + 1]; q++) mx = fmax(mx, fabs(s.Pv[q]));
      for (int k = s.Ap[j]; k < s.Ap[j + 1]; k++) mx = fmax(mx, fabs(s.Av[k]));
      s.tn[j] = 1.0 / sqrt(lim(mx));
    }
    for (int i = tid; i < m; i += NT) {
      double mx = 0.0;
      for (int q = s.Rp[i]; q < s.Rp[i + 1]; q++) mx = fmax(mx, fabs(s.Av[s.Rmap[q]]));
      s.tm[i] = 1.0 / sqrt(lim(mx));
    }
    __syncthreads();
    for (int r = tid; r < n; r += NT)
      for (int q = s.Fp[r]; q < s.Fp[r + 1]; q++) {
        int cc = s.Fc[q];
        int lo = cc < r ? cc : r, hi = cc < r ? r : cc;
        s.Pv[q] = (s.Pv[q] * s.tn[lo]) * s.tn[hi];
      }
    for (int j = tid; j < n; j += NT) {
      for (int k = s.Ap[j]; k < s.Ap[j + 1]; k++) s.Av[k] = (s.Av[k] * s.tm[s.Ai[k]]) * s.tn[j];
      s.q[j] *= s.tn[j];
      s.D[j] *= s.tn[j];
    }
    for (int i = tid; i < m; i += NT) s.E[i] *= s.tm[i];
    __syncthreads();
    double v[2] = {0.0, 0.0}, w[1] = {0.0};
    for (int j = tid; j < n; j += NT) {
      double mx = 0.0;
      for (int q = s.Fp[j]; q < s.Fp[j + 1]; q++) mx = fmax(mx, fabs(s.Pv[q]));
      w[0] += mx;
      v[0] = fmax(v[0], fabs(s.q[j]));
    }
    block_reduce<1>(w, 1, s.red);
    block_reduce<2>(v, 0, s.red);
    double c_temp = w[0] / (double)n;
    c_temp = lim(fmax(c_temp, lim(v[0])));
    c_temp = 1.0 / c_temp;
    for (int k = tid; k < P.nnzF; k += NT) s.Pv[k] *= c_temp;
    for (int j = tid; j < n; j += NT) s.q[j] *= c_temp;
    c *= c_temp;
    __syncthreads();
  }
  const double cinv = 1.0 / c;
  for (int i = tid; i < m; i += NT) { s.l[i] *= s.E[i]; s.u[i] *= s.E[i]; }
  __syncthreads();
  PROF(1)
  // ---- K1, K2 ----------------------------------------------------------------
  double rho = fmin(fmax(st.rho, B_RHO_MIN), B_RHO_MAX);
  set_rho(P, s, rho, true);
  int status = OSQP_UNSOLVED;
  if (!build_and_factor(P, s, st.sigma)) status = OSQP_NON_CVX;
  PROF(2)
  const bool uns = st.scaling && !st.scaled_termination;
  const int check = (int)st.check_termination;
  const int rho_interval = st.adaptive_rho ? (st.adaptive_rho_interval ? (int)st.adaptive_rho_interval : 100) : 0;
  const double alpha = st.alpha, sigma = st.sigma;
  double pri_res = 0.0, dua_res = 0.0, obj = 0.0;
  double nrm[14];
  int iter = 0, rho_updates = 0;
  bool checked_last = false;
  double *x = s.x, *xp = s.xp, *z = s.z, *zp = s.zp;

  // residual evaluation (K8): fills nrm[], pri_res, dua_res, obj
  auto update_info = [&]() {
    mul_A(P, s, x, s.Ax); mul_P(P, s, x, s.Px); mul_At(P, s, s.y, s.Aty);
    __syncthreads();
    double v[14];
#pragma unroll
    for (int k = 0; k < 14; k++) v[k] = 0.0;
    double sm[2] = {0.0, 0.0};
    for (int i = tid; i < m; i += NT) {
      double ax = s.Ax[i], zi = z[i], e = 1.0 / s.E[i], r = ax - zi;
      v[0] = nmax(v[0], fabs(r)); v[1] = nmax(v[1], fabs(e * r)); v[2] = nmax(v[2], fabs(zi)); v[3] = nmax(v[3], fabs(ax));
      v[4] = nmax(v[4], fabs(e * zi)); v[5] = nmax(v[5], fabs(e * ax));
    }
    for (int j = tid; j < n; j += NT) {
      double px = s.Px[j], qj = s.q[j], at = s.Aty[j], d = 1.0 / s.D[j], xj = x[j], r = (qj + px) + at;
      v[6] = nmax(v[6], fabs(r)); v[7] = nmax(v[7], fabs(d * r)); v[8] = nmax(v[8], fabs(qj)); v[9] = nmax(v[9], fabs(at));
      v[10] = nmax(v[10], fabs(px)); v[11] = nmax(v[11], fabs(d * qj)); v[12] = nmax(v[12], fabs(d * at)); v[13] = nmax(v[13], fabs(d * px));
      sm[0] += xj * px; sm[1] += qj * xj;
    }
    block_reduce<14>(v, 0, s.red);
    block_reduce<2>(sm, 1, s.red);
#pragma unroll
    for (int k = 0; k < 14; k++) nrm[k] = v[k];
    pri_res = m == 0 ? 0.0 : (uns ? v[1] : v[0]);
    dua_res = uns ? cinv * v[7] : v[6];
    obj = cinv * (0.5 * sm[0] + sm[1]);
  };
  // termination tests; returns true when a status was set
  auto check_termination = [&](bool approx) -> bool {
    double ea = st.eps_abs, er = st.eps_rel, epi = st.eps_prim_inf, edi = st.eps_dual_inf;
    if (!(pri_res <= OSQP_INFTY) || !(dua_res <= OSQP_INFTY)) { status = OSQP_NON_CVX; return true; }
    if (approx) { ea *= 10; er *= 10; epi *= 10; edi *= 10; }
    bool pc = false, dc = false, pinf = false, dinf = false;
    if (m == 0) pc = true;
    else {
      double eps_p = ea + er * (uns ? nmax(nrm[4], nrm[5]) : nmax(nrm[2], nrm[3]));
      if (pri_res < eps_p) pc = true;
      else {  // primal infeasibility on delta_y
        double v[1] = {0.0}, sm[1] = {0.0};
        for (int i = tid; i < m; i += NT) {
          double d = s.dy[i];
          if (s.u[i] > B_INF) { if (s.l[i] < -B_INF) d = 0.0; else d = fmin(d, 0.0); }
          else if (s.l[i] < -B_INF) d = fmax(d, 0.0);
          s.dy[i] = d;
          v[0] = nmax(v[0], fabs(uns ? s.E[i] * d : d));
          sm[0] += s.u[i] * fmax(d, 0.0) + s.l[i] * fmin(d, 0.0);
        }
        block_reduce<1>(v, 0, s.red);
        block_reduce<1>(sm, 1, s.red);
        if (v[0] > epi && sm[0] < -epi * v[0]) {
          mul_At(P, s, s.dy, s.tn);
          __syncthreads();
          double w[1] = {0.0};
          for (int j = tid; j < n; j += NT) w[0] = nmax(w[0], fabs(uns ? s.tn[j] / s.D[j] : s.tn[j]));
          block_reduce<1>(w, 0, s.red);
          pinf = w[0] < epi * v[0];
        }
      }
    }
    double eps_d = ea + er * (uns ? cinv * nmax(nrm[11], nmax(nrm[12], nrm[13])) : nmax(nrm[8], nmax(nrm[9], nrm[10])));
    if (dua_res < eps_d) dc = true;
    else {  // dual infeasibility on delta_x
      double v[1] = {0.0}, sm[1] = {0.0};
      for (int j = tid; j < n; j += NT) { v[0] = nmax(v[0], fabs(uns ? s.D[j] * s.dx[j] : s.dx[j])); sm[0] += s.q[j] * s.dx[j]; }
      block_reduce<1>(v, 0, s.red);
      block_reduce<1>(sm, 1, s.red);
      double cs = uns ? c : 1.0;
      if (v[0] > edi && sm[0] < -cs * edi * v[0]) {
        mul_P(P, s, s.dx, s.tn);
        __syncthreads();
        double w[1] = {0.0};
        for (int j = tid; j < n; j += NT) w[0] = nmax(w[0], fabs(uns ? s.tn[j] / s.D[j] : s.tn[j]));
        block_reduce<1>(w, 0, s.red);
        if (w[0] < cs * edi * v[0]) {
          mul_A(P, s, s.dx, s.tm);
          __syncthreads();
          double bad[1] = {0.0};
          for (int i = tid; i < m; i += NT) {
            double a = uns ? s.tm[i] / s.E[i] : s.tm[i];
            if ((s.u[i] < B_INF && a > edi * v[0]) || (s.l[i] > -B_INF && a < -edi * v[0]) || a != a) bad[0] = 1.0;
          }
          block_reduce<1>(bad, 0, s.red);
          dinf = bad[0] == 0.0;
        }
      }
    }
    if (pc && dc) { status = approx ? OSQP_SOLVED_INACCURATE : OSQP_SOLVED; return true; }
    if (pinf) { status = approx ? OSQP_PRIMAL_INFEASIBLE_INACCURATE : OSQP_PRIMAL_INFEASIBLE; return true; }
    if (dinf) { status = approx ? OSQP_DUAL_INFEASIBLE_INACCURATE : OSQP_DUAL_INFEASIBLE; return true; }
    return false;
  };

  // ---- ADMM loop --------------------------------------------------------------
  if (status == OSQP_UNSOLVED) {
    const int max_iter = (int)st.max_iter;
    for (int i = tid; i < m; i += NT) s.zt[i] = s.rho[i] * z[i] - s.y[i];
    __syncthreads();
    for (iter = 1; iter <= max_iter; iter++) {
      { double *t = x; x = xp; xp = t; t = z; z = zp; zp = t; }
      // b = sigma x_prev - q + A'(rho z_prev - y); s.zt = rho z_prev - y was left behind by the previous z / y update
      rows_dot<4>(n, s.Ap, [&](int k) { return s.Av[k] * s.zt[s.Ai[k]]; },
                  [&](int j, double a) { s.xt[j] = sigma * xp[j] - s.q[j] + a; });
      __syncthreads();
      PROF(3)
      // x~ = M^-1 b: lane = row (consecutive lanes read consecutive LDS words: no bank conflicts, b[j] is a broadcast),
      // the columns cut into NT / 128 parts whose partial sums meet in LDS
      {
        constexpr int PARTS = NT / 128;
        const int part = tid >> 7;
        const int qn = (n + PARTS - 1) / PARTS;
        const int j0 = part * qn, j1 = (j0 + qn < n) ? j0 + qn : n;
        double *partial = s.M + (size_t)n * s.ld;  // PARTS * n doubles behind the array (see lds_doubles)
        for (int row = tid & 127; row < n; row += 128) {
          double a0 = 0.0, a1 = 0.0;
          int j = j0;
          for (; j + 1 < j1; j += 2) { a0 += s.M[row + j * s.ld] * s.xt[j]; a1 += s.M[row + (j + 1) * s.ld] * s.xt[j + 1]; }
          if (j < j1) a0 += s.M[row + j * s.ld] * s.xt[j];
          partial[part * n + row] = a0 + a1;
        }
        __syncthreads();
        for (int i = tid; i < n; i += NT) {
          double a = partial[i];
#pragma unroll
          for (int q = 1; q < PARTS; q++) a += partial[q * n + i];
          s.xt[i] = a;
        }
        __syncthreads();
      }
      PROF(4)
      // z~ = A x~ row by row, each row finished on the spot: z, y, delta_y and s.zt = rho z - y for the next right-hand side
      rows_dot<2>(m, s.Rp, [&](int q) { return s.Av[s.Rmap[q]] * s.xt[s.Rc[q]]; }, [&](int i, double zt) {
        const double zh = alpha * zt + (1.0 - alpha) * zp[i];
        const double yo = s.y[i];
        const double zn = fmin(fmax(zh + s.rhoi[i] * yo, s.l[i]), s.u[i]);
        z[i] = zn;
        const double d = s.rho[i] * (zh - zn);
        s.dy[i] = d; s.y[i] = yo + d;
        s.zt[i] = s.rho[i] * zn - (yo + d);
      });
      for (int j = tid; j < n; j += NT) { double xn = alpha * s.xt[j] + (1.0 - alpha) * xp[j]; x[j] = xn; s.dx[j] = xn - xp[j]; }
      __syncthreads();
      PROF(5)
      checked_last = check && (iter % check == 0);
      if (checked_last) { update_info(); if (check_termination(false)) break; }
      PROF(6)
      if (rho_interval && (iter % rho_interval == 0)) {
        if (!checked_last) update_info();
        double pr = m == 0 ? 0.0 : nrm[0] / (nmax(nrm[2], nrm[3]) + 1e-10);
        double du = nrm[6] / (nmax(nmax(nrm[8], nrm[9]), nrm[10]) + 1e-10);
        double est = fmin(fmax(rho * sqrt(pr / (du + 1e-10)), B_RHO_MIN), B_RHO_MAX);
        if (est > rho * st.adaptive_rho_tolerance || est < rho / st.adaptive_rho_tolerance) {
          rho = est; rho_updates++;
          set_rho(P, s, rho, false);
          for (int i = tid; i < m; i += NT) s.zt[i] = s.rho[i] * z[i] - s.y[i];  // the carried vector follows rho
          if (!build_and_factor(P, s, st.sigma)) { status = OSQP_NON_CVX; break; }
        }
      }
      PROF(7)
    }
    if (status == OSQP_UNSOLVED) {  // max_iter reached: last residual evaluation, then the 10x-relaxed tests
      iter = max_iter;
      if (!checked_last) { update_info(); check_termination(false); }
      if (status == OSQP_UNSOLVED && !check_termination(true)) status = OSQP_MAX_ITER_REACHED;
    }
  }
  PROF_PRINT
  // ---- store -----------------------------------------------------
  const bool has_sol = status == OSQP_SOLVED || status == OSQP_SOLVED_INACCURATE || status == OSQP_MAX_ITER_REACHED;
  for (int j = tid; j < n; j += NT) x_out[(size_t)inst * x_stride + j] = has_sol ? s.D[j] * x[j] : NAN;
  for (int i = tid; i < m; i += NT) y_out[(size_t)inst * y_stride + i] = has_sol ? cinv * s.E[i] * s.y[i] : NAN;
  if (tid == 0) {
    double *o = info_out + (size_t)inst * info_stride;
    o[0] = (double)iter; o[1] = (double)status; o[2] = pri_res; o[3] = dua_res;
    if (info_cols > 4) { o[4] = status == OSQP_NON_CVX ? NAN : obj; o[5] = (double)rho_updates; }
  }
}

// ---------------------------------------------------------------------------
// MPC instance generator (same statement as gen_mpc in oracle/gen.c); one
// thread fills one instance.  Also run on the host for instance 0 to obtain the
// shared sparsity pattern.
// ---------------------------------------------------------------------------
constexpr int NX = 6, NU = 4, TT = 10, NS = NX + NU, MPC_N = NS * TT, MPC_M = NX * TT + MPC_N + NU * TT;
__host__ __device__ inline int mpc_nnzA() {
  int c = 0;
  for (int t = 0; t < TT; t++) c += NX * (2 + (t + 1 < TT ? NX : 0)) + NU * (NX + 2 + (t + 1 < TT ? 1 : 0));
  return c;
}
__host__ __device__ inline void mpc_fill(long long inst, unsigned long long seed, int *Ap, int *Ai, double *Ax, double *Pd,
                                         double *q, double *l, double *u) {
  const int row_box = NX * TT, row_rate = NX * TT + MPC_N;
  double Ad[NX][NX], Bd[NX][NU], x0[NX], xref[NX];
  for (int r = 0; r < NX; r++) {
    for (int c = 0; c < NX; c++) {
      double base = (r == c ? 0.9 : 0.0) + ((r - c == 1 || c - r == 1) ? 0.05 : 0.0);
      Ad[r][c] = base + 0.02 * gauss(seed, G_MPC_A, (unsigned long long)(inst * 36 + r * 6 + c));
    }
    for (int c = 0; c < NU; c++)
      Bd[r][c] = ((r % 4) == c ? 0.5 : 0.0) + 0.1 * gauss(seed, G_MPC_B, (unsigned long long)(inst * 24 + r * 4 + c));
    x0[r] = gauss(seed, G_MPC_X0, (unsigned long long)(inst * 6 + r));
    xref[r] = 0.5 * gauss(seed, G_MPC_REF, (unsigned long long)(inst * 6 + r));
  }
  int pos = 0, j = 0;
  for (int t = 0; t < TT; t++) {
    for (int r = 0; r < NX; r++, j++) {
      Pd[j] = 1.0 + 0.1 * (double)r;
      q[j] = -(1.0 + 0.1 * (double)r) * xref[r];
      if (Ap) Ap[j] = pos;
      if (Ai) Ai[pos] = NX * t + r;
      Ax[pos++] = 1.0;
      if (t + 1 < TT) for (int c = 0; c < NX; c++) { if (Ai) Ai[pos] = NX * (t + 1) + c; Ax[pos++] = -Ad[c][r]; }
      if (Ai) Ai[pos] = row_box + j;
      Ax[pos++] = 1.0;
    }
    for (int c = 0; c < NU; c++, j++) {
      Pd[j] = 0.1;
      q[j] = 0.0;
      if (Ap) Ap[j] = pos;
      for (int r = 0; r < NX; r++) { if (Ai) Ai[pos] = NX * t + r; Ax[pos++] = -Bd[r][c]; }
      if (Ai) Ai[pos] = row_box + j;
      Ax[pos++] = 1.0;
      if (Ai) Ai[pos] = row_rate + NU * t + c;
      Ax[pos++] = 1.0;
      if (t + 1 < TT) { if (Ai) Ai[pos] = row_rate + NU * (t + 1) + c; Ax[pos++] = -1.0; }
    }
  }
  if (Ap) Ap[MPC_N] = pos;
  for (int r = 0; r < MPC_M; r++) { l[r] = 0.0; u[r] = 0.0; }
  for (int r = 0; r < NX; r++) {
    double sum = 0.0;
    for (int c = 0; c < NX; c++) sum += Ad[r][c] * x0[c];
    l[r] = sum; u[r] = sum;
  }
  for (int jj = 0; jj < MPC_N; jj++) {
    double b = (jj % NS) < NX ? 20.0 : 1.0;
    l[row_box + jj] = -b; u[row_box + jj] = b;
  }
  for (int r = 0; r < NU * TT; r++) { l[row_rate + r] = -0.5; u[row_rate + r] = 0.5; }
}
__global__ __launch_bounds__(64) void k_gen_mpc(long long first, int count, unsigned long long seed, int nnzA, double *Ax_all,
                                                double *Pd_all, double *q_all, double *l_all, double *u_all) {
  int t = blockIdx.x * 64 + threadIdx.x;
  if (t >= count) return;
  mpc_fill(first + t, seed, nullptr, nullptr, Ax_all + (size_t)t * nnzA, Pd_all + (size_t)t * MPC_N, q_all + (size_t)t * MPC_N,
           l_all + (size_t)t * MPC_M, u_all + (size_t)t * MPC_M);
}

// shared pattern on the device, built from host CSC patterns
struct DevicePattern {
  Pattern P;
  DevBuf<int> Ap, Ai, Rp, Rc, Rmap, Fp, Fc, Fmap, Tp;
  DevBuf<unsigned short> Ti, Tj, Tr, Ta, Tb;
  void build(int n, int m, const std::vector<int> &hPp, const std::vector<int> &hPi, const std::vector<int> &hAp,
             const std::vector<int> &hAi, hipStream_t s) {
    const int nnzA = hAp[n], nnzP = hPp[n];
    std::vector<int> rp(m + 1, 0), rc(nnzA), rmap(nnzA);
    for (int k = 0; k < nnzA; k++) rp[hAi[k] + 1]++;
    for (int i = 0; i < m; i++) rp[i + 1] += rp[i];
    std::vector<int> f(rp.begin(), rp.end() - 1);
    for (int j = 0; j < n; j++)
      for (int k = hAp[j]; k < hAp[j + 1]; k++) { int q = f[hAi[k]]++; rc[q] = j; rmap[q] = k; }
    // full symmetric pattern, rows sorted by column
    std::vector<std::vector<std::pair<int, int>>> rows(n);
    for (int j = 0; j < n; j++)
      for (int k = hPp[j]; k < hPp[j + 1]; k++) {
        int i = hPi[k];
        if (i > j) throw Error(1, "P is not upper triangular");
        rows[j].push_back({i, k});
        if (i != j) rows[i].push_back({j, k});
      }
    std::vector<int> fp(n + 1, 0), fc, fmap;
    for (int r = 0; r < n; r++) {
      std::sort(rows[r].begin(), rows[r].end());
      for (auto &e : rows[r]) { fc.push_back(e.first); fmap.push_back(e.second); }
      fp[r + 1] = (int)fc.size();
    }
    auto up = [&](DevBuf<int> &d, const std::vector<int> &h) { d.alloc(h.size()); d.upload(h.data(), h.size(), s); };
    up(Ap, hAp); up(Ai, hAi); up(Rp, rp); up(Rc, rc); up(Rmap, rmap); up(Fp, fp); up(Fc, fc); up(Fmap, fmap);
    HIP_CHECK(hipStreamSynchronize(s));
    // term lists of A' rho A: rows of A give the products, grouped by (i >= j) pair in ascending row order
    // (the order of the sparse dot product of columns i and j, so the sums are the ones the merge would form)
    std::vector<int> tp(1, 0);
    std::vector<unsigned short> ti, tj, tr, ta, tb;
    for (int i = 0; i < n; i++)
      for (int j = 0; j <= i; j++) {
        int a = hAp[i], ae = hAp[i + 1], b = hAp[j], be = hAp[j + 1], cnt = 0;
        while (a < ae && b < be) {
          if (hAi[a] == hAi[b]) { tr.push_back((unsigned short)hAi[a]); ta.push_back((unsigned short)a); tb.push_back((unsigned short)b); cnt++; a++; b++; }
          else if (hAi[a] < hAi[b]) a++; else b++;
        }
        if (cnt) { ti.push_back((unsigned short)i); tj.push_back((unsigned short)j); tp.push_back((int)tr.size()); }
      }
    auto up16 = [&](DevBuf<unsigned short> &d, const std::vector<unsigned short> &h) { d.alloc(h.size()); d.upload(h.data(), h.size(), s); };
    up(Tp, tp); up16(Ti, ti); up16(Tj, tj); up16(Tr, tr); up16(Ta, ta); up16(Tb, tb);
    HIP_CHECK(hipStreamSynchronize(s));
    P = Pattern{n, m, nnzA, nnzP, (int)fc.size(), Ap.get(), Ai.get(), Rp.get(), Rc.get(), Rmap.get(), Fp.get(), Fc.get(), Fmap.get(),
                (int)ti.size(), Tp.get(), Ti.get(), Tj.get(), Tr.get(), Ta.get(), Tb.get()};
  }
};

// outputs: row i of x / y / info at x + i * x_stride etc. (packed layouts put all three in one row); info_cols 4 or 6
void launch_batch(const DevicePattern &dp, const OSQPSettings &st, int count, const double *Px, const double *Ax, const double *q,
                  const double *l, const double *u, double *x, double *y, double *info, int x_stride, int y_stride, int info_stride,
                  int info_cols, hipStream_t s) {
  const Pattern &P = dp.P;
  size_t bytes = lds_bytes(P.n, P.m, P.nnzA, P.nnzF);
  if (P.n > 192 || P.m > 65535 || P.nnzA > 65535 || P.nnzF > 65535) throw Error(1, "the batched path supports n <= 192 and fewer than 65536 rows / non-zeros");
  if (bytes > 160 * 1024) throw Error(1, "instance too large for the LDS-resident batched path (needs " + std::to_string(bytes) + " bytes of LDS)");
  HIP_CHECK(hipFuncSetAttribute((const void *)k_batch_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  OQ_LAUNCH(k_batch_solve, dim3(count), dim3(NT), bytes, s, P, st, count, Px, Ax, q, l, u, x, y, info, x_stride, y_stride, info_stride,
            info_cols);
}

// A batch of MPC instances resident in HBM, cut into contiguous equal blocks over the ranks of a communicator
// (instance i -> rank floor(i / (total / world))).  solve() = this rank's block, one workgroup per
// instance, results written straight into their rows of the packed [total x (n + m + 4)] array, then the one
// collective of the path: an in-place all-gather of the rank blocks (rows K11 + K12 in one library call).
struct BatchPlan {
  int device = 0, total = 0, first = 0, count = 0;
  Comm *comm = nullptr;  // not owned; nullptr = one rank
  OSQPSettings st;
  DevicePattern dp;
  DevBuf<double> Px, Ax, q, l, u;
  static constexpr int kRow = MPC_N + MPC_M + 4;
};

}  // namespace
}  // namespace oq

using namespace oq;

extern "C" {

c_int osqp_amd_batch_solve(c_int count, c_int n, c_int m, const c_int *Pp, const c_int *Pi, const c_float *Px_all, const c_int *Ap,
                           const c_int *Ai, const c_float *Ax_all, const c_float *q_all, const c_float *l_all, const c_float *u_all,
                           const OSQPSettings *settings, c_float *x_out, c_float *y_out, OSQPInfo *info_out, c_int device) {
  try {
    // the same checks osqp_setup makes [REF src/interface.jl:47-100 + the C side's validate_data / validate_settings]
    if (count <= 0 || n <= 0 || m < 0 || !Pp || !Pi || !Ap || !Ai || !q_all || (m > 0 && (!l_all || !u_all))) { set_last_error("invalid batch data"); return 1; }
    if (validate_settings(settings)) { set_last_error("invalid settings"); return 2; }
    if (n > 192 || m > 65535 || Pp[0] != 0 || Ap[0] != 0 || Pp[n] < 0 || Ap[n] < 0 || Pp[n] > 65535 || Ap[n] > 65535) {
      set_last_error("the batched path supports n <= 192 and fewer than 65536 rows / non-zeros"); return 1;
    }
    for (c_int j = 0; j < n; j++) {
      if (Pp[j + 1] < Pp[j] || Ap[j + 1] < Ap[j]) { set_last_error("column pointers must not decrease"); return 1; }
      for (c_int k = Pp[j]; k < Pp[j + 1]; k++) if (Pi[k] < 0 || Pi[k] > j) { set_last_error("P must be upper triangular with row indices in range"); return 1; }
      for (c_int k = Ap[j]; k < Ap[j + 1]; k++) if (Ai[k] < 0 || Ai[k] >= m) { set_last_error("row index of A out of range"); return 1; }
    }
    for (c_int i = 0; i < count * m; i++) if (l_all[i] > u_all[i]) { set_last_error("lower bound greater than upper bound"); return 1; }
    DeviceScope on_device((int)device);
    hipStream_t s = nullptr;
    std::vector<int> hPp(Pp, Pp + n + 1), hAp(Ap, Ap + n + 1);
    std::vector<int> hPi(Pi, Pi + Pp[n]), hAi(Ai, Ai + Ap[n]);
    const int nnzA = hAp[n], nnzP = hPp[n];
    DevicePattern dp;
    dp.build((int)n, (int)m, hPp, hPi, hAp, hAi, s);
    DevBuf<double> dPx((size_t)count * nnzP), dAx((size_t)count * nnzA), dq((size_t)count * n), dl((size_t)count * m), du((size_t)count * m);
    DevBuf<double> dx((size_t)count * n), dy((size_t)count * m), dinfo((size_t)count * 6);
    dPx.upload(Px_all, (size_t)count * nnzP, s); dAx.upload(Ax_all, (size_t)count * nnzA, s);
    dq.upload(q_all, (size_t)count * n, s); dl.upload(l_all, (size_t)count * m, s); du.upload(u_all, (size_t)count * m, s);
    auto t0 = std::chrono::steady_clock::now();
    launch_batch(dp, *settings, (int)count, dPx.get(), dAx.get(), dq.get(), dl.get(), du.get(), dx.get(), dy.get(), dinfo.get(), (int)n,
                 (int)m, 6, 6, s);
    HIP_CHECK(hipDeviceSynchronize());
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::vector<double> hinfo((size_t)count * 6);
    dx.download(x_out, (size_t)count * n, s); dy.download(y_out, (size_t)count * m, s); dinfo.download(hinfo.data(), hinfo.size(), s);
    HIP_CHECK(hipDeviceSynchronize());
    for (c_int i = 0; i < count; i++) {
      OSQPInfo &o = info_out[i];
      memset(&o, 0, sizeof(OSQPInfo));
      o.iter = (c_int)hinfo[i * 6 + 0];
      update_status(&o, (c_int)hinfo[i * 6 + 1]);
      o.pri_res = hinfo[i * 6 + 2]; o.dua_res = hinfo[i * 6 + 3]; o.obj_val = hinfo[i * 6 + 4];
      o.rho_updates = (c_int)hinfo[i * 6 + 5];
      o.solve_time = secs; o.run_time = secs;
      o.rho_estimate = settings->rho;
    }
    return 0;
  } catch (const Error &er) {
    set_last_error(er.what());
    return er.code ? er.code : 6;
  } catch (const std::exception &ex) {
    set_last_error(ex.what());
    return 6;
  }
}

c_int osqp_amd_batch_solve_generated(c_int first, c_int count, unsigned long long seed, const OSQPSettings *settings, c_float *x_dev,
                                     c_float *y_dev, c_float *info_dev, c_int device) {
  try {
    if (count <= 0 || first < 0) return 1;
    if (validate_settings(settings)) { set_last_error("invalid settings"); return 2; }
    DeviceScope on_device((int)device);
    hipStream_t s = nullptr;
    const int nnzA = mpc_nnzA();
    // shared pattern from instance `first` on the host
    std::vector<int> hAp(MPC_N + 1), hAi(nnzA), hPp(MPC_N + 1), hPi(MPC_N);
    {
      std::vector<double> ax(nnzA), pd(MPC_N), q(MPC_N), l(MPC_M), u(MPC_M);
      mpc_fill(first, seed, hAp.data(), hAi.data(), ax.data(), pd.data(), q.data(), l.data(), u.data());
      for (int j = 0; j <= MPC_N; j++) hPp[j] = j;
      for (int j = 0; j < MPC_N; j++) hPi[j] = j;
    }
    DevicePattern dp;
    dp.build(MPC_N, MPC_M, hPp, hPi, hAp, hAi, s);
    DevBuf<double> dPx((size_t)count * MPC_N), dAx((size_t)count * nnzA), dq((size_t)count * MPC_N), dl((size_t)count * MPC_M),
        du((size_t)count * MPC_M);
    OQ_LAUNCH(k_gen_mpc, dim3(blocks_for(count, 64)), dim3(64), 0, s, (long long)first, (int)count, seed, nnzA, dAx.get(), dPx.get(),
              dq.get(), dl.get(), du.get());
    launch_batch(dp, *settings, (int)count, dPx.get(), dAx.get(), dq.get(), dl.get(), du.get(), x_dev, y_dev, info_dev, MPC_N, MPC_M, 4, 4, s);
    HIP_CHECK(hipDeviceSynchronize());
    return 0;
  } catch (const Error &er) {
    set_last_error(er.what());
    return er.code ? er.code : 6;
  } catch (const std::exception &ex) {
    set_last_error(ex.what());
    return 6;
  }
}

// ---- sharded MPC batch: K11 + K12 behind one handle -------------------------------------------------------
c_int osqp_amd_batch_mpc_create(osqp_amd_batch **out, c_int total, unsigned long long seed, const OSQPSettings *settings,
                                osqp_amd_comm *comm, c_int device) {
  if (!out) return 1;
  *out = nullptr;
  try {
    if (total <= 0) { set_last_error("empty batch"); return 1; }
    if (validate_settings(settings)) { set_last_error("invalid settings"); return 2; }
    Comm *c = (Comm *)comm;
    const int world = c ? c->world : 1, rank = c ? c->rank : 0;
    if (total % world != 0) { set_last_error("instance count must be divisible by the number of ranks"); return 1; }
    DeviceScope on_device((int)device);
    std::unique_ptr<BatchPlan> b(new BatchPlan());
    b->device = (int)device; b->total = (int)total; b->count = (int)(total / world); b->first = rank * b->count;
    b->comm = c; b->st = *settings;
    hipStream_t s = nullptr;
    const int nnzA = mpc_nnzA();
    std::vector<int> hAp(MPC_N + 1), hAi(nnzA), hPp(MPC_N + 1), hPi(MPC_N);
    {
      std::vector<double> ax(nnzA), pd(MPC_N), q(MPC_N), l(MPC_M), u(MPC_M);
      mpc_fill(0, seed, hAp.data(), hAi.data(), ax.data(), pd.data(), q.data(), l.data(), u.data());  // the pattern is the same for every instance
      for (int j = 0; j <= MPC_N; j++) hPp[j] = j;
      for (int j = 0; j < MPC_N; j++) hPi[j] = j;
    }
    b->dp.build(MPC_N, MPC_M, hPp, hPi, hAp, hAi, s);
    const size_t cnt = (size_t)b->count;
    b->Px.alloc(cnt * MPC_N); b->Ax.alloc(cnt * nnzA); b->q.alloc(cnt * MPC_N); b->l.alloc(cnt * MPC_M); b->u.alloc(cnt * MPC_M);
    OQ_LAUNCH(k_gen_mpc, dim3(blocks_for(b->count, 64)), dim3(64), 0, s, (long long)b->first, b->count, seed, nnzA, b->Ax.get(),
              b->Px.get(), b->q.get(), b->l.get(), b->u.get());
    HIP_CHECK(hipDeviceSynchronize());
    *out = (osqp_amd_batch *)b.release();
    return 0;
  } catch (const Error &er) {
    set_last_error(er.what());
    return er.code ? er.code : 6;
  } catch (const std::exception &ex) {
    set_last_error(ex.what());
    return 6;
  }
}

c_int osqp_amd_batch_mpc_solve(osqp_amd_batch *handle, c_float *packed_dev) {
  if (!handle || !packed_dev) return 1;
  BatchPlan &b = *(BatchPlan *)handle;
  try {
    DeviceScope on_device(b.device);
    hipStream_t s = nullptr;
    double *mine = packed_dev + (size_t)b.first * BatchPlan::kRow;
    launch_batch(b.dp, b.st, b.count, b.Px.get(), b.Ax.get(), b.q.get(), b.l.get(), b.u.get(), mine, mine + MPC_N, mine + MPC_N + MPC_M,
                 BatchPlan::kRow, BatchPlan::kRow, BatchPlan::kRow, 4, s);
    if (b.comm && b.comm->world > 1) b.comm->all_gather(packed_dev, (size_t)b.count * BatchPlan::kRow, s);
    HIP_CHECK(hipStreamSynchronize(s));
    return 0;
  } catch (const Error &er) {
    set_last_error(er.what());
    return er.code ? er.code : 6;
  } catch (const std::exception &ex) {
    set_last_error(ex.what());
    return 6;
  }
}

c_int osqp_amd_batch_destroy(osqp_amd_batch *handle) {
  if (!handle) return 0;
  BatchPlan *b = (BatchPlan *)handle;
  try { DeviceScope on_device(b->device); delete b; } catch (...) { return 1; }
  return 0;
}

}  // extern "C"
