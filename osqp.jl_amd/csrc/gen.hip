// gen.hip -- device generators of the synthetic problem families of SURVEY.md
// section 8d (BASELINE.json configs 2-5), written straight into HBM in the form
// the C ABI takes from a host caller (triu(P) and A in CSC, q, l, u).
//
// Every number is a pure function of (seed, stream, index): SplitMix64 finaliser,
// integer arithmetic, at most one fp64 multiply/add -- the same bits as the host
// statement in oracle/gen.c (tests/test_gpu_parity.py compares them).
#include "engine.hpp"
#include "rng.hpp"

namespace oq {

// ---- random sparse QP --------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_gen_A(long long n, long long m, long long k, unsigned long long seed,
                                                  int64_t *__restrict__ Ap, int *__restrict__ Ai, double *__restrict__ Ax) {
  long long idx = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (idx <= n) Ap[idx] = idx * k;
  if (idx >= n * k) return;
  long long t = idx % k;
  long long lo = (t * m) / k, hi = ((t + 1) * m) / k;
  Ai[idx] = (int)(lo + (long long)(rnd(seed, G_AROW, (unsigned long long)idx) % (unsigned long long)(hi - lo)));
  Ax[idx] = gauss(seed, G_AVAL, (unsigned long long)idx);
}
__device__ __forceinline__ long long triu_colptr(long long j, long long kp) {
  // sum_{j' < j} (min(kp, j') + 1)
  long long s = (j <= kp + 1) ? j * (j - 1) / 2 : kp * (kp + 1) / 2 + (j - kp - 1) * kp;
  return s + j;
}
// one wavefront per column j of the strictly upper part U; integer magnitude sums S (order independent)
__global__ __launch_bounds__(kBlock) void k_gen_U(long long n, long long kp, unsigned long long seed, int64_t *__restrict__ Pp,
                                                  int *__restrict__ Pi, double *__restrict__ Px,
                                                  unsigned long long *__restrict__ S) {
  const int lane = threadIdx.x & 63;
  const long long j = ((long long)blockIdx.x * kBlock + threadIdx.x) >> 6;
  if (j > n) return;
  const long long base = triu_colptr(j, kp);
  if (lane == 0) Pp[j] = base;
  if (j == n) return;
  const long long cnt = j < kp ? j : kp;
  unsigned long long colsum = 0;
  for (long long t = lane; t < cnt; t += 64) {
    long long row;
    if (j <= kp) row = t;
    else {
      long long lo = (t * j) / kp, hi = ((t + 1) * j) / kp;
      row = lo + (long long)(rnd(seed, G_UROW, (unsigned long long)(j * kp + t)) % (unsigned long long)(hi - lo));
    }
    long long I = gauss_int(rnd(seed, G_UVAL, (unsigned long long)(j * kp + t)));
    Pi[base + t] = (int)row;
    Px[base + t] = (double)I * OQ_GAUSS_K;
    unsigned long long a = (unsigned long long)(I < 0 ? -I : I);
    atomicAdd(&S[row], a);
    colsum += a;
  }
  for (int o = 32; o > 0; o >>= 1) colsum += __shfl_xor(colsum, o, 64);
  if (lane == 0) {
    atomicAdd(&S[j], colsum);
    Pi[base + cnt] = (int)j;
  }
}
__global__ __launch_bounds__(kBlock) void k_gen_diag(long long n, const int64_t *__restrict__ Pp, double *__restrict__ Px,
                                                     const unsigned long long *__restrict__ S) {
  long long j = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (j < n) Px[Pp[j + 1] - 1] = 1.0 + (double)(long long)S[j] * OQ_GAUSS_K;
}
__global__ __launch_bounds__(kBlock) void k_gen_vecs(long long n, long long m, unsigned long long seed, double *__restrict__ q,
                                                     double *__restrict__ l, double *__restrict__ u) {
  long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) q[i] = gauss(seed, G_Q, (unsigned long long)i);
  if (i < m) { l[i] = -2.0 * u01(seed, G_L, (unsigned long long)i); u[i] = 2.0 * u01(seed, G_U, (unsigned long long)i); }
}

// ---- Lasso-as-QP ---------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_gen_lasso(long long n, unsigned long long seed, int64_t *__restrict__ Pp,
                                                      int *__restrict__ Pi, double *__restrict__ Px, int64_t *__restrict__ Ap,
                                                      int *__restrict__ Ai, double *__restrict__ Ax, double *__restrict__ q,
                                                      double *__restrict__ l, double *__restrict__ u) {
  long long j = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (j > n) return;
  Pp[j] = j; Ap[j] = 2 * j;
  if (j == n) return;
  Pi[j] = (int)j; Px[j] = 0.5 + u01(seed, G_PDIAG, (unsigned long long)j);
  Ai[2 * j] = (int)j; Ax[2 * j] = 1.0;
  Ai[2 * j + 1] = (int)(n + j); Ax[2 * j + 1] = -1.0;
  q[j] = gauss(seed, G_Q, (unsigned long long)j);
  l[j] = -OSQP_INFTY; u[j] = 0.1;
  l[n + j] = -OSQP_INFTY; u[n + j] = 0.1;
}

// ---- the same families column range by column range (sharded setup, engine.hip setup_sharded) ------------------
// columns [j0, j1) of A: entry idx = j*k + t of the full problem, stored at idx - j0*k
__global__ __launch_bounds__(kBlock) void k_gen_A_cols(long long j0, long long j1, long long m, long long k, unsigned long long seed,
                                                       int64_t *__restrict__ Ap, int *__restrict__ Ai, double *__restrict__ Ax) {
  const long long loc = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (loc <= j1 - j0) Ap[loc] = loc * k;
  if (loc >= (j1 - j0) * k) return;
  const long long idx = j0 * k + loc, t = idx % k;
  const long long lo = (t * m) / k, hi = ((t + 1) * m) / k;
  Ai[loc] = (int)(lo + (long long)(rnd(seed, G_AROW, (unsigned long long)idx) % (unsigned long long)(hi - lo)));
  Ax[loc] = gauss(seed, G_AVAL, (unsigned long long)idx);
}
// columns [j0, j1) of triu(P), one wavefront per column; Pp == nullptr: only the magnitude sums S (the diagonal of
// column j needs the entries of row j, which live in later columns, so the sums take one pass over all columns first)
__global__ __launch_bounds__(kBlock) void k_gen_U_cols(long long j0, long long j1, long long kp, unsigned long long seed,
                                                       int64_t *__restrict__ Pp, int *__restrict__ Pi, double *__restrict__ Px,
                                                       unsigned long long *__restrict__ S) {
  const int lane = threadIdx.x & 63;
  const long long j = j0 + (((long long)blockIdx.x * kBlock + threadIdx.x) >> 6);
  if (j > j1) return;
  const long long base = triu_colptr(j, kp) - triu_colptr(j0, kp);
  if (Pp && lane == 0) Pp[j - j0] = base;
  if (j == j1) return;
  const long long cnt = j < kp ? j : kp;
  unsigned long long colsum = 0;
  for (long long t = lane; t < cnt; t += 64) {
    long long row;
    if (j <= kp) row = t;
    else {
      long long lo = (t * j) / kp, hi = ((t + 1) * j) / kp;
      row = lo + (long long)(rnd(seed, G_UROW, (unsigned long long)(j * kp + t)) % (unsigned long long)(hi - lo));
    }
    const long long I = gauss_int(rnd(seed, G_UVAL, (unsigned long long)(j * kp + t)));
    if (Pp) { Pi[base + t] = (int)row; Px[base + t] = (double)I * OQ_GAUSS_K; }
    else {
      const unsigned long long a = (unsigned long long)(I < 0 ? -I : I);
      atomicAdd(&S[row], a);
      colsum += a;
    }
  }
  if (!Pp) {
    for (int o = 32; o > 0; o >>= 1) colsum += __shfl_xor(colsum, o, 64);
    if (lane == 0) atomicAdd(&S[j], colsum);
  } else if (lane == 0) {
    Pi[base + cnt] = (int)j;
    Px[base + cnt] = 1.0 + (double)(long long)S[j] * OQ_GAUSS_K;
  }
}
__global__ __launch_bounds__(kBlock) void k_gen_lasso_cols(long long j0, long long j1, long long n, unsigned long long seed,
                                                           int64_t *__restrict__ Pp, int *__restrict__ Pi, double *__restrict__ Px,
                                                           int64_t *__restrict__ Ap, int *__restrict__ Ai, double *__restrict__ Ax) {
  const long long loc = (long long)blockIdx.x * kBlock + threadIdx.x, j = j0 + loc;
  if (j > j1) return;
  if (Pp) Pp[loc] = loc;
  if (Ap) Ap[loc] = 2 * loc;
  if (j == j1) return;
  if (Pp) { Pi[loc] = (int)j; Px[loc] = 0.5 + u01(seed, G_PDIAG, (unsigned long long)j); }
  if (Ap) { Ai[2 * loc] = (int)j; Ax[2 * loc] = 1.0; Ai[2 * loc + 1] = (int)(n + j); Ax[2 * loc + 1] = -1.0; }
}
__global__ __launch_bounds__(kBlock) void k_gen_lasso_vecs(long long n, unsigned long long seed, double *__restrict__ q,
                                                           double *__restrict__ l, double *__restrict__ u) {
  const long long j = (long long)blockIdx.x * kBlock + threadIdx.x;
  if (j >= n) return;
  q[j] = gauss(seed, G_Q, (unsigned long long)j);
  l[j] = -OSQP_INFTY; u[j] = 0.1;
  l[n + j] = -OSQP_INFTY; u[n + j] = 0.1;
}

struct GeneratedColumns : ColumnSource {
  int kind;
  long long k = 0, kp = 0;
  unsigned long long seed;
  DevBuf<unsigned long long> S;  // magnitude sums of the random family's P (its diagonal)
  GeneratedColumns(int kind_, int n_, int per_row, unsigned long long seed_, hipStream_t s) : kind(kind_), seed(seed_) {
    n = n_;
    if (kind == OSQP_AMD_GEN_RANDOM_QP) {
      m = n; k = per_row;
      if (k > m) k = m;
      if (k < 1) throw Error(1, "per_row must be >= 1");
      kp = k / 2 > 0 ? k / 2 : 1;
      const long long j = n;
      nnzP = ((j <= kp + 1) ? j * (j - 1) / 2 : kp * (kp + 1) / 2 + (j - kp - 1) * kp) + j;
      nnzA = (long long)n * k;
      S.alloc((size_t)n); S.zero(s);
      OQ_LAUNCH(k_gen_U_cols, dim3(blocks_for(((long long)n + 1) * 64)), dim3(kBlock), 0, s, 0LL, (long long)n, kp, seed,
                (int64_t *)nullptr, (int *)nullptr, (double *)nullptr, S.get());
      HIP_CHECK(hipStreamSynchronize(s));  // S is read by launches on the engine's (non-blocking) stream later on
    } else if (kind == OSQP_AMD_GEN_LASSO) {
      m = 2 * n; nnzP = n; nnzA = 2LL * n;
    } else throw Error(1, "unknown problem kind for the device generator");
    if (nnzA >= 2147483647LL || 2 * nnzP >= 2147483647LL) throw Error(6, "matrix too large: more than 2^31-1 non-zeros");
  }
  int64_t P_chunk(int j0, int j1, DevBuf<int64_t> &p, DevBuf<int> &i, DevBuf<double> &x, hipStream_t s) override {
    const long long cols = j1 - j0;
    p.alloc((size_t)cols + 1);
    if (kind == OSQP_AMD_GEN_LASSO) {
      i.alloc((size_t)cols); x.alloc((size_t)cols);
      OQ_LAUNCH(k_gen_lasso_cols, dim3(blocks_for(cols + 1)), dim3(kBlock), 0, s, (long long)j0, (long long)j1, (long long)n, seed, p.get(),
                i.get(), x.get(), (int64_t *)nullptr, (int *)nullptr, (double *)nullptr);
      return cols;
    }
    auto colptr = [&](long long j) { return ((j <= kp + 1) ? j * (j - 1) / 2 : kp * (kp + 1) / 2 + (j - kp - 1) * kp) + j; };
    const int64_t cnt = colptr(j1) - colptr(j0);
    i.alloc((size_t)cnt); x.alloc((size_t)cnt);
    OQ_LAUNCH(k_gen_U_cols, dim3(blocks_for((cols + 1) * 64)), dim3(kBlock), 0, s, (long long)j0, (long long)j1, kp, seed, p.get(), i.get(),
              x.get(), S.get());
    return cnt;
  }
  int64_t A_chunk(int j0, int j1, DevBuf<int64_t> &p, DevBuf<int> &i, DevBuf<double> &x, hipStream_t s) override {
    const long long cols = j1 - j0;
    p.alloc((size_t)cols + 1);
    if (kind == OSQP_AMD_GEN_LASSO) {
      i.alloc((size_t)(2 * cols)); x.alloc((size_t)(2 * cols));
      OQ_LAUNCH(k_gen_lasso_cols, dim3(blocks_for(cols + 1)), dim3(kBlock), 0, s, (long long)j0, (long long)j1, (long long)n, seed,
                (int64_t *)nullptr, (int *)nullptr, (double *)nullptr, p.get(), i.get(), x.get());
      return 2 * cols;
    }
    i.alloc((size_t)(cols * k)); x.alloc((size_t)(cols * k));
    OQ_LAUNCH(k_gen_A_cols, dim3(blocks_for(std::max<long long>(cols * k, cols + 1))), dim3(kBlock), 0, s, (long long)j0, (long long)j1,
              (long long)m, k, seed, p.get(), i.get(), x.get());
    return cols * k;
  }
  void vectors(DevBuf<double> &q, DevBuf<double> &l, DevBuf<double> &u, hipStream_t s) override {
    q.alloc(n); l.alloc(m); u.alloc(m);
    if (kind == OSQP_AMD_GEN_LASSO)
      OQ_LAUNCH(k_gen_lasso_vecs, dim3(blocks_for(n)), dim3(kBlock), 0, s, (long long)n, seed, q.get(), l.get(), u.get());
    else
      OQ_LAUNCH(k_gen_vecs, dim3(blocks_for(std::max(n, m))), dim3(kBlock), 0, s, (long long)n, (long long)m, seed, q.get(), l.get(), u.get());
  }
};
std::unique_ptr<ColumnSource> generated_columns(int kind, int n, int per_row, unsigned long long seed, hipStream_t s) {
  return std::unique_ptr<ColumnSource>(new GeneratedColumns(kind, n, per_row, seed, s));
}

void generate_problem(int kind, int n, int per_row, unsigned long long seed, hipStream_t s, int &n_out, int &m_out,
                      DevBuf<int64_t> &Pp, DevBuf<int> &Pi, DevBuf<double> &Px, DevBuf<int64_t> &Ap, DevBuf<int> &Ai,
                      DevBuf<double> &Ax, DevBuf<double> &q, DevBuf<double> &l, DevBuf<double> &u) {
  if (kind == OSQP_AMD_GEN_RANDOM_QP) {
    long long m = n, k = per_row;
    if (k > m) k = m;
    if (k < 1) throw Error(1, "per_row must be >= 1");
    long long kp = k / 2 > 0 ? k / 2 : 1;
    long long nnzP = 0;
    {
      long long j = n;
      nnzP = ((j <= kp + 1) ? j * (j - 1) / 2 : kp * (kp + 1) / 2 + (j - kp - 1) * kp) + j;
    }
    long long nnzA = (long long)n * k;
    if (nnzA >= 2147483647LL || 2 * nnzP >= 2147483647LL) throw Error(6, "matrix too large: more than 2^31-1 non-zeros");
    n_out = n; m_out = (int)m;
    Pp.alloc((size_t)n + 1); Pi.alloc((size_t)nnzP); Px.alloc((size_t)nnzP);
    Ap.alloc((size_t)n + 1); Ai.alloc((size_t)nnzA); Ax.alloc((size_t)nnzA);
    q.alloc(n); l.alloc(m); u.alloc(m);
    DevBuf<unsigned long long> S((size_t)n);
    S.zero(s);
    OQ_LAUNCH(k_gen_A, dim3(blocks_for(std::max<long long>(nnzA, n + 1))), dim3(kBlock), 0, s, (long long)n, m, k, seed,
                       Ap.get(), Ai.get(), Ax.get());
    OQ_LAUNCH(k_gen_U, dim3(blocks_for(((long long)n + 1) * 64)), dim3(kBlock), 0, s, (long long)n, kp, seed, Pp.get(),
                       Pi.get(), Px.get(), S.get());
    OQ_LAUNCH(k_gen_diag, dim3(blocks_for(n)), dim3(kBlock), 0, s, (long long)n, Pp.get(), Px.get(), S.get());
    OQ_LAUNCH(k_gen_vecs, dim3(blocks_for(std::max<long long>(n, m))), dim3(kBlock), 0, s, (long long)n, m, seed, q.get(),
                       l.get(), u.get());
    HIP_CHECK(hipStreamSynchronize(s));
    return;
  }
  if (kind == OSQP_AMD_GEN_LASSO) {
    long long m = 2LL * n;
    n_out = n; m_out = (int)m;
    Pp.alloc((size_t)n + 1); Pi.alloc(n); Px.alloc(n);
    Ap.alloc((size_t)n + 1); Ai.alloc(2 * (size_t)n); Ax.alloc(2 * (size_t)n);
    q.alloc(n); l.alloc(m); u.alloc(m);
    OQ_LAUNCH(k_gen_lasso, dim3(blocks_for((long long)n + 1)), dim3(kBlock), 0, s, (long long)n, seed, Pp.get(), Pi.get(),
                       Px.get(), Ap.get(), Ai.get(), Ax.get(), q.get(), l.get(), u.get());
    HIP_CHECK(hipStreamSynchronize(s));
    return;
  }
  throw Error(1, "unknown problem kind for the device generator");
}

}  // namespace oq
