// direct_factor_kernels.hpp -- assembly of K into the pattern of L and the level-by-level numeric LDL' (rows K2 of SURVEY.md section 8a)
// Part of the direct KKT back-end: included by direct.hip (one translation unit, one anonymous namespace); split out in round 6
// for reviewability -- direct.hip keeps the factor object (LdlFactor), the back-end (Direct) and the set-up decisions.
#pragma once
#include "engine.hpp"

namespace oq {
namespace {

// ------------------------------------------------------------------ assembly
__global__ __launch_bounds__(kBlock) void k_diag_init(int N, int n, double sigma, const int *__restrict__ pinv,
                                                      const double *__restrict__ cdiag, double cconst, double *__restrict__ D) {
  int o = blockIdx.x * kBlock + threadIdx.x;
  if (o >= N) return;
  D[pinv[o]] = o < n ? sigma : (cdiag ? -cdiag[o - n] : cconst);
}
__global__ __launch_bounds__(kBlock) void k_scatter_P(int64_t nnz, const int64_t *__restrict__ PtoL, const int *__restrict__ k2lo,
                                                      const double *__restrict__ Pfval, double *__restrict__ Lx, double *__restrict__ D) {
  int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k >= nnz) return;
  int64_t t = PtoL[k];
  double v = Pfval[k2lo[k]];
  if (t >= 0) Lx[t] = v; else D[-t - 1] += v;
}
__global__ __launch_bounds__(kBlock) void k_scatter_A(int64_t nnz, const int64_t *__restrict__ AtoL, const double *__restrict__ Atval,
                                                      double *__restrict__ Lx) {
  int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k >= nnz) return;
  int64_t t = AtoL[k];
  if (t != INT64_MIN) Lx[t] = Atval[k];
}

// ------------------------------------------------------------------ K2: numeric LDL'
// Dot-product form, two launches per level (levels ascending; every column of a level only needs columns of
// lower levels).  On entry Lx / D hold the entries of K (lower part), on exit L and the pivots.
//   phase 1, one wavefront per column k:   d_k = K_kk - sum_j L_kj^2 d_j              (row k of L, CSR view; k_ldl_diag_w)
//   phase 2, one thread per entry (i, k):  L_ik = (K_ik - sum_j L_ij L_kj d_j) / d_k   (merge of rows i and k,
//            both sorted by column; only j < k can match because row k ends at k)
// All entries of a column -- and all columns of a level -- are independent, so a dense trailing block
// exposes (N - k) lanes per column instead of one wavefront walking k updates one after the other.
__global__ __launch_bounds__(kBlock) void k_ldl_entries(int c0, int c1, const int64_t *__restrict__ Lp, const int *__restrict__ Li,
                                                        double *__restrict__ Lx, const int64_t *__restrict__ Rp,
                                                        const int *__restrict__ Rj, const int64_t *__restrict__ Rmap,
                                                        const double *__restrict__ D, const double *__restrict__ Dinv,
                                                        const int *__restrict__ Lcol) {
  const int64_t e = Lp[c0] + (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= Lp[c1]) return;
  const int k = Lcol[e], i = Li[e];  // the entry's column from a table built once (a bisection in Lp was ~8 dependent loads per entry)
  int64_t a = Rp[i], ae = Rp[i + 1], b = Rp[k], be = Rp[k + 1];
  double acc = 0.0;
  while (a < ae && b < be) {
    const int ja = Rj[a], jb = Rj[b];
    if (ja == jb) { acc += Lx[Rmap[a]] * Lx[Rmap[b]] * D[ja]; a++; b++; }
    else if (ja < jb) a++; else b++;
  }
  Lx[e] = (Lx[e] - acc) * Dinv[k];
}
// Phase 2 for levels whose rows are long (the dense trailing block behind a few dense constraint rows): the
// thread-per-entry merge walks 10^4 entries serially.  Instead row k is scattered once into a dense work row
// w_k[j] = L_kj d_j (k_ldl_wrow, one wavefront per column of the level; fill = 0 clears it again afterwards),
// and one wavefront per entry (i, k) takes the sparse-times-dense product of row i with w_k: coalesced reads of
// row i, gathers from a work row that stays in L2.  Only columns below the level contribute (two columns of one
// level are independent in the elimination tree), which also keeps the waves of a level off each other's output.
__global__ __launch_bounds__(kBlock) void k_ldl_wrow(int c0, int c1, int N, const double *__restrict__ Lx, const int64_t *__restrict__ Rp,
                                                     const int *__restrict__ Rj, const int64_t *__restrict__ Rmap,
                                                     const double *__restrict__ D, double *__restrict__ W, int fill) {
  const int lane = threadIdx.x & 63;
  const int k = c0 + (int)(((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6);
  if (k >= c1) return;
  double *w = W + (size_t)(k - c0) * N;
  for (int64_t q = Rp[k] + lane; q < Rp[k + 1]; q += 64) { const int j = Rj[q]; w[j] = fill ? Lx[Rmap[q]] * D[j] : 0.0; }
}
// phase 1 of a level and the work rows of its phase 2 in one launch: one wavefront per column k of [c0, c1) walks row k
// once for d_k and (Wfill != nullptr) for w_k = L_k,: o d; the wavefronts behind them clear the work rows of the previous
// such level [p0, p1) in the other half of W (its phase 2 has run: stream order), so a level is two launches, not four.
__global__ __launch_bounds__(kBlock) void k_ldl_diag_w(int c0, int c1, int N, const double *__restrict__ Lx, const int64_t *__restrict__ Rp,
                                                       const int *__restrict__ Rj, const int64_t *__restrict__ Rmap, double *__restrict__ D,
                                                       double *__restrict__ Dinv, int *__restrict__ status, double *__restrict__ Wfill, int p0,
                                                       int p1, double *__restrict__ Wclear) {
  const int lane = threadIdx.x & 63;
  const int64_t wv = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  if (wv < c1 - c0) {
    const int k = c0 + (int)wv;
    double *w = Wfill ? Wfill + (size_t)(k - c0) * N : nullptr;
    double acc = 0.0;
    for (int64_t q = Rp[k] + lane; q < Rp[k + 1]; q += 64) {
      const int j = Rj[q];
      const double l = Lx[Rmap[q]], d = D[j];
      acc += l * l * d;  // (l l) d, the order of the oracle's column update: pivots agree to the last bit on short rows
      if (w) w[j] = l * d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) {
      const double dk = D[k] - acc;
      const bool bad = (dk == 0.0) || (dk != dk);
      D[k] = dk; Dinv[k] = 1.0 / dk;
      if (bad) atomicOr(&status[0], 1);
      else if (dk > 0.0) atomicAdd(&status[1], 1);
    }
  } else if (wv < (int64_t)(c1 - c0) + (p1 - p0)) {
    const int k = p0 + (int)(wv - (c1 - c0));
    double *w = Wclear + (size_t)(k - p0) * N;
    for (int64_t q = Rp[k] + lane; q < Rp[k + 1]; q += 64) w[Rj[q]] = 0.0;
  }
}
// phase 1 for levels of SHORT rows (no work rows involved): G lanes per column instead of a wavefront -- a level of 10^6
// columns with two entries each (the constraint rows of a lasso / box-constrained problem) is 10^6 wavefronts of which 62
// lanes idle in k_ldl_diag_w: 2.9 ms per factorisation on lasso-5e5 where the whole ADMM iteration takes 31 us.  G = 1 adds
// the terms in column order, the order of the oracle's update.
template <int G>
__global__ __launch_bounds__(kBlock) void k_ldl_diag_g(int c0, int c1, const double *__restrict__ Lx, const int64_t *__restrict__ Rp,
                                                       const int *__restrict__ Rj, const int64_t *__restrict__ Rmap, double *__restrict__ D,
                                                       double *__restrict__ Dinv, int *__restrict__ status) {
  const int64_t g = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / G;
  const int lane = threadIdx.x & (G - 1);
  const bool live = g < c1 - c0;
  const int k = c0 + (int)(live ? g : 0);
  double acc = 0.0;
  if (live)
    for (int64_t q = Rp[k] + lane; q < Rp[k + 1]; q += G) {
      const double l = Lx[Rmap[q]];
      acc += l * l * D[Rj[q]];
    }
#pragma unroll
  for (int o = G >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (live && lane == 0) {
    const double dk = D[k] - acc;
    const bool bad = (dk == 0.0) || (dk != dk);
    D[k] = dk; Dinv[k] = 1.0 / dk;
    if (bad) atomicOr(&status[0], 1);
    else if (dk > 0.0) atomicAdd(&status[1], 1);
  }
}
// The same entries with a WAVEFRONT each and no work rows: the lanes take the entries of the shorter of the two rows and
// look each column up in the longer one by bisection -- ~log2(length) dependent loads per lane where the thread-per-entry
// merge walks both rows (2 x 300 dependent loads per entry on the separators of a nested-dissection tree).  For levels whose
// rows are long but whose work rows (one dense N-vector per column) would not fit: control-1e6 has hundreds of separator
// columns per level at N = 2.7e6 (0.47 ms per level with the merge).
template <int G>  // lanes per entry: 64 on narrow levels (all latency), 16 on wide ones
__global__ __launch_bounds__(kBlock) void k_ldl_entries_bs(int c0, int c1, const int64_t *__restrict__ Lp, const int *__restrict__ Li,
                                                           double *__restrict__ Lx, const int64_t *__restrict__ Rp,
                                                           const int *__restrict__ Rj, const int64_t *__restrict__ Rmap,
                                                           const double *__restrict__ D, const double *__restrict__ Dinv,
                                                           const int *__restrict__ Lcol) {
  const int lane = threadIdx.x & (G - 1);
  const int64_t e = Lp[c0] + (((int64_t)blockIdx.x * kBlock + threadIdx.x) / G);
  const bool live = e < Lp[c1];  // all lanes of a group agree; dead groups still take part in the shuffles
  double acc = 0.0;
  int k = c0;
  if (live) {
    k = Lcol[e];
    const int i = Li[e];
    int64_t a0 = Rp[i], a1 = Rp[i + 1], b0 = Rp[k], b1 = Rp[k + 1];
    if (a1 - a0 > b1 - b0) { int64_t t = a0; a0 = b0; b0 = t; t = a1; a1 = b1; b1 = t; }  // [a0, a1): the shorter row
    for (int64_t q = a0 + lane; q < a1; q += G) {
      const int j = Rj[q];
      if (j >= k) break;  // row i holds columns up to i > k; only those below k meet row k
      int64_t l = b0, h = b1;
      while (l < h) { const int64_t mid = (l + h) >> 1; if (Rj[mid] < j) l = mid + 1; else h = mid; }
      if (l < b1 && Rj[l] == j) acc += Lx[Rmap[q]] * Lx[Rmap[l]] * D[j];
    }
  }
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (live && lane == 0) Lx[e] = (Lx[e] - acc) * Dinv[k];
}
__global__ __launch_bounds__(kBlock) void k_ldl_entries_w(int c0, int c1, int N, const int64_t *__restrict__ Lp, const int *__restrict__ Li,
                                                          double *__restrict__ Lx, const int64_t *__restrict__ Rp,
                                                          const int *__restrict__ Rj, const int64_t *__restrict__ Rmap,
                                                          const double *__restrict__ W, const double *__restrict__ Dinv,
                                                          const int *__restrict__ Lcol) {
  const int lane = threadIdx.x & 63;
  const int64_t e = Lp[c0] + (((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6);
  if (e >= Lp[c1]) return;
  const int k = Lcol[e], i = Li[e];
  const double *w = W + (size_t)(k - c0) * N;
  double acc = 0.0;
  for (int64_t q = Rp[i] + lane; q < Rp[i + 1]; q += 64) {
    const int j = Rj[q];
    if (j >= c0) break;  // columns ascending: nothing below the level is left (for any lane at or after this one)
    acc += Lx[Rmap[q]] * w[j];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) Lx[e] = (Lx[e] - acc) * Dinv[k];
}

}  // namespace
}  // namespace oq
