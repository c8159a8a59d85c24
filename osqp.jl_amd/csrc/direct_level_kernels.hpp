// direct_level_kernels.hpp -- K3 / K4: level-scheduled triangular solves, LDS chains of narrow levels, the fused kernels of a direct ADMM iteration
// Part of the direct KKT back-end: included by direct.hip (one translation unit, one anonymous namespace); split out in round 6
// for reviewability -- direct.hip keeps the factor object (LdlFactor), the back-end (Direct) and the set-up decisions.
#pragma once
#include "engine.hpp"

namespace oq {
namespace {

// ------------------------------------------------------------------ K3 / K4: level-scheduled triangular solves
template <int G>
__global__ __launch_bounds__(kBlock) void k_fwd_level(int r0, int r1, const int64_t *__restrict__ Rp, const int *__restrict__ Rj,
                                                      const double *__restrict__ Rx, double *__restrict__ b) {
  const int lane = threadIdx.x & (G - 1);
  const int row = r0 + (int)(((int64_t)blockIdx.x * kBlock + threadIdx.x) / G);
  if (row >= r1) return;
  double acc = gather_dot(Rp[row] + lane, Rp[row + 1], G, Rj, Rx, b);
#pragma unroll
  for (int o = G >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) b[row] -= acc;
}
template <int G>
__global__ __launch_bounds__(kBlock) void k_bwd_level(int r0, int r1, const int64_t *__restrict__ Lp, const int *__restrict__ Li,
                                                      const double *__restrict__ Lx, const double *__restrict__ Dinv,
                                                      double *__restrict__ b) {
  const int lane = threadIdx.x & (G - 1);
  const int row = r0 + (int)(((int64_t)blockIdx.x * kBlock + threadIdx.x) / G);
  if (row >= r1) return;
  double acc = gather_dot(Lp[row] + lane, Lp[row + 1], G, Li, Lx, b);
#pragma unroll
  for (int o = G >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) b[row] = b[row] * Dinv[row] - acc;
}
// Chains of narrow levels inside one workgroup (barrier between levels); 4 lanes per row, or a whole
// wavefront per row when the level has at most 16 rows.  A row of a chain splits at Rsplit[row] into the
// entries whose columns lie before the chain (all of them solved when the chain starts: k_fwd_far takes
// them for every row of the chain at once, T threads per row) and the entries inside the chain (the only part
// that is sequential).  With a dense trailing block -- a few dense constraint rows -- the first part is the
// long one: 10^4 entries per row against 10^2 inside the chain.
template <int T>
__global__ __launch_bounds__(kBlock) void k_fwd_far(int r0, int r1, const int64_t *__restrict__ Rp, const int64_t *__restrict__ Rsplit,
                                                    const int *__restrict__ Rj, const double *__restrict__ Rx, double *__restrict__ b,
                                                    double *__restrict__ out) {  // out != nullptr: out[row - r0] instead of b[row]
  __shared__ double part[kBlock / 64];
  const int lane = threadIdx.x & (T - 1);
  const int row = r0 + (int)(((int64_t)blockIdx.x * kBlock + threadIdx.x) / T);
  double acc = row < r1 ? gather_dot(Rp[row] + lane, Rsplit[row], T, Rj, Rx, b) : 0.0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (T == 64) {
    if (lane == 0 && row < r1) { if (out) out[row - r0] = b[row] - acc; else b[row] -= acc; }
  } else {  // T == kBlock: one row per workgroup
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && row < r1) {
      double t = 0.0;
      for (int w = 0; w < kBlock / 64; w++) t += part[w];
      if (out) out[row - r0] = b[row] - t; else b[row] -= t;
    }
  }
}
// Backward counterpart: the entries of column k of L with rows above the chain [c0, c1) (solved earlier in the
// backward pass) are taken for all pivots of the chain at once, together with the D^-1 scaling:
// b[k] = b[k] / d_k - sum_{i >= c1} L_ik b[i]; the chain then only walks the entries inside it.
template <int T>
__global__ __launch_bounds__(kBlock) void k_bwd_far(int r0, int r1, const int64_t *__restrict__ Lsplit, const int64_t *__restrict__ Lp,
                                                    const int *__restrict__ Li, const double *__restrict__ Lx,
                                                    const double *__restrict__ Dinv, double *__restrict__ b) {
  __shared__ double part[kBlock / 64];
  const int lane = threadIdx.x & (T - 1);
  const int row = r0 + (int)(((int64_t)blockIdx.x * kBlock + threadIdx.x) / T);
  double acc = row < r1 ? gather_dot(Lsplit[row] + lane, Lp[row + 1], T, Li, Lx, b) : 0.0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (T == 64) {
    if (lane == 0 && row < r1) b[row] = b[row] * Dinv[row] - acc;
  } else {
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && row < r1) {
      double t = 0.0;
      for (int w = 0; w < kBlock / 64; w++) t += part[w];
      b[row] = b[row] * Dinv[row] - t;
    }
  }
}
// LDS-resident chains.  A chain is cut so that its pivots [c0, c1) and its level table fit in LDS (build_schedule):
// the segment of the solution lives there for the whole chain and the workgroup never touches global memory on the
// critical path: row r belongs to wavefront (r - c0) mod 16 for good, so a wavefront knows its next row ahead of time and
// fetches its bounds and first 128 entries right after finishing the current one, levels before they are
// needed; the barrier between levels only waits for LDS traffic (s_waitcnt lgkmcnt(0); s_barrier -- the plain
// __syncthreads would also drain those prefetches).  Per level that leaves an LDS gather, a wavefront reduction
// and the barrier: ~0.15 us instead of ~2 us of dependent global round trips.
constexpr int kChainLdsRows = 8192, kChainLdsLevels = 8192;
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// Two rows ahead: the bounds of the row after next (so that fetching the entries of the next row never waits for
// its own bounds), one row ahead: bounds and first G U entries of the next row (U = 2, 4 or 8 by the
// longest row inside the chain: a dense trailing block has rows as long as the block).
template <int U>
struct RowPrefetch {
  int64_t q0, q1;    // entries of the next row inside the chain
  int64_t nq0, nq1;  // the same for the row after it
  double v[U];
  int c[U];
};
// G lanes share a row (64: a wavefront per row -- long rows of a dense block; 16 or 4: several short rows per wavefront)
template <int U, int G>
__device__ __forceinline__ void prefetch_entries(RowPrefetch<U> &p, const int *__restrict__ idx, const double *__restrict__ val, int lane) {
  p.q0 = p.nq0; p.q1 = p.nq1;
#pragma unroll
  for (int u = 0; u < U; u++) {
    const int64_t a = p.q0 + G * u + lane;
    p.v[u] = a < p.q1 ? val[a] : 0.0;
    p.c[u] = a < p.q1 ? idx[a] : -1;
  }
}
template <int U, int G>
__global__ __launch_bounds__(kChainThreads) void k_fwd_chain_lds(int l0, int l1, const int *__restrict__ level_ptr,
                                                                 const int64_t *__restrict__ Rsplit, const int64_t *__restrict__ Rp,
                                                                 const int *__restrict__ Rj, const double *__restrict__ Rx,
                                                                 double *__restrict__ b) {
  __shared__ double bl[kChainLdsRows];
  __shared__ int lp[kChainLdsLevels + 1];
  constexpr int kStride = kChainThreads / G;  // rows in flight: one per group of G lanes
  const int c0 = level_ptr[l0], c1 = level_ptr[l1];
  for (int i = threadIdx.x; i < c1 - c0; i += kChainThreads) bl[i] = b[c0 + i];
  for (int i = threadIdx.x; i <= l1 - l0; i += kChainThreads) lp[i] = level_ptr[l0 + i];
  const int grp = threadIdx.x / G, lane = threadIdx.x & (G - 1);
  int next = c0 + grp;
  RowPrefetch<U> pf;
  pf.nq0 = pf.nq1 = 0;
#pragma unroll
  for (int u = 0; u < U; u++) { pf.v[u] = 0.0; pf.c[u] = -1; }
  pf.q0 = pf.q1 = 0;
  if (next < c1) { pf.nq0 = Rsplit[next]; pf.nq1 = Rp[next + 1]; prefetch_entries<U, G>(pf, Rj, Rx, lane); }
  if (next + kStride < c1) { pf.nq0 = Rsplit[next + kStride]; pf.nq1 = Rp[next + kStride + 1]; }
  __syncthreads();
  for (int l = 0; l < l1 - l0; l++) {
    const int r1 = lp[l + 1];
    while (__any(next < r1)) {  // the groups of a wavefront may differ by one row: idle ones ride along
      const bool mine = next < r1;
      double acc = 0.0;
      if (mine) {
#pragma unroll
        for (int u = 0; u < U; u++) if (pf.c[u] >= 0) acc += pf.v[u] * bl[pf.c[u] - c0];
        for (int64_t q = pf.q0 + G * U + lane; q < pf.q1; q += G) acc += Rx[q] * bl[Rj[q] - c0];
      }
#pragma unroll
      for (int o = G >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
      if (mine) {
        if (lane == 0) bl[next - c0] -= acc;
        next += kStride;
        if (next < c1) prefetch_entries<U, G>(pf, Rj, Rx, lane);
        if (next + kStride < c1) { pf.nq0 = Rsplit[next + kStride]; pf.nq1 = Rp[next + kStride + 1]; }
      }
    }
    lds_barrier();
  }
  for (int i = threadIdx.x; i < c1 - c0; i += kChainThreads) b[c0 + i] = bl[i];
}
// backward: the column of L below pivot k is row k of L'; only its rows inside the chain are left (k_bwd_far took
// the rest and the D^-1 scaling).  Pivots and levels descend.
template <int U, int G>
__global__ __launch_bounds__(kChainThreads) void k_bwd_chain_lds(int l0, int l1, const int *__restrict__ level_ptr,
                                                                 const int64_t *__restrict__ Lp, const int64_t *__restrict__ Lsplit,
                                                                 const int *__restrict__ Li, const double *__restrict__ Lx,
                                                                 double *__restrict__ b) {
  __shared__ double bl[kChainLdsRows];
  __shared__ int lp[kChainLdsLevels + 1];
  constexpr int kStride = kChainThreads / G;
  const int c0 = level_ptr[l0], c1 = level_ptr[l1];
  for (int i = threadIdx.x; i < c1 - c0; i += kChainThreads) bl[i] = b[c0 + i];
  for (int i = threadIdx.x; i <= l1 - l0; i += kChainThreads) lp[i] = level_ptr[l0 + i];
  const int grp = threadIdx.x / G, lane = threadIdx.x & (G - 1);
  int next = c1 - 1 - grp;
  RowPrefetch<U> pf;
  pf.nq0 = pf.nq1 = 0;
#pragma unroll
  for (int u = 0; u < U; u++) { pf.v[u] = 0.0; pf.c[u] = -1; }
  pf.q0 = pf.q1 = 0;
  if (next >= c0) { pf.nq0 = Lp[next]; pf.nq1 = Lsplit[next]; prefetch_entries<U, G>(pf, Li, Lx, lane); }
  if (next - kStride >= c0) { pf.nq0 = Lp[next - kStride]; pf.nq1 = Lsplit[next - kStride]; }
  __syncthreads();
  for (int l = l1 - l0 - 1; l >= 0; l--) {
    const int r0 = lp[l];
    while (__any(next >= r0)) {
      const bool mine = next >= r0;
      double acc = 0.0;
      if (mine) {
#pragma unroll
        for (int u = 0; u < U; u++) if (pf.c[u] >= 0) acc += pf.v[u] * bl[pf.c[u] - c0];
        for (int64_t t = pf.q0 + G * U + lane; t < pf.q1; t += G) acc += Lx[t] * bl[Li[t] - c0];
      }
#pragma unroll
      for (int o = G >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
      if (mine) {
        if (lane == 0) bl[next - c0] -= acc;
        next -= kStride;
        if (next >= c0) prefetch_entries<U, G>(pf, Li, Lx, lane);
        if (next - kStride >= c0) { pf.nq0 = Lp[next - kStride]; pf.nq1 = Lsplit[next - kStride]; }
      }
    }
    lds_barrier();
  }
  for (int i = threadIdx.x; i < c1 - c0; i += kChainThreads) b[c0 + i] = bl[i];
}
__global__ __launch_bounds__(kBlock) void k_perm_in(int N, const int *__restrict__ perm, const double *__restrict__ in, double *__restrict__ bp) {
  int k = blockIdx.x * kBlock + threadIdx.x;
  if (k < N) bp[k] = in[perm[k]];
}
// ADMM form: x~ = sol_x ; z~ = rhs_z + rho^-1 nu   (SURVEY.md A.2).  plain form: out = sol
__global__ __launch_bounds__(kBlock) void k_perm_out(int N, int n, const int *__restrict__ pinv, const double *__restrict__ bp,
                                                     const double *__restrict__ rho_inv, double *__restrict__ out) {
  int o = blockIdx.x * kBlock + threadIdx.x;
  if (o >= N) return;
  double v = bp[pinv[o]];
  if (rho_inv && o >= n) out[o] += rho_inv[o - n] * v; else out[o] = v;
}

// fused iteration ends (direct back-end): the right-hand side is written straight into the pivot order and the
// ADMM update reads the solution through the inverse permutation, so an iteration is rhs | trisolves | update.
__global__ __launch_bounds__(kBlock) void k_direct_rhs(int n, int m, double sigma, const int *__restrict__ pinv,
                                                       const double *__restrict__ x, const double *__restrict__ q,
                                                       const double *__restrict__ z, const double *__restrict__ rho_inv,
                                                       const double *__restrict__ y, double *__restrict__ bp) {
  int o = blockIdx.x * kBlock + threadIdx.x;
  if (o < n) bp[pinv[o]] = sigma * x[o] - q[o];
  else if (o < n + m) { int j = o - n; bp[pinv[o]] = z[j] - rho_inv[j] * y[j]; }
}
__global__ __launch_bounds__(kBlock) void k_direct_update(int n, int m, double alpha, const int *__restrict__ pinv,
                                                          const double *__restrict__ bp, const double *__restrict__ rho,
                                                          const double *__restrict__ rho_inv, const double *__restrict__ l,
                                                          const double *__restrict__ u, double *__restrict__ x, double *__restrict__ z,
                                                          double *__restrict__ y, double *__restrict__ delta_x,
                                                          double *__restrict__ delta_y) {
  int o = blockIdx.x * kBlock + threadIdx.x;
  if (o < n) {
    double xp = x[o];
    double xn = alpha * bp[pinv[o]] + (1.0 - alpha) * xp;
    x[o] = xn;
    delta_x[o] = xn - xp;
  } else if (o < n + m) {
    int j = o - n;
    double zp = z[j], yj = y[j], ri = rho_inv[j];
    double zt = (zp - ri * yj) + ri * bp[pinv[o]];  // z~ = rhs_z + rho^-1 nu  
    double zh = alpha * zt + (1.0 - alpha) * zp;
    double zn = fmin(fmax(zh + ri * yj, l[j]), u[j]);
    z[j] = zn;
    double dy = rho[j] * (zh - zn);
    delta_y[j] = dy;
    y[j] = yj + dy;
  }
}

// The same two ends with the neighbouring level folded in (one thread per index; used when the rows of level 1 / the
// columns of level 0 are short -- bound constraints, diagonal blocks): the right-hand side of a level-0 pivot is cheap
// to recompute, so a level-1 row takes what it needs from the original vectors instead of waiting for a kernel that
// writes them; and the update computes the solution of a level-0 pivot on the spot instead of reading it back.
__device__ __forceinline__ double direct_rhs_value(int o, int n, double sigma, const double *__restrict__ x, const double *__restrict__ q,
                                                   const double *__restrict__ z, const double *__restrict__ rho_inv,
                                                   const double *__restrict__ y) {
  if (o < n) return sigma * x[o] - q[o];
  const int j = o - n;
  return z[j] - rho_inv[j] * y[j];
}
__global__ __launch_bounds__(kBlock) void k_direct_rhs_fwd1(int n, int m, double sigma, const int *__restrict__ pinv, const int *__restrict__ perm,
                                                            int l1_begin, int l1_end, const int64_t *__restrict__ Rp, const int *__restrict__ Rj,
                                                            const double *__restrict__ Rx, const double *__restrict__ x,
                                                            const double *__restrict__ q, const double *__restrict__ z,
                                                            const double *__restrict__ rho_inv, const double *__restrict__ y,
                                                            double *__restrict__ bp) {
  int o = blockIdx.x * kBlock + threadIdx.x;
  if (o >= n + m) return;
  const int k = pinv[o];
  double v = direct_rhs_value(o, n, sigma, x, q, z, rho_inv, y);
  if (k >= l1_begin && k < l1_end) {
    double acc = 0.0;
    for (int64_t t = Rp[k]; t < Rp[k + 1]; t++) acc += Rx[t] * direct_rhs_value(perm[Rj[t]], n, sigma, x, q, z, rho_inv, y);
    v -= acc;
  }
  bp[k] = v;
}
__global__ __launch_bounds__(kBlock) void k_direct_bwd0_update(int n, int m, double alpha, const int *__restrict__ pinv, int l0_end,
                                                               const int64_t *__restrict__ Lp, const int *__restrict__ Li,
                                                               const double *__restrict__ Lx, const double *__restrict__ Dinv,
                                                               const double *__restrict__ bp, const double *__restrict__ rho,
                                                               const double *__restrict__ rho_inv, const double *__restrict__ l,
                                                               const double *__restrict__ u, double *__restrict__ x, double *__restrict__ z,
                                                               double *__restrict__ y, double *__restrict__ delta_x,
                                                               double *__restrict__ delta_y) {
  int o = blockIdx.x * kBlock + threadIdx.x;
  if (o >= n + m) return;
  const int k = pinv[o];
  double sol = bp[k];
  if (k < l0_end) {  // level 0: the last backward step, done here
    double acc = 0.0;
    for (int64_t t = Lp[k]; t < Lp[k + 1]; t++) acc += Lx[t] * bp[Li[t]];
    sol = sol * Dinv[k] - acc;
  }
  if (o < n) {
    double xp = x[o];
    double xn = alpha * sol + (1.0 - alpha) * xp;
    x[o] = xn;
    delta_x[o] = xn - xp;
  } else {
    int j = o - n;
    double zp = z[j], yj = y[j], ri = rho_inv[j];
    double zt = (zp - ri * yj) + ri * sol;
    double zh = alpha * zt + (1.0 - alpha) * zp;
    double zn = fmin(fmax(zh + ri * yj, l[j]), u[j]);
    z[j] = zn;
    double dy = rho[j] * (zh - zn);
    delta_y[j] = dy;
    y[j] = yj + dy;
  }
}

// Two-level factors (every constraint row a leaf under the variable it bounds: lasso, box-constrained problems; KKT
// systems whose fill-free elimination has height 1) need no level kernel at all -- the whole iteration is two launches:
//   k_direct2_fwd         thread per level-1 pivot k: the right-hand sides of k and of its level-0 columns are
//                         recomputed from (x, q, z, rho^-1, y) -- a level-0 right-hand side is never stored -- the
//                         forward step and, level 1 being the top of the tree, the D^-1 scaling: bp[k] is final;
//   k_direct2_bwd_update  thread per KKT index o: a level-0 pivot takes its right-hand side from the same vectors the
//                         ADMM update reads anyway, does its backward step against the level-1 solutions and goes
//                         straight into the update of x / z / y.
// Same operations in the same order as k_direct_rhs_fwd1 | k_bwd_level | k_direct_bwd0_update: bit-identical iterates.
__global__ __launch_bounds__(kBlock) void k_direct2_fwd(int n, int N, double sigma, const int *__restrict__ perm, int l1_begin,
                                                        const int64_t *__restrict__ Rp, const int *__restrict__ Rj,
                                                        const double *__restrict__ Rx, const double *__restrict__ Dinv,
                                                        const double *__restrict__ x, const double *__restrict__ q,
                                                        const double *__restrict__ z, const double *__restrict__ rho_inv,
                                                        const double *__restrict__ y, double *__restrict__ bp) {
  const int k = l1_begin + blockIdx.x * kBlock + threadIdx.x;
  if (k >= N) return;
  double v = direct_rhs_value(perm[k], n, sigma, x, q, z, rho_inv, y);
  double acc = 0.0;
  for (int64_t t = Rp[k]; t < Rp[k + 1]; t++) acc += Rx[t] * direct_rhs_value(perm[Rj[t]], n, sigma, x, q, z, rho_inv, y);
  v -= acc;
  bp[k] = v * Dinv[k] - 0.0;
}
__global__ __launch_bounds__(kBlock) void k_direct2_bwd_update(int n, int m, double sigma, double alpha, const int *__restrict__ pinv,
                                                               int l1_begin, const int64_t *__restrict__ Lp, const int *__restrict__ Li,
                                                               const double *__restrict__ Lx, const double *__restrict__ Dinv,
                                                               const double *__restrict__ bp, const double *__restrict__ q,
                                                               const double *__restrict__ rho, const double *__restrict__ rho_inv,
                                                               const double *__restrict__ l, const double *__restrict__ u,
                                                               double *__restrict__ x, double *__restrict__ z, double *__restrict__ y,
                                                               double *__restrict__ delta_x, double *__restrict__ delta_y) {
  const int o = blockIdx.x * kBlock + threadIdx.x;
  if (o >= n + m) return;
  const int k = pinv[o];
  const bool leaf = k < l1_begin;
  double acc = 0.0, dk = 0.0;
  if (leaf) {
    for (int64_t t = Lp[k]; t < Lp[k + 1]; t++) acc += Lx[t] * bp[Li[t]];
    dk = Dinv[k];
  }
  if (o < n) {
    const double xp = x[o];
    const double sol = leaf ? (sigma * xp - q[o]) * dk - acc : bp[k];
    const double xn = alpha * sol + (1.0 - alpha) * xp;
    x[o] = xn;
    delta_x[o] = xn - xp;
  } else {
    const int j = o - n;
    const double zp = z[j], yj = y[j], ri = rho_inv[j];
    const double rhs = zp - ri * yj;
    const double sol = leaf ? rhs * dk - acc : bp[k];
    const double zt = rhs + ri * sol;  // z~ = rhs_z + rho^-1 nu  
    const double zh = alpha * zt + (1.0 - alpha) * zp;
    const double zn = fmin(fmax(zh + ri * yj, l[j]), u[j]);
    z[j] = zn;
    const double dy = rho[j] * (zh - zn);
    delta_y[j] = dy;
    y[j] = yj + dy;
  }
}

}  // namespace
}  // namespace oq
