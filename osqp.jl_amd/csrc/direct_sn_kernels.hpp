// direct_sn_kernels.hpp -- supernodal triangular solves: per-level launches, wavefront / quarter-wavefront forms, the one-launch tree
// Part of the direct KKT back-end: included by direct.hip (one translation unit, one anonymous namespace); split out in round 6
// for reviewability -- direct.hip keeps the factor object (LdlFactor), the back-end (Direct) and the set-up decisions.
#pragma once
#include "engine.hpp"

namespace oq {
namespace {

// ------------------------------------------------------------------ supernodal triangular solves
// (symbolic.hpp, Supernodes) One workgroup per supernode, one launch per level of the supernode graph.  With
// W = L_JJ^-1 (unit lower triangular, dense s x s, stored twice: Wc[j*s+a] = W(a,j), Wr[j*s+a] = W(j,a)):
//   forward   t = b_J - F_J y (entries of the rows of J outside its block),  y_J = W t
//   backward  u = D_J^-1 y_J - G_J x (entries of the columns of J outside its block),  x_J = W' u
// A deep elimination tree (nested dissection of a long banded problem: 300 pivot levels) is 15 such levels.
constexpr int kSnMax = 64, kSnThreads = 256;
constexpr int kSnBusyLevel = 2048;   // supernodes in a level from which its workgroups no longer fit the device at once
constexpr int kSnWaveLevel = 16384;  // supernodes in a level from which each gets a wavefront instead of a workgroup (below: the device is not full either way and a workgroup finishes its supernode sooner)

__global__ __launch_bounds__(kSnThreads) void k_sn_invert(const int *__restrict__ ptr, const int64_t *__restrict__ woff,
                                                          const int64_t *__restrict__ wmap, const double *__restrict__ Lx,
                                                          double *__restrict__ Wc, double *__restrict__ Wr) {
  __shared__ double Ld[kSnMax * kSnMax], Wd[kSnMax * kSnMax];
  const int J = blockIdx.x, s = ptr[J + 1] - ptr[J];
  const int64_t w0 = woff[J];
  // blocks are stored as packed lower triangles (round 4): wmap row-major (a (a + 1) / 2 + b, b <= a), the inverse twice --
  // Wc column by column (what a forward row product walks with its lanes along the rows), Wr row by row (backward)
  for (int e = threadIdx.x; e < s * s; e += kSnThreads) {
    const int i = e / s, k = e - i * s;
    double v = 0.0;
    if (k < i) { const int64_t t = wmap[w0 + (int64_t)i * (i + 1) / 2 + k]; if (t >= 0) v = Lx[t]; }
    Ld[e] = v;
  }
  __syncthreads();
  if ((int)threadIdx.x < s) {  // column j of the inverse by forward substitution: W(i,j) = -sum_{k=j}^{i-1} L(i,k) W(k,j)
    const int j = threadIdx.x;
    for (int i = 0; i < s; i++) {
      double w = 0.0;
      if (i == j) w = 1.0;
      else if (i > j) {
        for (int k = j; k < i; k++) w -= Ld[i * s + k] * Wd[k * s + j];
      }
      Wd[i * s + j] = w;
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < s * s; e += kSnThreads) {
    const int j = e / s, a = e - j * s;
    if (a >= j) Wc[w0 + (int64_t)j * s - (int64_t)j * (j - 1) / 2 + (a - j)] = Wd[a * s + j];  // W(a, j), column j from its diagonal down
    if (a <= j) Wr[w0 + (int64_t)j * (j + 1) / 2 + a] = Wd[j * s + a];                           // W(j, a), row j up to its diagonal
  }
}

__global__ __launch_bounds__(kBlock) void k_sn_gather(int64_t nf, const int64_t *__restrict__ Fpos, double *__restrict__ Fx, int64_t ng,
                                                      const int64_t *__restrict__ Gpos, double *__restrict__ Gx, int N,
                                                      const int *__restrict__ piv, const double *__restrict__ Dinv,
                                                      double *__restrict__ Dinv_s, const double *__restrict__ Lx) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < nf) Fx[i] = Lx[Fpos[i]];
  if (i < ng) Gx[i] = Lx[Gpos[i]];
  if (i < N) Dinv_s[i] = Dinv[piv[i]];
}

// sum over j = j0, j0 + dj, ... < s of W(a, j) t[j] (forward: the packed columns, j <= a) or W(j, a) t[j] (backward: the packed
// rows, j >= a) for row a of an s x s block; j is uniform over the lanes that call it together, four loads in flight
template <bool kForward>
__device__ __forceinline__ double sn_block_row(const double *__restrict__ Wj, const double *t, int s, int a, int j0, int dj) {
  auto w = [&](int j) -> double {
    if (kForward) return (a < s && j <= a) ? Wj[j * s - j * (j - 1) / 2 + (a - j)] : 0.0;
    return (a < s && j >= a) ? Wj[j * (j + 1) / 2 + a] : 0.0;
  };
  double acc = 0.0;
  int j = j0;
  for (; j + 3 * dj < s; j += 4 * dj) {
    const double w0 = w(j), w1 = w(j + dj), w2 = w(j + 2 * dj), w3 = w(j + 3 * dj);
    acc += w0 * t[j]; acc += w1 * t[j + dj]; acc += w2 * t[j + 2 * dj]; acc += w3 * t[j + 3 * dj];
  }
  for (; j < s; j += dj) acc += w(j) * t[j];
  return acc;
}

// LA lanes per row for the entries outside the block, 4 lanes per row for the block product
template <int LA, bool kForward>
__global__ __launch_bounds__(kSnThreads) void k_sn_level(int J0, const int *__restrict__ ptr, const int64_t *__restrict__ woff,
                                                         const int64_t *__restrict__ Ep, const int *__restrict__ Ej,
                                                         const double *__restrict__ Ex, const double *__restrict__ W,
                                                         const double *__restrict__ Dinv_s, double *__restrict__ b) {
  __shared__ double t[kSnMax];
  __shared__ double part[kSnThreads / 64][kSnMax];
  const int J = J0 + blockIdx.x, q0 = ptr[J], s = ptr[J + 1] - q0;
  {
    const int lane = threadIdx.x % LA;
    for (int a = threadIdx.x / LA; a < s; a += kSnThreads / LA) {
      const int q = q0 + a;
      double acc = gather_dot(Ep[q] + lane, Ep[q + 1], LA, Ej, Ex, b);
#pragma unroll
      for (int o = LA / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
      if (lane == 0) t[a] = (kForward ? b[q] : b[q] * Dinv_s[q]) - acc;
    }
  }
  __syncthreads();
  // The block product, round 5: lane = row a, the step index j is wavefront-uniform, so every load of the block is ONE
  // contiguous piece (column j of the packed columns forward, row j of the packed rows backward) instead of 64 scattered
  // doubles (four lanes per row, each walking its own row); the four wavefronts take every fourth j, their partial sums meet
  // in LDS in a fixed order.
  const double *Wj = W + woff[J];
  const int wv = threadIdx.x >> 6, a = threadIdx.x & 63;
  part[wv][a] = sn_block_row<kForward>(Wj, t, s, a, wv, kSnThreads / 64);
  __syncthreads();
  if (wv == 0 && a < s) b[q0 + a] = (part[0][a] + part[1][a]) + (part[2][a] + part[3][a]);
}

// The same step with a WAVEFRONT per supernode (four per workgroup), for levels of many small supernodes: level 0 of a
// nested-dissection tree is the leaves -- 540 000 subtrees of five pivots on average for control-1e6 -- and a 256-thread
// workgroup each leaves 250 of them idle (533 us for the forward level 0 of that problem).  Same arithmetic as k_sn_level.  The wavefront's t vector sits in its own slab of LDS; its writes are
// drained (s_waitcnt) before its reads, no workgroup barrier.  (The caller gives wide levels a notch fewer lanes per row
// than narrow ones -- rows in flight x latency is what bounds them -- so the order of a row's sum may differ between forms.)
template <int LA, bool kForward, int GS>  // GS: lanes per supernode, 64 or 16 (supernodes of at most 16 pivots: four to a wavefront)
__global__ __launch_bounds__(kSnThreads) void k_sn_level_w(int J0, int J1, const int *__restrict__ ptr, const int64_t *__restrict__ woff,
                                                           const int64_t *__restrict__ Ep, const int *__restrict__ Ej,
                                                           const double *__restrict__ Ex, const double *__restrict__ W,
                                                           const double *__restrict__ Dinv_s, double *__restrict__ b) {
  static_assert(LA <= GS, "the lanes of a row lie inside its supernode's group");
  constexpr int NG = 64 / GS, TS = GS == 64 ? kSnMax : GS;
  __shared__ double tt[kSnThreads / 64][NG][TS];
  const int wv = threadIdx.x >> 6, l64 = threadIdx.x & 63, g = l64 / GS, gl = l64 % GS;
  const int J = J0 + (blockIdx.x * (kSnThreads / 64) + wv) * NG + g;
  const bool live = J < J1;  // dead groups run no loop; the lanes of a row (and of its shuffles) share a group: all in or all out
  double *t = tt[wv][g];
  const int q0 = live ? ptr[J] : 0, s = live ? ptr[J + 1] - q0 : 0;
  {
    const int lane = gl % LA;
    for (int a = gl / LA; a < s; a += GS / LA) {
      const int q = q0 + a;
      double acc = gather_dot(Ep[q] + lane, Ep[q + 1], LA, Ej, Ex, b);
#pragma unroll
      for (int o = LA / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
      if (lane == 0) t[a] = (kForward ? b[q] : b[q] * Dinv_s[q]) - acc;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wavefront's LDS writes are done before any of its lanes reads them
  __builtin_amdgcn_wave_barrier();
  const double *Wj = W + (live ? woff[J] : 0);
  // lane = row, the step index uniform over the lanes of the supernode: contiguous loads of the block (see k_sn_level)
  const double acc = sn_block_row<kForward>(Wj, t, s, gl, 0, 1);
  if (gl < s) b[q0 + gl] = acc;
}

// The same two steps with the entries outside the block taken FLAT (round 5).  The forms above give every row LA lanes and
// walk the rows of a supernode LA-th by LA-th: a level-3 supernode of control-1e6 (54 rows of 94 entries) is fourteen rounds of
// {row pointers -> entries -> gathered solution -> shuffle sum} behind each other, 42 memory latencies for 60 KB -- the levels
// are bound by that chain, not by bytes (3 TB/s).  The entries of the rows of a supernode are ONE contiguous stretch of the
// lists (its rows are consecutive slots), so: every thread takes kSnFlatU entries of the stretch whatever their row -- full
// wavefront loads of the index and value streams, kSnFlatU gathers in flight per lane -- and leaves the products in LDS; then
// the rows add up their own piece of the chunk in entry order (four lanes per row taking every fourth entry, met in a fixed
// order: the same sum on every run).  Two latencies per chunk of 2048 entries.
constexpr int kSnFlatU = 8;
template <bool kForward, int DK>
__device__ __forceinline__ void sn_block_fold(const double *__restrict__ Wj, const double *t, int s, int gl, int k0, double &lo, double &hi);  // below
template <bool kForward, bool kFold>
__global__ __launch_bounds__(kSnThreads) void k_sn_level_f(int J0, const int *__restrict__ ptr, const int64_t *__restrict__ woff,
                                                           const int64_t *__restrict__ Ep, const int *__restrict__ Ej,
                                                           const double *__restrict__ Ex, const double *__restrict__ W,
                                                           const double *__restrict__ Dinv_s, double *__restrict__ b) {
  constexpr int C = kSnThreads * kSnFlatU;
  __shared__ double prod[C];
  __shared__ double t[kSnMax];
  __shared__ double part[kSnThreads / 64][kSnMax];
  const int J = J0 + blockIdx.x, q0 = ptr[J], s = ptr[J + 1] - q0;
  const int tid = threadIdx.x, ra = tid >> 2, rk = tid & 3;  // row ra (kSnThreads / 4 = kSnMax of them), lane rk of its four
  const int64_t E0 = Ep[q0], E1 = Ep[q0 + s];
  int64_t r1 = 0, pos = 0;
  double rhs = 0.0;
  if (ra < s) {
    pos = Ep[q0 + ra] + rk; r1 = Ep[q0 + ra + 1];
    if (rk == 0) rhs = kForward ? b[q0 + ra] : b[q0 + ra] * Dinv_s[q0 + ra];
  }
  double acc = 0.0;
  for (int64_t base = E0; base < E1; base += C) {
    int idx[kSnFlatU];
    double val[kSnFlatU], bv[kSnFlatU];
#pragma unroll
    for (int u = 0; u < kSnFlatU; u++) {
      const int64_t e = base + u * kSnThreads + tid;
      const bool ok = e < E1;
      idx[u] = ok ? Ej[e] : -1;
      val[u] = ok ? Ex[e] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < kSnFlatU; u++) bv[u] = idx[u] >= 0 ? b[idx[u]] : 0.0;
#pragma unroll
    for (int u = 0; u < kSnFlatU; u++) prod[u * kSnThreads + tid] = val[u] * bv[u];
    __syncthreads();
    const int64_t end = r1 < base + C ? r1 : base + C;
    for (; pos < end; pos += 4) acc += prod[pos - base];
    __syncthreads();
  }
  {
    const double a1 = __shfl_xor(acc, 1);
    const double pair = (rk & 1) ? a1 + acc : acc + a1;  // (lane 0 + lane 1), (lane 2 + lane 3) -- the same operands in the same order on both lanes
    const double other = __shfl_xor(pair, 2);
    const double sum = (rk & 2) ? other + pair : pair + other;
    if (rk == 0 && ra < s) t[ra] = rhs - sum;
  }
  __syncthreads();
  const double *Wj = W + woff[J];
  const int wv = tid >> 6, a = tid & 63;
  if (kFold) {  // the block is stored folded (sn_block_fold): the eight half-wavefronts take every eighth step
    double lo, hi;
    sn_block_fold<kForward, 8>(Wj, t, s, a, 2 * wv + (a >> 5), lo, hi);
    if (a < 32) { part[wv][a] = lo; part[wv][32 + a] = hi; }
    __syncthreads();
    const int h = (s + 1) >> 1;
    if (wv == 0 && a < h) {
      const double slo = (part[0][a] + part[1][a]) + (part[2][a] + part[3][a]);
      const double shi = (part[0][32 + a] + part[1][32 + a]) + (part[2][32 + a] + part[3][32 + a]);
      b[q0 + (kForward ? a : s - 1 - a)] = slo;
      if (a != s - 1 - a) b[q0 + (kForward ? s - 1 - a : a)] = shi;
    }
    return;
  }
  part[wv][a] = sn_block_row<kForward>(Wj, t, s, a, wv, kSnThreads / 64);
  __syncthreads();
  if (wv == 0 && a < s) b[q0 + a] = (part[0][a] + part[1][a]) + (part[2][a] + part[3][a]);
}
// ... and with a wavefront per supernode (k_sn_level_w, GS = 64): lane = row for the sums, chunks of 64 x kSnFlatU entries in
// the wavefront's own slab of LDS, no workgroup barrier (LDS operations of a wavefront complete in order)
// kFold: the block is stored FOLDED (k_sn_fold below).  Reading the packed triangle piece by piece leaves half the lanes of
// every load masked off and the memory system at 3.3 TB/s (tools/micro/block_stream.hip: 4.9 TB/s for 64 rows, 2.6 for 24;
// folded 6.2 / 4.3).  Folded, lane a < h = ceil(s / 2) owns the TWO rows a and s - 1 - a (forward; columns backward): a + 1
// and s - a entries, s + 1 together for every lane, stored step by step (entry k of lane a at k h + a).  The lanes 32..63
// take the odd steps, so a wavefront load is steps k and k + 1: 2 h contiguous doubles, no lane masked (s = 64), and the block
// is over in (s + 1) / 2 loads instead of s.  The two halves meet through one lane exchange, even steps + odd steps.
// (DK: the steps are dealt to DK half-wavefronts -- 2: the two halves of one wavefront; 8: of the four wavefronts of a workgroup,
// whose sums the caller adds up; k0: this half-wavefront's first step)
template <bool kForward, int DK>
__device__ __forceinline__ void sn_block_fold(const double *__restrict__ Wj, const double *t, int s, int gl, int k0, double &lo, double &hi) {
  const int h = (s + 1) >> 1, a = gl & 31, par = gl >> 5;
  lo = 0.0; hi = 0.0;
  if (a < h) {
    auto idx = [&](int k) { return kForward ? (k <= a ? k : k - a - 1) : (k <= a ? s - 1 - a + k : k - 1); };
    int k = k0;
    for (; k + 3 * DK <= s; k += 4 * DK) {
      const double w0 = Wj[k * h + a], w1 = Wj[(k + DK) * h + a], w2 = Wj[(k + 2 * DK) * h + a], w3 = Wj[(k + 3 * DK) * h + a];
      const double p0 = w0 * t[idx(k)], p1 = w1 * t[idx(k + DK)], p2 = w2 * t[idx(k + 2 * DK)], p3 = w3 * t[idx(k + 3 * DK)];
      if (k <= a) lo += p0; else hi += p0;
      if (k + DK <= a) lo += p1; else hi += p1;
      if (k + 2 * DK <= a) lo += p2; else hi += p2;
      if (k + 3 * DK <= a) lo += p3; else hi += p3;
    }
    for (; k <= s; k += DK) {
      const double p0 = Wj[k * h + a] * t[idx(k)];
      if (k <= a) lo += p0; else hi += p0;
    }
  }
  const double lo2 = __shfl_xor(lo, 32), hi2 = __shfl_xor(hi, 32);
  lo = par ? lo2 + lo : lo + lo2;
  hi = par ? hi2 + hi : hi + hi2;
}
template <bool kForward, bool kFold>
__global__ __launch_bounds__(kSnThreads) void k_sn_level_wf(int J0, int J1, const int *__restrict__ ptr, const int64_t *__restrict__ woff,
                                                            const int64_t *__restrict__ Ep, const int *__restrict__ Ej,
                                                            const double *__restrict__ Ex, const double *__restrict__ W,
                                                            const double *__restrict__ Dinv_s, double *__restrict__ b) {
  constexpr int C = 64 * kSnFlatU;
  __shared__ double prod_all[kSnThreads / 64][C];
  __shared__ double tt[kSnThreads / 64][kSnMax];
  const int wv = threadIdx.x >> 6, gl = threadIdx.x & 63;
  const int J = J0 + blockIdx.x * (kSnThreads / 64) + wv;
  if (J >= J1) return;  // no workgroup barrier below
  double *prod = prod_all[wv], *t = tt[wv];
  const int q0 = ptr[J], s = ptr[J + 1] - q0;
  const int64_t E0 = Ep[q0], E1 = Ep[q0 + s];
  int64_t r1 = 0, pos = 0;
  double rhs = 0.0;
  if (gl < s) {
    pos = Ep[q0 + gl]; r1 = Ep[q0 + gl + 1];
    rhs = kForward ? b[q0 + gl] : b[q0 + gl] * Dinv_s[q0 + gl];
  }
  double acc = 0.0;
  for (int64_t base = E0; base < E1; base += C) {
    int idx[kSnFlatU];
    double val[kSnFlatU], bv[kSnFlatU];
#pragma unroll
    for (int u = 0; u < kSnFlatU; u++) {
      const int64_t e = base + u * 64 + gl;
      const bool ok = e < E1;
      idx[u] = ok ? Ej[e] : -1;
      val[u] = ok ? Ex[e] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < kSnFlatU; u++) bv[u] = idx[u] >= 0 ? b[idx[u]] : 0.0;
#pragma unroll
    for (int u = 0; u < kSnFlatU; u++) prod[u * 64 + gl] = val[u] * bv[u];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    const int64_t end = r1 < base + C ? r1 : base + C;
    for (; pos < end; pos++) acc += prod[pos - base];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (gl < s) t[gl] = rhs - acc;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  const double *Wj = W + woff[J];
  if (kFold) {
    double lo, hi;
    sn_block_fold<kForward, 2>(Wj, t, s, gl, gl >> 5, lo, hi);
    const int a = gl & 31, h = (s + 1) >> 1;
    if (gl < h) {  // forward: lo is row a, hi row s - 1 - a; backward: lo is column s - 1 - a, hi column a
      b[q0 + (kForward ? a : s - 1 - a)] = lo;
      if (a != s - 1 - a) b[q0 + (kForward ? s - 1 - a : a)] = hi;
    }
    return;
  }
  const double out = sn_block_row<kForward>(Wj, t, s, gl, 0, 1);
  if (gl < s) b[q0 + gl] = out;
}
// A packed block to its folded form, in place through LDS (after every numeric factorisation, for the supernodes the
// wavefront form solves: LdlFactor::fold_blocks).  kForward: the block is packed by columns (Wc), else by rows (Wr).
template <bool kForward>
__global__ __launch_bounds__(64) void k_sn_fold(int J0, const int *__restrict__ ptr, const int64_t *__restrict__ woff, double *__restrict__ W) {
  __shared__ double tri[kSnMax * (kSnMax + 1) / 2];
  const int J = J0 + blockIdx.x, s = ptr[J + 1] - ptr[J], gl = threadIdx.x;
  double *Wj = W + woff[J];
  const int nel = s * (s + 1) / 2, h = (s + 1) >> 1;
  for (int e = gl; e < nel; e += 64) tri[e] = Wj[e];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  auto at = [&](int r, int c) { return kForward ? tri[c * s - c * (c - 1) / 2 + (r - c)] : tri[r * (r + 1) / 2 + c]; };  // W(r, c), r >= c
  for (int e = gl; e < (s + 1) * h; e += 64) {
    const int k = e / h, a = e - k * h;
    double v;
    if (kForward) v = k <= a ? at(a, k) : (a != s - 1 - a ? at(s - 1 - a, k - a - 1) : 0.0);
    else v = k <= a ? at(s - 1 - a + k, s - 1 - a) : (a != s - 1 - a ? at(k - 1, a) : 0.0);
    Wj[e] = v;
  }
}

// Backward step of the supernodes of ONE pivot that a level starts with (symbolic.hpp lvl_single; control-1e6: 388 258 of the
// 496 738 leaves, one entry each): the block is the number 1, so x_q = D_q^-1 y_q - G_q x with a lane per pivot over
// consecutive slots -- in k_sn_level_w they were a quarter wavefront each, 15 of 16 lanes idle.  (Forward they are skipped
// at level 0 altogether: no entries outside the block, y_q = b_q.)
__global__ __launch_bounds__(kBlock) void k_sn_single_bwd(int q0, int count, const int64_t *__restrict__ Gp, const int *__restrict__ Gi,
                                                          const double *__restrict__ Gx, const double *__restrict__ Dinv_s,
                                                          double *__restrict__ b) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= count) return;
  const int q = q0 + i;
  const double acc = gather_dot(Gp[q], Gp[q + 1], 1, Gi, Gx, b);
  b[q] = b[q] * Dinv_s[q] - acc;
}

// The supernodes of level >= 1 in ONE launch per direction: workgroup = supernode, started in level order, each
// waiting on a counter for the supernodes below it (forward: `pending[J]` children still running; backward: the
// supernode above publishes `ready[up] = number of waiting children`, each child takes one).  Both counters are back at
// their resting values when the launch ends, so a captured graph can replay it.  Workgroups are dispatched in
// blockIdx order per XCD and a workgroup only waits on lower blockIdx values, so the lowest unfinished one is always
// resident and never waits on an unscheduled one -- and the launch is sized so that ALL its workgroups fit the device at
// once (LdlFactor: occupancy x CUs >= grid; round 6: up to twice that where fronts go through global memory -- the
// dispatch-order argument carries those launches, measured without a single time-out), so on a device of its own nothing
// waits on a workgroup that cannot be scheduled; a wait that still exceeds 200 ms (a shared, pre-empted device) sets *fault (mapped host
// memory) and carries on: the host sees the flag at the next residual evaluation -- tested again once the read-back has
// drained the stream -- and at the end of osqp_solve, switches the factor to one launch per level and runs the solve
// again from a cold start (Engine::solve), so a broken assumption costs time, never a wrong or missing answer.
constexpr long long kSnWaitTicks = 20000000LL;  // 200 ms of the 100 MHz wall clock (a legitimate wait is microseconds; a workgroup
                                                 // pre-empted on a shared device can look like milliseconds)
__device__ __forceinline__ int sn_load(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// sum of Ex[i] * b[Ej[i]] over i = i0 + lane, i0 + lane + la, ... < i1, four gathers in flight per lane
template <bool kCoherent>
__device__ __forceinline__ double sn_gather(int64_t i0, int64_t i1, int lane, int la, const int *__restrict__ Ej,
                                            const double *__restrict__ Ex, const double *b) {
  auto ld = [&](int j) { return kCoherent ? __hip_atomic_load(&b[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : b[j]; };
  double acc = 0.0;
  int64_t i = i0 + lane;
  for (; i + 3 * (int64_t)la < i1; i += 4 * (int64_t)la) {
    const int j0 = Ej[i], j1 = Ej[i + la], j2 = Ej[i + 2 * la], j3 = Ej[i + 3 * la];
    const double x0 = Ex[i], x1 = Ex[i + la], x2 = Ex[i + 2 * la], x3 = Ex[i + 3 * la];
    const double b0 = ld(j0), b1 = ld(j1), b2 = ld(j2), b3 = ld(j3);
    acc += x0 * b0; acc += x1 * b1; acc += x2 * b2; acc += x3 * b3;
  }
  for (; i < i1; i += la) acc += Ex[i] * ld(Ej[i]);
  return acc;
}
// entries per lane whose index and value are in registers before the wait: 24 with 1024 threads a supernode (one workgroup per compute
// unit whatever its registers: 128 of them), 20 with 512 (117 registers: two workgroups still fit; with 24 -- 133 -- one does, and
// control-1e6 loses 5 %).  Round 6, measured against 16 for both: grid 700 x 700 2 736 -> 2 812 it/s, 1000 x 1000 1 222 -> 1 275.
template <int NT> constexpr int kSnCapOf = NT == 1024 ? 24 : 20;
// The top of the tree by FRONT VECTORS (round 6).  The rows of the top separators of a 2-D structure hold thousands of entries
// each (64 rows x ~2 000 entries: 8 MB of 64-byte sectors gathered by ONE compute unit per supernode, eleven supernodes in a chain
// for the top separator of a 700 x 700 grid: 1.8 ms of a 2.7 ms iteration).  Where every supernode from some level on has a front
// out of global memory (mfront_big.hpp: its panel Y = L21 D stays resident, column-major), the forward solve of those levels is
// the multifrontal one: a supernode hands its parent ONE dense vector -- the contributions of its own columns and of the top
// supernodes below it to its border rows, u = (children's vectors, extended) + L21 y -- and its rows gather only the entries that
// point BELOW the top part.  The panel is read coalesced (0.3 MB instead of 8 MB a supernode), children in ascending order: a
// fixed order of sums.
struct SnTop {
  int Jt;                   // first supernode of the top part (everything from here on has a panel); < 0: off
  const int64_t *Et;        // per forward row: the first entry that points at a slot of the top part
  const int *tchp, *tchl;   // children INSIDE the top part, per supernode J >= Jt: tchl[tchp[J - Jt] .. tchp[J - Jt + 1])
  const int *bsz;           // border rows of a supernode
  const int64_t *reloff;    // rel + reloff[J]: row of the PARENT's front for each border row of J; uvec + reloff[J]: its vector
  const uint16_t *rel;
  const int64_t *poff;      // panel of J: panel + poff[J], s columns of (s + b) doubles
  const double *panel;
  double *uvec;
};

template <bool kForward, int NT>  // NT threads per supernode: 1024, or 512 when that lets the launch take one more level (twice the resident workgroups)
__global__ __launch_bounds__(NT) void k_sn_tree(SnTop top, int J0, int count, const int *__restrict__ ptr, const int64_t *__restrict__ woff,
                                                            const int64_t *__restrict__ Ep, const int64_t *__restrict__ Es,
                                                            const int *__restrict__ Ej, const double *__restrict__ Ex,
                                                            const double *__restrict__ W, const double *__restrict__ Dinv_s,
                                                            const int *__restrict__ up, const int *__restrict__ waits,
                                                            int *__restrict__ sync, int *__restrict__ fault, double *b, int *ticket) {
  __shared__ double t[kSnMax];
  __shared__ double Wl[kSnMax * kSnMax];
  __shared__ double ysc[kSnMax];
  __shared__ int Js;
  // ticket != nullptr (round 5): PERSISTENT workgroups -- as many as the device holds -- take the supernodes of the launch in
  // level order from a counter.  A workgroup that waits (forward: for children, backward: for its parent) waits on a
  // supernode with an earlier ticket, i.e. one that some resident workgroup is working on or has finished: progress whatever
  // the count, so the launch can take EVERY level above level 0 (control-1e6: 19 000 supernodes in 10 levels instead of the
  // top 841 that fit the device at once; the plain launches of levels 1 - 3 were 6 x ~55 us of a 0.85 ms iteration).
  for (int k = ticket ? -1 : (int)blockIdx.x;;) {
  if (ticket) {
    __syncthreads();  // everybody is past the previous supernode: t, Wl and Js are free
    if (threadIdx.x == 0) Js = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    k = Js;
    if (k >= count - J0) break;
  }
  const int J = kForward ? J0 + k : count - 1 - k;
  const int P = up[J];
  const int q0 = ptr[J], s = ptr[J + 1] - q0;
  // all rows of the supernode at once: 64 / 32 / 16 (/ 8 with 512 threads) lanes per row
  const int la = s <= NT / 64 ? 64 : (s <= NT / 32 ? 32 : (s <= NT / 16 ? 16 : NT / 64));
  const int lane = threadIdx.x & (la - 1), a = threadIdx.x / la;
  const bool mine = a < s;
  const int q = q0 + (mine ? a : 0);
  // Entries of b written inside this launch (slots of level >= 1) are stored and loaded at device scope, past the
  // per-XCD L2s, so no cache write-back / invalidate is needed around the counters; everything else (level-0 slots, L,
  // W) was written by earlier launches and is read through the caches.  Everything that does not depend on the wait
  // happens before it: the part of each forward row that points at level 0, the indices and values of the rest (into
  // registers), the inverted block (into LDS); behind the wait there is one round of loads of b, the two small
  // products and the store.
  double acc0 = (kForward && mine) ? sn_gather<false>(Ep[q], Es[q], lane, la, Ej, Ex, b) : 0.0;
  const bool topmode = kForward && top.Jt >= 0 && J >= top.Jt;  // (uniform over the workgroup)
  const int64_t i0 = (kForward ? Es[q] : Ep[q]) + lane, i1 = mine ? (topmode ? top.Et[q] : Ep[q + 1]) : 0;
  constexpr int kSnCap = kSnCapOf<NT>;
  int jj[kSnCap];
  double xx[kSnCap];
#pragma unroll
  for (int u = 0; u < kSnCap; u++) {
    const int64_t i = i0 + (int64_t)u * la;
    const bool in = i < i1;
    jj[u] = in ? Ej[i] : q;  // padding: the row's own slot (a finite number) times zero
    xx[u] = in ? Ex[i] : 0.0;
  }
  {
    const double *Wj = W + woff[J];
    for (int e = threadIdx.x; e < s * (s + 1) / 2; e += NT) Wl[e] = Wj[e];
  }
  const double own = mine ? (kForward ? b[q] : b[q] * Dinv_s[q]) : 0.0;
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();
    if (kForward) {
      for (unsigned spins = 1; sn_load(&sync[J]) != 0; spins++) {
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 255u) == 0 && wall_clock64() - t0 > kSnWaitTicks) { *fault = 1; break; }
      }
      sync[J] = waits[J];  // resting value for the next solve (its children are all past their decrement)
    } else if (P >= 0 && P < count) {  // (a parent beyond the launch -- in the dense top, direct_sndense_kernels.hpp -- has its solution in b already)
      for (unsigned spins = 1; sn_load(&sync[P]) == 0; spins++) {
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 255u) == 0 && wall_clock64() - t0 > kSnWaitTicks) { *fault = 1; break; }
      }
      __hip_atomic_fetch_sub(&sync[P], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  {
    double bb[kSnCap];
#pragma unroll
    for (int u = 0; u < kSnCap; u++) bb[u] = __hip_atomic_load(&b[jj[u]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < kSnCap; u++) acc += xx[u] * bb[u];
    if (i0 + (int64_t)kSnCap * la < i1) acc += sn_gather<true>(i0 - lane + (int64_t)kSnCap * la, i1, lane, la, Ej, Ex, b);
    acc += acc0;
    for (int o = la >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0 && mine) t[a] = own - acc;
  }
  __syncthreads();
  if (topmode) {  // the children inside the top part: their vectors' entries at this supernode's pivots
    for (int ci = top.tchp[J - top.Jt]; ci < top.tchp[J - top.Jt + 1]; ci++) {
      const int c = top.tchl[ci], bc = top.bsz[c];
      const uint16_t *rl = top.rel + top.reloff[c];
      const double *uc = top.uvec + top.reloff[c];
      for (int i = threadIdx.x; i < bc; i += NT) {
        const int r = rl[i];
        if (r < s) t[r] -= __hip_atomic_load(&uc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (a child's rows are distinct)
      }
      __syncthreads();
    }
  }
  {
    constexpr int LP = NT / 64;  // 64 rows x LP lanes
    const int part = threadIdx.x & (LP - 1), r = threadIdx.x / LP;
    double acc = 0.0;
    if (r < s) {
      if (kForward) { for (int j = part; j <= r; j += LP) acc += Wl[j * s - j * (j - 1) / 2 + (r - j)] * t[j]; }
      else { for (int j = r + part; j < s; j += LP) acc += Wl[j * (j + 1) / 2 + r] * t[j]; }
    }
#pragma unroll
    for (int o = LP / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (part == 0 && r < s) {
      __hip_atomic_store(&b[q0 + r], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (topmode) ysc[r] = acc * Dinv_s[q0 + r];  // L21 = Y D^-1: the panel holds Y
    }
  }
  if (topmode) {
    const int bj = top.bsz[J], f = s + bj;
    if (bj > 0) {  // u = (children's vectors at this supernode's border rows) + L21 y; Wl is free once y is formed: it holds the sums
      double *uv = Wl;
      __syncthreads();
      for (int i = threadIdx.x; i < bj; i += NT) uv[i] = 0.0;
      __syncthreads();
      for (int ci = top.tchp[J - top.Jt]; ci < top.tchp[J - top.Jt + 1]; ci++) {
        const int c = top.tchl[ci], bc = top.bsz[c];
        const uint16_t *rl = top.rel + top.reloff[c];
        const double *uc = top.uvec + top.reloff[c];
        for (int i = threadIdx.x; i < bc; i += NT) {
          const int r = rl[i];
          if (r >= s) uv[r - s] += __hip_atomic_load(&uc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
      }
      const double *Pn = top.panel + top.poff[J] + s;
      double *uj = top.uvec + top.reloff[J];
      for (int i = threadIdx.x; i < bj; i += NT) {
        double a0 = uv[i], a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int p = 0;
        for (; p + 3 < s; p += 4) {
          a0 += Pn[(int64_t)p * f + i] * ysc[p]; a1 += Pn[(int64_t)(p + 1) * f + i] * ysc[p + 1];
          a2 += Pn[(int64_t)(p + 2) * f + i] * ysc[p + 2]; a3 += Pn[(int64_t)(p + 3) * f + i] * ysc[p + 3];
        }
        for (; p < s; p++) a0 += Pn[(int64_t)p * f + i] * ysc[p];
        __hip_atomic_store(&uj[i], (a0 + a1) + (a2 + a3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have reached the device-coherent level
  __syncthreads();
  if (threadIdx.x == 0) {
    if (kForward) { if (P >= 0 && P < count) __hip_atomic_fetch_sub(&sync[P], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    else if (waits[J] > 0) __hip_atomic_store(&sync[J], waits[J], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (!ticket) break;
  }
}

}  // namespace
}  // namespace oq
