// direct_dense_kernels.hpp -- the dense top block: Schur complement, explicit inverse by (block) Gauss-Jordan sweeps, the product of a solve
// Part of the direct KKT back-end: included by direct.hip (one translation unit, one anonymous namespace); split out in round 6
// for reviewability -- direct.hip keeps the factor object (LdlFactor), the back-end (Direct) and the set-up decisions.
#pragma once
#include "engine.hpp"

namespace oq {
namespace {

// ------------------------------------------------------------------ dense top block
// The top of the elimination tree of a problem with a few dense rows / columns (a budget constraint, a factor
// model, a data matrix) is a dense block: every pivot its own level, rows as long as the block.  Its triangular
// solves are a chain of k dependent steps forward and k backward whatever the kernel.  So the last kD pivots are
// not factorised at all: their Schur complement S0 = K22 - L21 D1 L21' is assembled as a dense kD x kD array,
// inverted once per factorisation by kD Gauss-Jordan sweeps (two per launch, ping-pong buffers, the pivots
// are the same Schur complements LDL' would meet, so the inertia count is unchanged), and a solve replaces both
// chains by one dense product x2 = S0^-1 (b2 - L21 y1).
// wave per entry (i, k) of columns [b0, b1) of L's pattern inside the block; the work rows w_k hold L_kj d_j (k_ldl_wrow)
__global__ __launch_bounds__(kBlock) void k_dense_entries(int b0, int b1, int cD, int kD, int ld, int N, const int64_t *__restrict__ Lp,
                                                          const int *__restrict__ Li, const double *__restrict__ Lx,
                                                          const int64_t *__restrict__ Rp, const int *__restrict__ Rj,
                                                          const int64_t *__restrict__ Rmap, const double *__restrict__ W,
                                                          double *__restrict__ S0) {
  const int lane = threadIdx.x & 63;
  const int64_t e = Lp[b0] + (((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6);
  if (e >= Lp[b1]) return;
  int lo = b0, hi = b1;
  while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (Lp[mid] <= e) lo = mid; else hi = mid; }
  const int k = lo, i = Li[e];
  const double *w = W + (size_t)(k - b0) * N;
  double acc = 0.0;
  for (int64_t q = Rp[i] + lane; q < Rp[i + 1]; q += 64) {
    const int j = Rj[q];
    if (j >= cD) break;
    acc += Lx[Rmap[q]] * w[j];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) {
    const double v = Lx[e] - acc;
    S0[(size_t)(i - cD) + (size_t)(k - cD) * ld] = v;
    S0[(size_t)(k - cD) + (size_t)(i - cD) * ld] = v;
  }
}
// wave per column k of [b0, b1): diagonal of the Schur complement
__global__ __launch_bounds__(kBlock) void k_dense_diag(int b0, int b1, int cD, int kD, int ld, int N, const double *__restrict__ Lx,
                                                       const int64_t *__restrict__ Rp, const int *__restrict__ Rj,
                                                       const int64_t *__restrict__ Rmap, const double *__restrict__ D,
                                                       const double *__restrict__ W, double *__restrict__ S0) {
  const int lane = threadIdx.x & 63;
  const int k = b0 + (int)(((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6);
  if (k >= b1) return;
  const double *w = W + (size_t)(k - b0) * N;
  double acc = 0.0;
  for (int64_t q = Rp[k] + lane; q < Rp[k + 1]; q += 64) {
    const int j = Rj[q];
    if (j >= cD) break;
    acc += Lx[Rmap[q]] * w[j];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) S0[(size_t)(k - cD) * (ld + 1)] = D[k] - acc;
}
// one Gauss-Jordan sweep on pivot p, out of place (no ordering between the threads of a launch is needed)
__global__ __launch_bounds__(kBlock) void k_dense_sweep(int kD, int p, const double *__restrict__ Sold, double *__restrict__ Snew,
                                                        int *__restrict__ status) {
  const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (idx >= (int64_t)kD * kD) return;
  const int i = (int)(idx % kD), j = (int)(idx / kD);
  const double piv = Sold[(size_t)p * (kD + 1)];
  const double ip = 1.0 / piv;
  const double aip = Sold[(size_t)i + (size_t)p * kD], apj = Sold[(size_t)p + (size_t)j * kD];
  double v;
  if (i == p && j == p) v = -ip;
  else if (i == p) v = apj * ip;
  else if (j == p) v = aip * ip;
  else v = Sold[idx] - aip * apj * ip;
  Snew[idx] = v;
  if (idx == 0) {
    if (piv == 0.0 || piv != piv) atomicOr(&status[0], 1);
    else if (piv > 0.0) atomicAdd(&status[1], 1);
  }
}
// Two consecutive sweeps (pivots p, then q = p + 1) in one pass over the array: every thread recomputes the four
// once-swept values its element needs (S'_ij, S'_iq, S'_qj, S'_qq) with the very expressions of k_dense_sweep, so
// the result is bit-identical to two launches at half the traffic.  (A block sweep through the inverse of the 2 x 2
// pivot block is NOT: on the quasi-definite Schur complement that block can be badly conditioned -- sigma next to a
// large off-diagonal -- and the feasibility test of the reference lost its accuracy with it.)
__device__ __forceinline__ double sweep_value(bool row_p, bool col_p, double xij, double xip, double xpj, double ip) {
  if (row_p && col_p) return -ip;
  if (row_p) return xpj * ip;
  if (col_p) return xip * ip;
  return xij - xip * xpj * ip;
}
__global__ __launch_bounds__(kBlock) void k_dense_sweep2(int kD, int p, const double *__restrict__ Sold, double *__restrict__ Snew,
                                                         int *__restrict__ status) {
  const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (idx >= (int64_t)kD * kD) return;
  const int i = (int)(idx % kD), j = (int)(idx / kD), q = p + 1;
  const size_t cp = (size_t)p * kD, cq = (size_t)q * kD, cj = (size_t)j * kD;
  const double piv = Sold[cp + p], ip = 1.0 / piv;
  const double sip = Sold[cp + i], spj = Sold[cj + p], siq = Sold[cq + i], sqj = Sold[cj + q];
  const double spq = Sold[cq + p], sqp = Sold[cp + q], sqq = Sold[cq + q];
  // after the first sweep
  const double t_ij = sweep_value(i == p, j == p, Sold[idx], sip, spj, ip);
  const double t_iq = sweep_value(i == p, false, siq, sip, spq, ip);
  const double t_qj = sweep_value(false, j == p, sqj, sqp, spj, ip);
  const double piv2 = sqq - sqp * spq * ip, ip2 = 1.0 / piv2;
  Snew[idx] = sweep_value(i == q, j == q, t_ij, t_iq, t_qj, ip2);
  if (idx == 0) {
    if (piv == 0.0 || piv != piv || piv2 == 0.0 || piv2 != piv2) atomicOr(&status[0], 1);
    else atomicAdd(&status[1], (piv > 0.0 ? 1 : 0) + (piv2 > 0.0 ? 1 : 0));
  }
}
// x2 = -(S v) with S = -S0^-1 as the sweeps leave it (symmetric: row a is read as column a, contiguous); wave per row
// The same product from the LOWER triangle of the (symmetric) array alone: half the bytes of what is the iteration of a dense-P
// problem (6000 pivots: 288 MB per product, 60 us of a 106 us iteration).  A workgroup per 64 x 64 tile (rb, cb), rb >= cb,
// brought to LDS with full-width loads (lane = row: 512 contiguous bytes per wavefront load, 16 in flight per lane); from LDS
// thread c adds up S(b, c) v[b] over the tile's rows (its share of out[column block cb]) and thread b adds up S(b, c) v[c] over
// the tile's columns (the mirrored entries: its share of out[row block rb]; not for a diagonal tile, whose both halves are in
// the tile).  The shares go to two [blocks x blocks x 64] buffers and k_dense_sym_reduce adds them up in a fixed order.
constexpr int kDsT = 64;
__global__ __launch_bounds__(256) void k_dense_apply_sym(int kD, int ld, int nb, const double *__restrict__ S, const double *__restrict__ v,
                                                         double *__restrict__ P1, double *__restrict__ P2) {
  __shared__ double tile[kDsT][kDsT + 1];
  __shared__ double vr[kDsT], vc[kDsT], part[4][kDsT];
  // tile t of the lower triangle, row by row: rb (rb + 1) / 2 + cb
  const int t = blockIdx.x;
  int rb = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((rb + 1) * (rb + 2) / 2 <= t) rb++;
  while (rb * (rb + 1) / 2 > t) rb--;
  const int cb = t - rb * (rb + 1) / 2;
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  {
    const double *base = S + (size_t)(cb * kDsT) * ld + (size_t)rb * kDsT + lane;
    double w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = base[(size_t)(wv * 16 + i) * ld];
    if (wv == 0) { const int g = rb * kDsT + lane; vr[lane] = g < kD ? v[g] : 0.0; }
    if (wv == 1) { const int g = cb * kDsT + lane; vc[lane] = g < kD ? v[g] : 0.0; }
#pragma unroll
    for (int i = 0; i < 16; i++) tile[wv * 16 + i][lane] = w[i];  // tile[c][b] = S(row rb*64 + b, column cb*64 + c)
  }
  __syncthreads();
  const int half = (tid >> 6) & 1;
  double acc = 0.0;
  if (tid < 128) {  // column sums: thread c over the rows b of its half
    for (int b = 32 * half; b < 32 * half + 32; b++) acc += tile[lane][b] * vr[b];
  } else {          // row sums: thread b over the columns c of its half
    for (int c = 32 * half; c < 32 * half + 32; c++) acc += tile[c][lane] * vc[c];
  }
  part[wv][lane] = acc;
  __syncthreads();
  if (wv == 0) P1[((size_t)cb * nb + rb) * kDsT + lane] = part[0][lane] + part[1][lane];
  else if (wv == 2 && rb != cb) P2[((size_t)rb * nb + cb) * kDsT + lane] = part[2][lane] + part[3][lane];
}
// out[a] = -(sum over the tiles below and on the diagonal of a's column block + sum over the tiles left of the diagonal of a's
// row block), the four wavefronts of a workgroup taking every fourth term, met in a fixed order
// (`omap` != nullptr: entry a goes to out[omap[a]] -- the dense top over the supernodes keeps its pivots in an order of its own)
__global__ __launch_bounds__(256) void k_dense_sym_reduce(int kD, int nb, const double *__restrict__ P1, const double *__restrict__ P2,
                                                          double *__restrict__ out, const int *__restrict__ omap) {
  __shared__ double part[4][kDsT];
  const int B = blockIdx.x, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double acc = 0.0;
  // terms 0 .. nb - B - 1: P1[B][B + j]; then B terms P2[B][j]
  const int n1 = nb - B, n = n1 + B;
  int j = wv;
  for (; j + 12 < n; j += 16) {
    double x[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int jj = j + 4 * u;
      x[u] = jj < n1 ? P1[((size_t)B * nb + B + jj) * kDsT + lane] : P2[((size_t)B * nb + (jj - n1)) * kDsT + lane];
    }
    acc += x[0]; acc += x[1]; acc += x[2]; acc += x[3];
  }
  for (; j < n; j += 4) acc += j < n1 ? P1[((size_t)B * nb + B + j) * kDsT + lane] : P2[((size_t)B * nb + (j - n1)) * kDsT + lane];
  part[wv][lane] = acc;
  __syncthreads();
  const int a = B * kDsT + lane;
  if (wv == 0 && a < kD) out[omap ? omap[a] : a] = -((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]));
}
__global__ __launch_bounds__(kBlock) void k_dense_apply(int kD, int ld, const double *__restrict__ S, const double *__restrict__ v,
                                                        double *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int a = (int)(((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6);
  if (a >= kD) return;
  const double *col = S + (size_t)a * ld;
  double acc = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;  // four loads in flight per lane (a 6000-row block: 94 rounds of one)
  int b = lane;
  for (; b + 192 < kD; b += 256) { acc += col[b] * v[b]; acc1 += col[b + 64] * v[b + 64]; acc2 += col[b + 128] * v[b + 128]; acc3 += col[b + 192] * v[b + 192]; }
  for (; b < kD; b += 64) acc += col[b] * v[b];
  acc = (acc + acc1) + (acc2 + acc3);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) out[a] = -acc;
}
// ---- the same inverse by BLOCK sweeps on the fp64 matrix cores (blocks of kGjK pivots; large dense blocks) -----------
// With B the pivot indices of a step, G = S_BB^-1, the kGjK single sweeps of the block amount to
//   S_RR <- S_RR - S_RB G S_BR,   S_BR <- G S_BR (and its mirror),   S_BB <- -G
// i.e. one rank-kGjK update of the whole array -- v_mfma_f64_16x16x4_f64 -- instead of kGjK passes over it: the array
// (288 MB at 6000 pivots) is read and written once per 32 pivots, n^3 multiply-adds at matrix-core rate.  Three launches
// per step: (1) one workgroup sweeps the pivot block by itself, pivot by pivot (the pivots are the ones the single sweeps
// and LDL' would meet: inertia and zero-pivot checks unchanged), leaving T = -G; (2) the row panel W = G S_B: (kGjK x ld)
// and a copy C of S_B: as it was; (3) the update of the tiles on and below the diagonal, each written to both triangles
// through an LDS transposition, so that the array stays bit-symmetric.  The array is padded to a multiple of 64 (identity
// on the padding: its sweeps change nothing).
// Round 5: blocks of 64 pivots (32 before): half the passes over the array, twice the matrix-core work per tile load; the
// pivot block -- the one serial piece of a step, a single workgroup -- is swept by 1024 threads (four elements each per pivot).
constexpr int kGjK = 64, kGjPivotThreads = 1024;
__global__ __launch_bounds__(kGjPivotThreads) void k_gj_pivot(int kD, int ld, int p0, const double *__restrict__ S, double *__restrict__ T,
                                                              int *__restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) double gj_lds[];
  double (*buf)[kGjK][kGjK + 1] = (double (*)[kGjK][kGjK + 1])gj_lds;  // [2][kGjK][kGjK + 1]
  const int t = threadIdx.x;
  for (int e = t; e < kGjK * kGjK; e += kGjPivotThreads) { const int i = e % kGjK, j = e / kGjK; buf[0][i][j] = S[(size_t)(p0 + i) + (size_t)(p0 + j) * ld]; }
  __syncthreads();
  int cur = 0, pos = 0, bad = 0;
  for (int p = 0; p < kGjK; p++) {
    const double piv = buf[cur][p][p], ip = 1.0 / piv;
    for (int e = t; e < kGjK * kGjK; e += kGjPivotThreads) {
      const int i = e % kGjK, j = e / kGjK;
      buf[cur ^ 1][i][j] = sweep_value(i == p, j == p, buf[cur][i][j], buf[cur][i][p], buf[cur][p][j], ip);
    }
    if (p0 + p < kD) { if (piv == 0.0 || piv != piv) bad = 1; else if (piv > 0.0) pos++; }
    cur ^= 1;
    __syncthreads();
  }
  for (int e = t; e < kGjK * kGjK; e += kGjPivotThreads) { const int i = e % kGjK, j = e / kGjK; T[i + j * kGjK] = buf[cur][i][j]; }
  if (t == 0) { if (bad) atomicOr(&status[0], 1); atomicAdd(&status[1], pos); }
}
// The same sweeps with the block in REGISTERS (round 6: the pivot kernel is the one serial piece of a step -- a single workgroup,
// 64 dependent pivots -- and in LDS with a barrier and twelve LDS operations per element and pivot it took 67 us of a ~100 us
// step; a dense top of 2 900 pivots is 46 steps).  256 threads: thread (c, h) holds rows 16 h .. 16 h + 15 of column c.  Per
// pivot p (unrolled: every register index and the pivot's lane are compile-time constants): the pivot row comes through LDS
// (written by the wavefront that holds it, one barrier per pivot, two buffers), the pivot column of a wavefront's own rows
// is in ITS lane p -- sixteen v_readlane, no LDS --, and every element is swept by the very expression of the LDS form
// (sweep_value): the same bits.
constexpr int kGjPivotRThreads = 256;
__device__ __forceinline__ double gj_readlane(double x, int lane) {
  union { double d; int i[2]; } u;
  u.d = x;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
  return u.d;
}
// The diagonal tile is bit-symmetric (assembled and updated that way) and the sweeps keep it so -- the mirror of an element is the
// same expression with the two factors of its product swapped -- so thread (c, h) reads ITS elements (16 h + k, c) from the places of
// their mirrors (c, 16 h + k), the lanes along the contiguous direction, and writes T the same way: no transposition through LDS.
__device__ __forceinline__ void gj_pivot_sweep(int kD, int ld, int p0, const double *S, double *__restrict__ T, int *__restrict__ status,
                                               double (*rowp)[kGjK]) {
  const int t = threadIdx.x, c = t & 63, h = t >> 6;
  double v[16];
#pragma unroll
  for (int k = 0; k < 16; k++) v[k] = S[(size_t)(p0 + c) + (size_t)(p0 + 16 * h + k) * ld];
  int pos = 0, bad = 0;
#pragma unroll
  for (int p = 0; p < kGjK; p++) {
    const int hp = p >> 4, rp = p & 15;
    if (h == hp) rowp[p & 1][c] = v[rp];
    __syncthreads();
    const double xpj = rowp[p & 1][c], piv = rowp[p & 1][p], ip = 1.0 / piv;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const double xip = gj_readlane(v[k], p);
      v[k] = sweep_value(k == rp && h == hp, c == p, v[k], xip, xpj, ip);
    }
    if (p0 + p < kD) { if (piv == 0.0 || piv != piv) bad = 1; else if (piv > 0.0) pos++; }
  }
#pragma unroll
  for (int k = 0; k < 16; k++) T[c + (16 * h + k) * kGjK] = v[k];
  if (t == 0) { if (bad) atomicOr(&status[0], 1); atomicAdd(&status[1], pos); }
}
__global__ __launch_bounds__(kGjPivotRThreads) void k_gj_pivot_r(int kD, int ld, int p0, const double *__restrict__ S, double *__restrict__ T,
                                                                 int *__restrict__ status) {
  __shared__ double rowp[2][kGjK];
  gj_pivot_sweep(kD, ld, p0, S, T, status, rowp);
}
// thread per column j of the array: W[k][j] = sum_l G[k][l] S[p0 + l][j], C[k][j] = S[p0 + k][j]  (G = -T)
// (`cols` != nullptr, round 6: the launch covers only the listed column blocks -- the ones coupled with the pivot block in a
// block-sparse array, direct.hip gj_symbolic -- and the update below only the tiles between them)
__global__ __launch_bounds__(256) void k_gj_panel(int ld, int p0, const double *__restrict__ S, const double *__restrict__ T,
                                                  double *__restrict__ Wp, double *__restrict__ Cp, const int *__restrict__ cols) {
  __shared__ double G[kGjK][kGjK];
  for (int e = threadIdx.x; e < kGjK * kGjK; e += 256) G[e / kGjK][e % kGjK] = -T[e];  // (T is bit-symmetric, gj_pivot_sweep: no transposition, no bank conflicts)
  __syncthreads();
  // 64 columns per workgroup, its four wavefronts a quarter of the panel's rows each (round 5: a thread per column and all
  // kGjK rows left the 6000-column panel of equality_qp on 24 compute units, 86 us per step)
  const int jb = cols ? cols[blockIdx.x] : (int)blockIdx.x, c = threadIdx.x & 63, kq = threadIdx.x >> 6;
  const int j0 = jb * 64, j = j0 + c;
  if (j0 >= ld) return;
  // the 64 x 64 piece S(p0 + l, j0 + c) from the LOWER triangle (round 5: the sweeps keep only that one current, the mirror is
  // written once at the end; a diagonal tile is whole): left of the pivot block it sits in its own place, rows contiguous;
  // right of it in the place of its mirror, columns contiguous.  Round 6: brought to LDS with the lanes along the contiguous
  // direction either way (a thread per column reading its 64 rows was 64 cache lines per load instruction on the left side:
  // 21 us a step of a 4 160-pivot block) -- slot (l, c ^ l): no bank conflicts for lanes along l or along c.
  __shared__ double tile[kGjK * kGjK];
  if (j0 <= p0) {
    for (int e = threadIdx.x; e < kGjK * kGjK; e += 256) { const int l = e & 63, cc = e >> 6; tile[l * 64 + (cc ^ l)] = S[(size_t)(p0 + l) + (size_t)(j0 + cc) * ld]; }
  } else {
    for (int e = threadIdx.x; e < kGjK * kGjK; e += 256) { const int cc = e & 63, l = e >> 6; tile[l * 64 + (cc ^ l)] = S[(size_t)(j0 + cc) + (size_t)(p0 + l) * ld]; }
  }
  __syncthreads();
  double s[kGjK];
#pragma unroll
  for (int l = 0; l < kGjK; l++) s[l] = tile[l * 64 + (c ^ l)];
#pragma unroll 4
  for (int k = kq * (kGjK / 4); k < (kq + 1) * (kGjK / 4); k++) {
    double w = 0.0;
#pragma unroll
    for (int l = 0; l < kGjK; l++) w = __builtin_fma(G[k][l], s[l], w);
    Wp[(size_t)k * ld + j] = w;
    Cp[(size_t)k * ld + j] = tile[k * 64 + (c ^ k)];
  }
}
typedef double gj_d4 __attribute__((ext_vector_type(4)));
// workgroup (I, J), I >= J: the 64 x 64 tile of rows I, columns J and its mirror; wavefront w its 32 x 32 quadrant
__device__ __forceinline__ void gj_tile_update(int I, int J, int ld, int p0, double *S, const double *__restrict__ T,
                                               const double *__restrict__ Wp, const double *__restrict__ Cp, double (*tr)[16][17]) {
  if (J > I) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wi = w >> 1, wj = w & 1;
  if (I == J && wi < wj) return;  // the upper quadrant of a diagonal tile is the mirror of the lower one
  const int lr = lane >> 4, lc = lane & 15;
  const int r0 = I * 64 + wi * 32, c0 = J * 64 + wj * 32;  // rows r0 .. r0 + 32, columns c0 .. c0 + 32
  const bool rowsB = I * 64 == p0, colsB = J * 64 == p0;   // the pivot block is one 64 x 64 tile (64 pivots, 64-aligned)
#pragma unroll
  for (int ti = 0; ti < 2; ti++)
#pragma unroll
    for (int tj = 0; tj < 2; tj++) {
      const int ri = r0 + ti * 16, cj = c0 + tj * 16;
      if (ri < cj) continue;  // above the diagonal inside a diagonal quadrant: written by its mirror
      gj_d4 acc = {0.0, 0.0, 0.0, 0.0};
      double nv[4];
      if (!rowsB && !colsB) {
#pragma unroll
        for (int kk = 0; kk < kGjK / 4; kk++) {
          const double aop = Wp[(size_t)(4 * kk + lr) * ld + cj + lc];  // A[m = lane & 15][k = lane >> 4] = W[k][cj + m]
          const double bop = Cp[(size_t)(4 * kk + lr) * ld + ri + lc];  // B[k = lane >> 4][n = lane & 15] = C[ri + n][k]
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; r++) nv[r] = S[(size_t)(ri + lc) + (size_t)(cj + lr + 4 * r) * ld] - acc[r];  // D[m = lr + 4 r][n = lc]
      } else {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int i = ri + lc, j = cj + lr + 4 * r;
          nv[r] = (rowsB && colsB) ? T[(i - p0) + (j - p0) * kGjK] : (rowsB ? Wp[(size_t)(i - p0) * ld + j] : Wp[(size_t)(j - p0) * ld + i]);
        }
      }
      // both triangles: the tile as computed (rows contiguous across the lanes) and its transpose through LDS
#pragma unroll
      for (int r = 0; r < 4; r++) tr[w][lr + 4 * r][lc] = nv[r];  // tr[column offset][row offset]
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wavefront's own LDS writes before its reads (one wavefront per slab)
      if (ri == cj) {  // a diagonal 16 x 16 tile: the lower triangle decides
#pragma unroll
        for (int r = 0; r < 4; r++) { const int m = lr + 4 * r; if (lc < m) nv[r] = tr[w][lc][m]; }
      }
#pragma unroll
      for (int r = 0; r < 4; r++) S[(size_t)(ri + lc) + (size_t)(cj + lr + 4 * r) * ld] = nv[r];
      if (ri != cj && I == J) {  // the mirror only inside a diagonal 64 x 64 tile (the pivot kernel reads those whole); the rest of
                                 // the upper triangle is written once, when the sweeps are done (k_gj_mirror)
#pragma unroll
        for (int r = 0; r < 4; r++) S[(size_t)(cj + lc) + (size_t)(ri + lr + 4 * r) * ld] = tr[w][lc][lr + 4 * r];  // S[j][i] = value of (i, j)
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
    }
}
// The launch of a step: a workgroup per tile -- from a list (`tiles`: the tiles the block pattern says can change, direct.hip
// gj_symbolic) or, without one, every tile on and below the diagonal by its running number -- and, with `next` != 0, ONE MORE
// workgroup that prepares the NEXT step while the others stream the array (round 6: the pivot block was a launch of its own, a
// single workgroup for 38 us between two updates of ~35 us): it owns the next pivot tile (I = J = p0 / 64 + 1; the other
// workgroups leave it alone), brings it up to date (next == 2; next == 1: this step does not touch it), sweeps it in registers
// and leaves -G in Tnext.
__global__ __launch_bounds__(256) void k_gj_update(int ld, int p0, double *S, const double *__restrict__ T, const double *__restrict__ Wp,
                                                   const double *__restrict__ Cp, const int *__restrict__ tiles, int ntiles, int next,
                                                   double *__restrict__ Tnext, int kD, int *__restrict__ status) {
  __shared__ double tr[4][16][17];
  __shared__ double rowp[2][kGjK];
  const int nextb = p0 / kGjK + 1;
  if (next && blockIdx.x == 0) {  // (the FIRST workgroup of the launch: it is the longest one -- started last it trailed the launch by its whole length)
    if (next == 2) gj_tile_update(nextb, nextb, ld, p0, S, T, Wp, Cp, tr);
    __threadfence_block();
    __syncthreads();
    gj_pivot_sweep(kD, ld, nextb * kGjK, S, Tnext, status, rowp);
    return;
  }
  int I, J;
  const int t = (int)blockIdx.x - (next ? 1 : 0);
  if (tiles) { I = tiles[2 * t]; J = tiles[2 * t + 1]; }
  else {  // tile t of the lower triangle, row by row: I (I + 1) / 2 + J
    I = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((I + 1) * (I + 2) / 2 <= t) I++;
    while (I * (I + 1) / 2 > t) I--;
    J = t - I * (I + 1) / 2;
  }
  if (next && I == nextb && J == nextb) return;
  gj_tile_update(I, J, ld, p0, S, T, Wp, Cp, tr);
}
// ---- the Schur complement of a LARGE dense block on the matrix cores (round 5) ---------------------------------------------
// S0 = K22 - L21 D1 L21' entry by entry is a sparse dot product per entry of the block (k_dense_entries: 3.6e7 wavefronts for
// the 6000-pivot block of equality_qp, 22 ms -- as much as the whole inversion).  When L21 is not very sparse the same sum is
// a symmetric rank-k update: 64 columns of L at a time are spread out as dense 64 x ld panels (W = the column times its pivot,
// C = the column) and k_gj_update -- the rank-64 matrix-core update of the block sweeps, with no pivot block (p0 = -64) --
// subtracts W' C from the array; S0 starts as K22.  The zeros it multiplies are cheaper than the gathers they replace.
__global__ __launch_bounds__(kBlock) void k_dense_init(int cD, int N, int ld, const int64_t *__restrict__ Lp, const int *__restrict__ Li,
                                                       const int *__restrict__ Lcol, const double *__restrict__ Lx,
                                                       const double *__restrict__ D, double *__restrict__ S0) {
  const int64_t e = Lp[cD] + (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e < Lp[N]) {
    const int k = Lcol[e] - cD, i = Li[e] - cD;
    const double v = Lx[e];
    S0[(size_t)i + (size_t)k * ld] = v;
    S0[(size_t)k + (size_t)i * ld] = v;
  }
  const int64_t d = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (d < N - cD) S0[(size_t)d * (ld + 1)] = D[cD + d];
}
// wavefront w of the launch: column cols[c0 + w] of L, its entries in the rows of the block into row w of the two panels
__global__ __launch_bounds__(kBlock) void k_dense_chunk(int c0, int ncols, int cD, int ld, const int *__restrict__ cols,
                                                        const int64_t *__restrict__ Lp, const int *__restrict__ Li,
                                                        const double *__restrict__ Lx, const double *__restrict__ D,
                                                        double *__restrict__ Wp, double *__restrict__ Cp) {
  const int lane = threadIdx.x & 63, w = (int)(((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6);
  if (w >= kGjK || c0 + w >= ncols) return;
  const int j = cols[c0 + w];
  const double dj = D[j];
  for (int64_t e = Lp[j] + lane; e < Lp[j + 1]; e += 64) {
    const int i = Li[e];
    if (i < cD) continue;
    const double v = Lx[e];
    Wp[(size_t)w * ld + (i - cD)] = v * dj;
    Cp[(size_t)w * ld + (i - cD)] = v;
  }
}
// upper triangle := transpose of the lower one, tile by tile through LDS (both sides contiguous)
__global__ __launch_bounds__(256) void k_gj_mirror(int ld, double *__restrict__ S) {
  const int I = blockIdx.y, J = blockIdx.x;
  if (J >= I) return;
  __shared__ double tile[64][65];
  for (int e = threadIdx.x; e < 64 * 64; e += 256) { const int r = e & 63, c = e >> 6; tile[c][r] = S[(size_t)(I * 64 + r) + (size_t)(J * 64 + c) * ld]; }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 64; e += 256) { const int r = e & 63, c = e >> 6; S[(size_t)(J * 64 + r) + (size_t)(I * 64 + c) * ld] = tile[r][c]; }
}
__global__ __launch_bounds__(kBlock) void k_gj_pad(int kD, int ld, double *__restrict__ S) {
  const int i = kD + blockIdx.x * kBlock + threadIdx.x;
  if (i < ld) S[(size_t)i * (ld + 1)] = 1.0;
}
__global__ __launch_bounds__(kBlock) void k_gather_csr(int64_t nnz, const int64_t *__restrict__ Rmap, const double *__restrict__ Lx,
                                                       double *__restrict__ Rx) {
  int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (q < nnz) Rx[q] = Lx[Rmap[q]];
}

// sum of val[q] * b[idx[q]] over q = q0, q0 + stride, ... < q1: four gathers in flight per lane, added in the order of the plain
// loop (a plain loop is one dependent index -> value round trip per entry)
__device__ __forceinline__ double gather_dot(int64_t q0, int64_t q1, int stride, const int *__restrict__ idx, const double *__restrict__ val,
                                             const double *b) {
  double acc = 0.0;
  int64_t q = q0;
  for (; q + 3 * (int64_t)stride < q1; q += 4 * (int64_t)stride) {
    const int j0 = idx[q], j1 = idx[q + stride], j2 = idx[q + 2 * (int64_t)stride], j3 = idx[q + 3 * (int64_t)stride];
    const double x0 = val[q], x1 = val[q + stride], x2 = val[q + 2 * (int64_t)stride], x3 = val[q + 3 * (int64_t)stride];
    const double b0 = b[j0], b1 = b[j1], b2 = b[j2], b3 = b[j3];
    acc += x0 * b0; acc += x1 * b1; acc += x2 * b2; acc += x3 * b3;
  }
  for (; q < q1; q += stride) acc += val[q] * b[idx[q]];
  return acc;
}

}  // namespace
}  // namespace oq
