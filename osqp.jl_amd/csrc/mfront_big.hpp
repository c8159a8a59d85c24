// mfront_big.hpp -- fronts that do not fit one workgroup's LDS (round 6), included by direct.hip after mfront.hpp.
//
// Up to round 5 ONE supernode whose front had more than 192 rows sent the whole matrix back to the level-by-level
// factorisation (two launches per PIVOT level): every 2-D structure does that -- a g x g grid has separators of g nodes, so
// fronts of up to ~2 g rows.  Such a front is now factorised out of global memory (it is L2-sized: 700 rows x 64 pivots
// = 0.36 MB of panel), in two launches per supernode level next to the LDS launches of the level's small fronts:
//
//   k_mfb_panel   one workgroup per front.  The s x s pivot block lives in LDS (assembled, factorised and inverted there
//                 exactly as mfront.hpp does for a small front); the rows below it -- the b x s panel F21 -- are assembled
//                 in a global scratch (row-major, s doubles per row) by OWNER-COMPUTES gathers from the children's update
//                 matrices (a per-child inverse map  row of this front -> row of the child  in LDS: every entry is summed
//                 by one thread, children in ascending order: a fixed order of sums, no atomics), then turned into
//                 Y = F21 L11^-T = L21 D with the inverted block, written back to the scratch, and scattered to Lx as L21.
//   k_mfb_update  one workgroup per 64 x 64 tile of the front's update matrix  U = (children) - Y D^-1 Y'  (lower
//                 triangle): the children's entries gathered the same way, the rank-s product from two 64 x s slabs in LDS.
//
// Update matrices, `rel` and `loc` are the arrays of mfront.hpp (a big front's parent or child may be a small one).
#pragma once
#include "mfront.hpp"

namespace oq {
namespace {

constexpr int kMfbPanelThreads = 512;
constexpr int kMfbTile = 64;
constexpr int kMfbSmax = 64;  // pivots of a big front (the LDS block and the slabs of the update are sized for it)

struct MfbArgs {
  MfArgs a;                   // the arrays of mfront.hpp; a.list / a.count: the big fronts of this launch
  const int64_t *poff;        // panel scratch of supernode J: panel + poff[J], (s + b) rows of s doubles (rows 0..s-1 unused)
  double *panel;
  const int *tiles;           // k_mfb_update: (index into a.list, tile row, tile column) per workgroup
};

// LDS of k_mfb_panel: the pivot block (packed lower triangle of order <= 64, column-major), reciprocal pivots, one row of
// the panel per wavefront, the inverse maps (16 bits per row of the front)
__host__ __device__ inline size_t mfb_panel_lds(int fcap) {
  return sizeof(double) * (kMfbSmax * (kMfbSmax + 1) / 2 + kMfbSmax + (kMfbPanelThreads / 64) * kMfbSmax) + sizeof(uint16_t) * ((size_t)fcap + 8);
}

__global__ __launch_bounds__(kMfbPanelThreads) void k_mfb_panel(MfbArgs g) {
  extern __shared__ __attribute__((aligned(16))) double mfb_lds[];
  __shared__ int blk_pos, blk_bad;
  const MfArgs &a = g.a;
  constexpr int TF = kMfbPanelThreads, NWV = TF / 64, RW = 32, CW = TF / RW;
  const int tid = threadIdx.x, wv = tid / 64, ln = tid % 64, ta = tid % RW, tb = tid / RW;
  const int J = a.list[blockIdx.x];
  const int q0 = a.ptr[J], s = a.ptr[J + 1] - q0, b = a.bsz[J], f = s + b;
  double *F = mfb_lds;                                   // pivot block, packed lower triangle of order s
  double *dv = F + kMfbSmax * (kMfbSmax + 1) / 2;        // reciprocal pivots
  double *xrow = dv + kMfbSmax + wv * kMfbSmax;          // this wavefront's row of the panel
  uint16_t *inv = (uint16_t *)(dv + kMfbSmax + NWV * kMfbSmax);  // row of this front -> row of the child's border (0xFFFF: none)
  double *Pn = g.panel + g.poff[J];
  auto cs = [&](int j) { return j * (2 * s - j - 1) / 2; };
  if (tid == 0) { blk_pos = 0; blk_bad = 0; }
  for (int e = tid; e < s * (s + 1) / 2; e += TF) F[e] = 0.0;
  for (int64_t e = (int64_t)s * s + tid; e < (int64_t)f * s; e += TF) Pn[e] = 0.0;
  __syncthreads();
  // 1. the entries of K in the columns of the supernode: pivot block to LDS, the rest to the panel
  for (int c = tid / 16; c < s; c += TF / 16) {
    const int k = a.piv[q0 + c], cc = cs(c);
    if (tid % 16 == 0) F[cc + c] = a.D[k];
    for (int64_t t = a.Lp[k] + tid % 16; t < a.Lp[k + 1]; t += 16) {
      const int r = a.loc[t];
      if (r < s) F[cc + r] = a.Lx[t];
      else Pn[(int64_t)r * s + c] = a.Lx[t];
    }
  }
  __syncthreads();
  // 2. extend-add of the children.  Pivot block: as mfront.hpp (a target column belongs to one wavefront, whose LDS operations
  //    run in program order).  Panel: every entry (row R >= s, column C < s) belongs to one thread, which looks the two rows up
  //    in the child's inverse map.
  const int c0 = a.chp[J], c1 = a.chp[J + 1];
  for (int ci = c0; ci < c1; ci++) {
    const int cn = a.chl[ci], bc = a.bsz[cn];
    const double *Uc = a.U + a.uoff[cn];
    const uint16_t *rl = a.rel + a.reloff[cn];
    if (bc == 0) continue;
    const bool wide = rl[bc - 1] >= s;  // the child reaches rows below the pivot block (rel ascends)
    if (wide) {
      for (int R = tid; R < f; R += TF) inv[R] = 0xFFFF;
      __syncthreads();
      for (int i = tid; i < bc; i += TF) inv[rl[i]] = (uint16_t)i;
      __syncthreads();
    }
    for (int bb = 0; bb < bc; bb++) {
      const int tc = rl[bb];
      if (tc >= s) break;
      if ((tc & (NWV - 1)) != wv) continue;
      const int cb = cs(tc), ub = bb * (2 * bc - bb - 1) / 2;
      for (int r = bb + ln; r < bc; r += 64) { const int tr = rl[r]; if (tr < s) F[cb + tr] += Uc[ub + r]; }
    }
    if (wide) {
      for (int64_t e = tid; e < (int64_t)b * s; e += TF) {
        const int R = s + (int)(e / s), C = (int)(e % s);
        const unsigned i = inv[R], j = inv[C];
        if (i != 0xFFFFu && j != 0xFFFFu) Pn[(int64_t)R * s + C] += Uc[(int64_t)j * (2 * bc - j - 1) / 2 + i];
      }
      __syncthreads();  // the map is rewritten for the next child
    }
  }
  __syncthreads();
  // 3. the pivots, right-looking inside the block
  for (int p = 0; p < s; p++) {
    const int cp = cs(p);
    const double d = F[cp + p];
    const double dinv = 1.0 / d;
    if (tid == 0) { dv[p] = dinv; if ((d == 0.0) || (d != d)) blk_bad = 1; }
    for (int j = p + 1 + tb; j < s; j += CW) {
      const double w = F[cp + j] * dinv;
      const int cj = cs(j);
      for (int i = j + ta; i < s; i += RW) F[cj + i] -= F[cp + i] * w;
    }
    __syncthreads();
  }
  // 4. pivots and the block's columns of L to D / Dinv / Lx
  int pos = 0;
  for (int c = tid / 16; c < s; c += TF / 16) {
    const int k = a.piv[q0 + c], cc = cs(c);
    const double dinv = dv[c];
    if (tid % 16 == 0) { const double d = F[cc + c]; a.D[k] = d; a.Dinv[k] = dinv; pos += d > 0.0; }
    for (int64_t t = a.Lp[k] + tid % 16; t < a.Lp[k + 1]; t += 16) { const int r = a.loc[t]; if (r < s) a.Lx[t] = F[cc + r] * dinv; }
  }
  __syncthreads();
  // 5. W = L11^-1 in place (mfront.hpp step 6): B = -W below the diagonal
  for (int j = tb; j < s; j += CW) {
    const int cj = cs(j);
    const double dj = dv[j];
    for (int i = j + 1 + ta; i < s; i += RW) F[cj + i] *= dj;
  }
  __syncthreads();
  for (int p = 1; p + 1 < s; p++) {
    const int cp = cs(p);
    for (int c = tb; c < p; c += CW) {
      const int cc = cs(c);
      const double w = F[cc + p];
      for (int i = p + 1 + ta; i < s; i += RW) F[cc + i] -= F[cp + i] * w;
    }
    __syncthreads();
  }
  if (a.Wc) {
    double *Wc = a.Wc + a.woff[J], *Wr = a.Wr + a.woff[J];
    for (int j = tb; j < s; j += CW) {
      const int cj = cs(j), cw = j * (2 * s - j - 1) / 2;
      for (int i = j + ta; i < s; i += RW) {
        const double v = i == j ? 1.0 : -F[cj + i];
        Wc[cw + i] = v;
        Wr[i * (i + 1) / 2 + j] = v;
      }
    }
  }
  // 6. Y = F21 L11^-T = L21 D: row R of the panel times W', one row per wavefront at a time, lane c = column c:
  //    Y(R, c) = x(c) + sum_{k < c} W(c, k) x(k),  W(c, k) = -F[cs(k) + c]
  for (int R = s + wv; R < f; R += NWV) {
    double *row = Pn + (int64_t)R * s;
    if (ln < s) xrow[ln] = row[ln];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (ln < s) {
      double acc = xrow[ln];
      for (int k = 0; k < ln; k++) acc -= F[cs(k) + ln] * xrow[k];
      row[ln] = acc;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  }
  __threadfence_block();
  __syncthreads();
  // 7. the panel's part of the columns of L
  for (int c = tid / 16; c < s; c += TF / 16) {
    const int k = a.piv[q0 + c];
    const double dinv = dv[c];
    for (int64_t t = a.Lp[k] + tid % 16; t < a.Lp[k + 1]; t += 16) { const int r = a.loc[t]; if (r >= s) a.Lx[t] = Pn[(int64_t)r * s + c] * dinv; }
  }
  if (pos) atomicAdd(&blk_pos, pos);
  __syncthreads();
  if (tid == 0) {
    if (blk_bad) atomicOr(&a.status[0], 1);
    if (blk_pos) atomicAdd(&a.status[1], blk_pos);
  }
}

// lower bound in an ascending 16-bit list
__device__ __forceinline__ int mfb_lower(const uint16_t *v, int n, int key) {
  int l = 0, h = n;
  while (l < h) { const int mid = (l + h) >> 1; if ((int)v[mid] < key) l = mid + 1; else h = mid; }
  return l;
}

constexpr int kMfbPass = 32;  // pivots of the rank-s product staged per pass (two slabs of 32 x 64 doubles: 32 KB of LDS)

__global__ __launch_bounds__(256) void k_mfb_update(MfbArgs g) {
  __shared__ double Yr[kMfbPass][kMfbTile], Yc[kMfbPass][kMfbTile];  // [pivot][row of the tile]: a lane group reads consecutive rows
  __shared__ uint16_t invR[kMfbTile], invC[kMfbTile];
  const MfArgs &a = g.a;
  const int tid = threadIdx.x, tx = tid % 16, ty = tid / 16;
  const int li = g.tiles[3 * blockIdx.x], ti = g.tiles[3 * blockIdx.x + 1], tj = g.tiles[3 * blockIdx.x + 2];
  const int J = a.list[li];
  const int q0 = a.ptr[J], s = a.ptr[J + 1] - q0, b = a.bsz[J];
  const int r0 = ti * kMfbTile, cc0 = tj * kMfbTile;
  const double *Pn = g.panel + g.poff[J];
  double acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = 0.0;
  // children, in ascending order: the entries of their update matrices whose two rows fall into this tile
  const int c0 = a.chp[J], c1 = a.chp[J + 1];
  for (int ci = c0; ci < c1; ci++) {
    const int cn = a.chl[ci], bc = a.bsz[cn];
    if (bc == 0) continue;
    const uint16_t *rl = a.rel + a.reloff[cn];
    if ((int)rl[bc - 1] < s + r0) continue;  // nothing of the child at or below the tile's first row (uniform over the workgroup)
    const int ir0 = mfb_lower(rl, bc, s + r0), ir1 = mfb_lower(rl, bc, s + r0 + kMfbTile);
    const int ic0 = mfb_lower(rl, bc, s + cc0), ic1 = mfb_lower(rl, bc, s + cc0 + kMfbTile);
    if (ir0 == ir1 || ic0 == ic1) continue;
    __syncthreads();
    if (tid < kMfbTile) { invR[tid] = 0xFFFF; invC[tid] = 0xFFFF; }
    __syncthreads();
    for (int i = ir0 + tid; i < ir1; i += 256) invR[rl[i] - s - r0] = (uint16_t)i;
    for (int i = ic0 + tid; i < ic1; i += 256) invC[rl[i] - s - cc0] = (uint16_t)i;
    __syncthreads();
    const double *Uc = a.U + a.uoff[cn];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const unsigned ri = invR[ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const unsigned cj = invC[tx * 4 + j];
        if (ri != 0xFFFFu && cj != 0xFFFFu && ri >= cj) acc[i][j] += Uc[(int64_t)cj * (2 * bc - cj - 1) / 2 + ri];
      }
    }
  }
  // rank-s product Y D^-1 Y', kMfbPass pivots at a time: rows r0.. as they are, rows cc0.. times the reciprocal pivots
  for (int p0 = 0; p0 < s; p0 += kMfbPass) {
    const int np = s - p0 < kMfbPass ? s - p0 : kMfbPass;
    __syncthreads();
    for (int e = tid; e < kMfbTile * np; e += 256) {
      const int r = e / np, p = e % np;
      Yr[p][r] = r0 + r < b ? Pn[(int64_t)(s + r0 + r) * s + p0 + p] : 0.0;
      Yc[p][r] = cc0 + r < b ? Pn[(int64_t)(s + cc0 + r) * s + p0 + p] * a.Dinv[a.piv[q0 + p0 + p]] : 0.0;
    }
    __syncthreads();
    for (int p = 0; p < np; p++) {
      double yr[4], yc[4];
#pragma unroll
      for (int i = 0; i < 4; i++) { yr[i] = Yr[p][ty * 4 + i]; yc[i] = Yc[p][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] -= yr[i] * yc[j];
    }
  }
  double *Uj = a.U + a.uoff[J];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int R = r0 + ty * 4 + i;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int C = cc0 + tx * 4 + j;
      if (R < b && C < b && R >= C) Uj[(int64_t)C * (2 * b - C - 1) / 2 + R] = acc[i][j];
    }
  }
}

}  // namespace
}  // namespace oq
