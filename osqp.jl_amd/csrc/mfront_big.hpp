// mfront_big.hpp -- fronts that do not fit one workgroup's LDS (round 6), included by direct.hip after mfront.hpp.
//
// Up to round 5 ONE supernode whose front had more than 192 rows sent the whole matrix back to the level-by-level
// factorisation (two launches per PIVOT level): every 2-D structure does that -- a g x g grid has separators of g nodes, so
// fronts of up to ~2 g rows.  Such a front is now factorised out of global memory (it is L2-sized: 700 rows x 64 pivots
// = 0.36 MB of panel), in two launches per supernode level next to the LDS launches of the level's small fronts:
//
//   k_mfb_panel   one workgroup per front.  The s x s pivot block lives in LDS (assembled, factorised and inverted there
//                 exactly as mfront.hpp does for a small front); the rows below it -- the b x s panel F21 -- are assembled
//                 in a global scratch (column-major: a column of the panel is what the entries of L, the children's columns and the
//                 lanes of a wavefront walk) by OWNER-COMPUTES gathers from the children's update
//                 matrices (a per-child inverse map  row of this front -> row of the child  in LDS: every entry is summed
//                 by one thread, children in ascending order: a fixed order of sums, no atomics), then turned into
//                 Y = F21 L11^-T = L21 D with the inverted block, written back to the scratch, and scattered to Lx as L21.
//   k_mfb_update  one workgroup per 64 x 64 tile of the front's update matrix  U = (children) - Y D^-1 Y'  (lower
//                 triangle): the children's entries gathered the same way, the rank-s product from two 64 x s slabs in LDS.
//
// Update matrices, `rel` and `loc` are the arrays of mfront.hpp (a big front's parent or child may be a small one).
#pragma once
#include "mfront.hpp"
#include <utility>

namespace oq {
namespace {

#ifdef OQ_MFB_PROFILE  // experiment build: wall-clock stamps (100 MHz) per phase of k_mfb_panel, printed by launches of one front
#define MFB_T0 long long mt0 = wall_clock64(), macc[12] = {0};
#define MFB_T(k) { __syncthreads(); long long mt1 = wall_clock64(); macc[k] += mt1 - mt0; mt0 = mt1; }
#define MFB_PRINT if (blockIdx.x == 0 && tid == 0) printf("mfb f %d s %d children %d: zero %lld scatter %lld narrow %lld wide %lld ldl %lld lx %lld inv %lld w %lld y %lld lxp %lld (x10 ns)\n", f, s, c1 - c0, macc[0], macc[1], macc[2], macc[3], macc[4], macc[5], macc[6], macc[7], macc[8], macc[9]);
#else
#define MFB_T0
#define MFB_T(k)
#define MFB_PRINT
#endif
constexpr int kMfbPanelThreads = 512;
constexpr int kMfbTile = 64;
constexpr int kMfbSmax = 64;  // pivots of a big front (the LDS block and the slabs of the update are sized for it)
constexpr int kMfbRows = 4;   // rows of the panel a wavefront turns into rows of Y at a time
constexpr int kMfbNarrow = 8, kMfbNarrowU = kMfbNarrow * (kMfbNarrow + 1) / 2;  // children of at most this many border rows are staged in LDS

struct MfbArgs {
  MfArgs a;                   // the arrays of mfront.hpp; a.list / a.count: the big fronts of this launch
  const int64_t *poff;        // panel scratch of supernode J: panel + poff[J], s columns of (s + b) doubles (rows 0..s-1 of a column unused)
  double *panel;
  const int *tiles;           // k_mfb_update: (index into a.list, tile row, tile column) per workgroup
};

// LDS of k_mfb_panel: the pivot block (packed lower triangle of order <= 64, column-major), reciprocal pivots, one row of
// the panel per wavefront, the inverse maps (16 bits per row of the front)
__host__ __device__ inline size_t mfb_panel_lds(int fcap) {
  return sizeof(double) * (kMfbSmax * (kMfbSmax + 1) / 2 + kMfbSmax + (kMfbPanelThreads / 64) * kMfbRows * kMfbSmax + 64 * kMfbNarrowU) + sizeof(uint16_t) * ((size_t)fcap + 8 + 64 * kMfbNarrow);
}

// acc += (b of lane LN of the sixteen-lane row) * m.  volatile: the multiply-adds of a row that is only stored by some lanes must
// not be sunk into that condition -- the broadcast reads the register of ANOTHER lane, which has to be executing
template <int LN>
__device__ __forceinline__ void mfb_fmac_bc(double &acc, double b, double m) {
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(m), "n"(LN));
}

__global__ __launch_bounds__(kMfbPanelThreads) void k_mfb_panel(MfbArgs g) {
  extern __shared__ __attribute__((aligned(16))) double mfb_lds[];
  __shared__ int blk_pos, blk_bad;
  __shared__ int h_bc[64], h_first[64], h_last[64];  // headers of up to 64 children at a time (a separator supernode has one pendant
  __shared__ long long h_uoff[64], h_roff[64];       // constraint row per pivot among its children: 66 dependent trips to memory otherwise)
  __shared__ double h_val[64];
  const MfArgs &a = g.a;
  constexpr int TF = kMfbPanelThreads, NWV = TF / 64, RW = 32, CW = TF / RW;
  const int tid = threadIdx.x, wv = tid / 64, ln = tid % 64, ta = tid % RW, tb = tid / RW;
  const int J = a.list[blockIdx.x];
  const int q0 = a.ptr[J], s = a.ptr[J + 1] - q0, b = a.bsz[J], f = s + b;
  double *F = mfb_lds;                                   // pivot block, packed lower triangle of order s
  double *dv = F + kMfbSmax * (kMfbSmax + 1) / 2;        // reciprocal pivots
  double *stU = dv + kMfbSmax + NWV * kMfbRows * kMfbSmax;  // [64][kMfbNarrowU]: update matrices of the staged children
  uint16_t *stR = (uint16_t *)(stU + 64 * kMfbNarrowU);      // [64][kMfbNarrow]: their rows in this front
  uint16_t *inv = stR + 64 * kMfbNarrow;                      // row of this front -> row of the child's border (0xFFFF: none)
  double *Pn = g.panel + g.poff[J];
  auto cs = [&](int j) { return j * (2 * s - j - 1) / 2; };
  MFB_T0
  if (tid == 0) { blk_pos = 0; blk_bad = 0; }
  for (int e = tid; e < s * (s + 1) / 2; e += TF) F[e] = 0.0;
  for (int64_t e = tid; e < (int64_t)f * s; e += TF) Pn[e] = 0.0;
  __syncthreads();
  MFB_T(0)
  // 1. the entries of K in the columns of the supernode: pivot block to LDS, the rest to the panel
  for (int c = tid / 32; c < s; c += TF / 32) {
    const int k = a.piv[q0 + c], cc = cs(c);
    if (tid % 32 == 0) F[cc + c] = a.D[k];
    const int64_t t1 = a.Lp[k + 1];
    int64_t t = a.Lp[k] + tid % 32;
    for (; t + 96 < t1; t += 128) {  // four entries of the column in flight per lane
      const int r0 = a.loc[t], r1 = a.loc[t + 32], r2 = a.loc[t + 64], r3 = a.loc[t + 96];
      const double v0 = a.Lx[t], v1 = a.Lx[t + 32], v2 = a.Lx[t + 64], v3 = a.Lx[t + 96];
      if (r0 < s) F[cc + r0] = v0; else Pn[(int64_t)c * f + r0] = v0;
      if (r1 < s) F[cc + r1] = v1; else Pn[(int64_t)c * f + r1] = v1;
      if (r2 < s) F[cc + r2] = v2; else Pn[(int64_t)c * f + r2] = v2;
      if (r3 < s) F[cc + r3] = v3; else Pn[(int64_t)c * f + r3] = v3;
    }
    for (; t < t1; t += 32) {
      const int r = a.loc[t];
      if (r < s) F[cc + r] = a.Lx[t];
      else Pn[(int64_t)c * f + r] = a.Lx[t];
    }
  }
  __syncthreads();
  MFB_T(1)
  // 2. extend-add of the children.  Pivot block: as mfront.hpp (a target column belongs to one wavefront, whose LDS operations
  //    run in program order).  Panel: every entry (row R >= s, column C < s) belongs to one thread, which looks the two rows up
  //    in the child's inverse map.
  const int c0 = a.chp[J], c1 = a.chp[J + 1];
  for (int cbase = c0; cbase < c1; cbase += 64) {
    const int nc = c1 - cbase < 64 ? c1 - cbase : 64;
    __syncthreads();
    if (tid < nc) {
      const int cn = a.chl[cbase + tid], bc = a.bsz[cn];
      const long long uo = a.uoff[cn], ro = a.reloff[cn];
      h_bc[tid] = bc; h_uoff[tid] = uo; h_roff[tid] = ro;
      h_first[tid] = bc ? a.rel[ro] : 0; h_last[tid] = bc ? a.rel[ro + bc - 1] : 0;
      h_val[tid] = bc == 1 ? a.U[uo] : 0.0;
    }
    __syncthreads();
    // children of a few border rows, all of them pivots of this front (the leaf subtrees along a separator: ~100 per front, each
    // 2 - 3 dependent trips to memory when walked from global arrays): rows and update matrices staged in LDS, eight lanes a child
    {
      const int k = tid / 8, l = tid % 8;
      if (k < nc) {
        const int bc = h_bc[k];
        if (bc >= 2 && bc <= kMfbNarrow && h_last[k] < s) {
          if (l < bc) stR[k * kMfbNarrow + l] = a.rel[h_roff[k] + l];
          for (int e = l; e < bc * (bc + 1) / 2; e += 8) stU[k * kMfbNarrowU + e] = a.U[h_uoff[k] + e];
        }
      }
    }
    __syncthreads();
    // children of ONE border row (a pendant constraint row of a pivot): their single entry goes to the diagonal of their pivot,
    // summed by the pivot's thread in the order of the children
    if (tid < s) {
      double d = F[cs(tid) + tid];
      for (int k = 0; k < nc; k++) if (h_bc[k] == 1 && h_first[k] == tid) d += h_val[k];
      F[cs(tid) + tid] = d;
    }
    __syncthreads();  // (the diagonal sums)
    for (int k = 0; k < nc; k++) {  // the staged children: a target column belongs to one wavefront, children in order
      const int bc = h_bc[k];
      if (!(bc >= 2 && bc <= kMfbNarrow && h_last[k] < s)) continue;
      const uint16_t *rl = stR + k * kMfbNarrow;
      const double *Uc = stU + k * kMfbNarrowU;
      for (int bb = 0; bb < bc; bb++) {
        const int tc = rl[bb];
        if ((tc & (NWV - 1)) != wv) continue;
        const int r = bb + ln;
        if (r < bc) F[cs(tc) + rl[r]] += Uc[bb * (2 * bc - bb - 1) / 2 + r];
      }
    }
    MFB_T(2)
    for (int k = 0; k < nc; k++) {
      const int bc = h_bc[k];
      if (bc <= 1 || (bc <= kMfbNarrow && h_last[k] < s)) continue;
      const double *Uc = a.U + h_uoff[k];
      const uint16_t *rl = a.rel + h_roff[k];
      const bool wide = h_last[k] >= s;  // the child reaches rows below the pivot block (rel ascends)
      __syncthreads();                   // (the diagonal sums above, the previous child's map)
      if (wide) {
        for (int R = tid; R < f; R += TF) inv[R] = 0xFFFF;
        __syncthreads();
        for (int i = tid; i < bc; i += TF) inv[rl[i]] = (uint16_t)i;
        __syncthreads();
      }
      if (h_first[k] < s)
        for (int bb = 0; bb < bc; bb++) {
          const int tc = rl[bb];
          if (tc >= s) break;
          if ((tc & (NWV - 1)) != wv) continue;
          const int cb = cs(tc), ub = bb * (2 * bc - bb - 1) / 2;
          for (int r = bb + ln; r < bc; r += 64) { const int tr = rl[r]; if (tr < s) F[cb + tr] += Uc[ub + r]; }
        }
      if (wide) {  // a wavefront takes eight columns of the panel (their rows of the child looked up once), its lanes the rows: a row's
                   // place in the child is looked up once per row, the eight gathers and the eight sums of a row are independent
        unsigned jv[kMfbSmax / NWV];
        const double *Ucj[kMfbSmax / NWV];
#pragma unroll
        for (int qq = 0; qq < kMfbSmax / NWV; qq++) {
          const int C = wv + NWV * qq;
          jv[qq] = C < s ? inv[C] : 0xFFFFu;
          Ucj[qq] = Uc + (int64_t)jv[qq] * (2 * bc - (int64_t)jv[qq] - 1) / 2;
        }
        for (int R = s + ln; R < f; R += 64) {
          const unsigned i = inv[R];
          if (i == 0xFFFFu) continue;
          double u[kMfbSmax / NWV], o[kMfbSmax / NWV];
#pragma unroll
          for (int qq = 0; qq < kMfbSmax / NWV; qq++) {
            const bool on = jv[qq] != 0xFFFFu;
            u[qq] = on ? Ucj[qq][i] : 0.0;
            o[qq] = on ? Pn[(int64_t)(wv + NWV * qq) * f + R] : 0.0;
          }
#pragma unroll
          for (int qq = 0; qq < kMfbSmax / NWV; qq++) if (jv[qq] != 0xFFFFu) Pn[(int64_t)(wv + NWV * qq) * f + R] = o[qq] + u[qq];
        }
      }
    }
  }
  __syncthreads();
  MFB_T(3)
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
  MFB_T(4)
  // 4. pivots and the block's columns of L to D / Dinv / Lx
  int pos = 0;
  for (int c = tid / 16; c < s; c += TF / 16) {
    const int k = a.piv[q0 + c], cc = cs(c);
    const double dinv = dv[c];
    if (tid % 16 == 0) { const double d = F[cc + c]; a.D[k] = d; a.Dinv[k] = dinv; pos += d > 0.0; }
    for (int64_t t = a.Lp[k] + tid % 16; t < a.Lp[k + 1]; t += 16) { const int r = a.loc[t]; if (r < s) a.Lx[t] = F[cc + r] * dinv; }
  }
  __syncthreads();
  MFB_T(5)
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
  MFB_T(6)
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
  MFB_T(7)
  // 6. Y = F21 L11^-T = L21 D: row R of the panel times W', one row per wavefront at a time, lane c = column c:
  //    Y(R, c) = x(c) + sum_{k < c} W(c, k) x(k),  W(c, k) = -F[cs(k) + c]
  //    A lane keeps ONE row: its s sums in registers, x(k) read once per k (a coalesced read of column k of the panel), the
  //    coefficients W(c, k) broadcast from LDS.
  //    Both loops unrolled (k and c compile-time: the sums stay in registers).  The coefficients of a step -- column k of W below
  //    its diagonal -- are the same for every row: sixteen of them sit in one register across a sixteen-lane row (ONE ds_read per
  //    sixteen coefficients) and reach the multiply-add through the DPP broadcast (v_fmac_f64_dpp row_newbcast, the idiom of the
  //    batched kernel), where a ds_read per coefficient made the LDS port the bound (85 us of a 310 us front).  Columns c >= s of
  //    a narrower block are computed on whatever lies behind the block and never stored.
  {
    const int lane16 = ln & 15;
    for (int R0 = s + wv * 64; R0 < f; R0 += NWV * 64) {
      const int R = R0 + ln;
      const bool valid = R < f;
      double acc[kMfbSmax];
#pragma unroll
      for (int c = 0; c < kMfbSmax; c++) acc[c] = 0.0;
      [&]<int... KS>(std::integer_sequence<int, KS...>) {
        ([&] {
          constexpr int K = KS;
          if (K < s) {
            const double xk = valid ? Pn[(int64_t)K * f + R] : 0.0;
            acc[K] += xk;
            constexpr int NBK = (kMfbSmax - 1 - K + 15) / 16;
            double Bk[NBK > 0 ? NBK : 1];
            const double *Fk = F + cs(K) + K + 1 + lane16;
#pragma unroll
            for (int mm = 0; mm < NBK; mm++) Bk[mm] = -Fk[16 * mm];
            [&]<int... CS>(std::integer_sequence<int, CS...>) { (mfb_fmac_bc<CS & 15>(acc[K + 1 + CS], Bk[CS >> 4], xk), ...); }(std::make_integer_sequence<int, kMfbSmax - 1 - K>{});
          }
        }(), ...);
      }(std::make_integer_sequence<int, kMfbSmax>{});
      if (valid) [&]<int... CS>(std::integer_sequence<int, CS...>) { ((CS < s ? (void)(Pn[(int64_t)CS * f + R] = acc[CS]) : (void)0), ...); }(std::make_integer_sequence<int, kMfbSmax>{});
    }
  }
  __threadfence_block();
  __syncthreads();
  MFB_T(8)
  // 7. the panel's part of the columns of L
  for (int c = tid / 32; c < s; c += TF / 32) {
    const int k = a.piv[q0 + c];
    const double dinv = dv[c];
    const int64_t t1 = a.Lp[k + 1];
    int64_t t = a.Lp[k] + tid % 32;
    for (; t + 96 < t1; t += 128) {
      const int r0 = a.loc[t], r1 = a.loc[t + 32], r2 = a.loc[t + 64], r3 = a.loc[t + 96];
      const double v0 = r0 >= s ? Pn[(int64_t)c * f + r0] : 0.0, v1 = r1 >= s ? Pn[(int64_t)c * f + r1] : 0.0;
      const double v2 = r2 >= s ? Pn[(int64_t)c * f + r2] : 0.0, v3 = r3 >= s ? Pn[(int64_t)c * f + r3] : 0.0;
      if (r0 >= s) a.Lx[t] = v0 * dinv;
      if (r1 >= s) a.Lx[t + 32] = v1 * dinv;
      if (r2 >= s) a.Lx[t + 64] = v2 * dinv;
      if (r3 >= s) a.Lx[t + 96] = v3 * dinv;
    }
    for (; t < t1; t += 32) { const int r = a.loc[t]; if (r >= s) a.Lx[t] = Pn[(int64_t)c * f + r] * dinv; }
  }
  MFB_T(9)
  MFB_PRINT
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

// ---- the same front in TWO launches (round 6, second pass) ---------------------------------------------------------------
// One workgroup per front is one compute unit per front: the phases that scale with the front's rows -- zero, scatter,
// the wide children's gathers, Y, the panel's columns of L: 190 of the 270 us of a 659-row front -- run on one of 256 units
// while a launch of the top levels holds nine to forty fronts.  Only the 64 x 64 pivot block is serial.  So:
//   k_mfb_pivot  one workgroup per front: the pivot block alone (assembly, LDL', inverse), W = L11^-1 left in the unused
//                block rows of the front's panel scratch (rows < s of its s columns), the children that reach below the block
//                listed for the second launch;
//   k_mfb_rows   one workgroup per 64 ROWS of a front's panel: the piece X (64 rows x s columns) assembled in LDS -- entries of
//                K found by bisection in the columns of L (rows ascend within a column), the listed children gathered through
//                per-child inverse maps of the piece, one thread per entry, children in ascending order --, Y = X W' with the
//                four wavefronts taking every fourth column (coefficients broadcast from LDS, plain fused multiply-adds: the DPP
//                broadcast of the one-launch form runs at a quarter of that rate for 64-bit operands), Y to the panel scratch
//                and Y D^-1 to the columns of L.
// OSQP_AMD_MFB_SPLIT=0 keeps the one-launch form (A/B, tests).
struct MfbSplitArgs {
  MfbArgs g;
  int *wide;                  // per front: the children that reach below the pivot block, at wide + chp[J] (ascending)
  int *nwide;                 // ... how many
  const int *chunks;          // k_mfb_rows: (index into a.list, 64-row piece) per workgroup
  int64_t *ct0;               // per piece and column c < 64: where the rows of the piece begin in column c of L (set once: k_mfb_chunk_t0)
};

__global__ __launch_bounds__(kMfbPanelThreads) void k_mfb_pivot(MfbSplitArgs sa) {
  extern __shared__ __attribute__((aligned(16))) double mfb_lds[];
  __shared__ int blk_pos, blk_bad, n_wide;
  __shared__ int h_bc[64], h_first[64], h_last[64];
  __shared__ long long h_uoff[64], h_roff[64];
  __shared__ double h_val[64];
  const MfbArgs &g = sa.g;
  const MfArgs &a = g.a;
  constexpr int TF = kMfbPanelThreads, NWV = TF / 64, RW = 32, CW = TF / RW;
  const int tid = threadIdx.x, wv = tid / 64, ln = tid % 64, ta = tid % RW, tb = tid / RW;
  const int J = a.list[blockIdx.x];
  const int q0 = a.ptr[J], s = a.ptr[J + 1] - q0, b = a.bsz[J], f = s + b;
  double *F = mfb_lds;                                   // pivot block, packed lower triangle of order s
  double *dv = F + kMfbSmax * (kMfbSmax + 1) / 2;        // reciprocal pivots
  double *stU = dv + kMfbSmax + NWV * kMfbRows * kMfbSmax;
  uint16_t *stR = (uint16_t *)(stU + 64 * kMfbNarrowU);
  double *Pn = g.panel + g.poff[J];
  auto cs = [&](int j) { return j * (2 * s - j - 1) / 2; };
  if (tid == 0) { blk_pos = 0; blk_bad = 0; n_wide = 0; }
  for (int e = tid; e < s * (s + 1) / 2; e += TF) F[e] = 0.0;
  __syncthreads();
  // 1. the entries of K inside the pivot block (the rows of a column ascend: the block's come first)
  for (int c = tid / 32; c < s; c += TF / 32) {
    const int k = a.piv[q0 + c], cc = cs(c);
    if (tid % 32 == 0) F[cc + c] = a.D[k];
    const int64_t t1 = a.Lp[k + 1];
    for (int64_t t = a.Lp[k] + tid % 32; t < t1; t += 32) {
      const int r = a.loc[t];
      if (r >= s) break;
      F[cc + r] = a.Lx[t];
    }
  }
  __syncthreads();
  // 2. extend-add of the children into the pivot block (as k_mfb_panel), the children that reach below it listed
  const int c0 = a.chp[J], c1 = a.chp[J + 1];
  for (int cbase = c0; cbase < c1; cbase += 64) {
    const int nc = c1 - cbase < 64 ? c1 - cbase : 64;
    __syncthreads();
    if (tid < nc) {
      const int cn = a.chl[cbase + tid], bc = a.bsz[cn];
      const long long uo = a.uoff[cn], ro = a.reloff[cn];
      h_bc[tid] = bc; h_uoff[tid] = uo; h_roff[tid] = ro;
      h_first[tid] = bc ? a.rel[ro] : 0; h_last[tid] = bc ? a.rel[ro + bc - 1] : 0;
      h_val[tid] = bc == 1 ? a.U[uo] : 0.0;
    }
    __syncthreads();
    if (tid == 0) {
      int nw = n_wide;
      for (int k = 0; k < nc; k++)
        if (h_bc[k] >= 2 && h_first[k] < s && h_last[k] >= s) sa.wide[c0 + nw++] = a.chl[cbase + k];
      n_wide = nw;
    }
    {
      const int k = tid / 8, l = tid % 8;
      if (k < nc) {
        const int bc = h_bc[k];
        if (bc >= 2 && bc <= kMfbNarrow && h_last[k] < s) {
          if (l < bc) stR[k * kMfbNarrow + l] = a.rel[h_roff[k] + l];
          for (int e = l; e < bc * (bc + 1) / 2; e += 8) stU[k * kMfbNarrowU + e] = a.U[h_uoff[k] + e];
        }
      }
    }
    __syncthreads();
    if (tid < s) {
      double d = F[cs(tid) + tid];
      for (int k = 0; k < nc; k++) if (h_bc[k] == 1 && h_first[k] == tid) d += h_val[k];
      F[cs(tid) + tid] = d;
    }
    __syncthreads();
    for (int k = 0; k < nc; k++) {
      const int bc = h_bc[k];
      if (!(bc >= 2 && bc <= kMfbNarrow && h_last[k] < s)) continue;
      const uint16_t *rl = stR + k * kMfbNarrow;
      const double *Uc = stU + k * kMfbNarrowU;
      for (int bb = 0; bb < bc; bb++) {
        const int tc = rl[bb];
        if ((tc & (NWV - 1)) != wv) continue;
        const int r = bb + ln;
        if (r < bc) F[cs(tc) + rl[r]] += Uc[bb * (2 * bc - bb - 1) / 2 + r];
      }
    }
    for (int k = 0; k < nc; k++) {
      const int bc = h_bc[k];
      if (bc <= 1 || (bc <= kMfbNarrow && h_last[k] < s) || h_first[k] >= s) continue;
      const double *Uc = a.U + h_uoff[k];
      const uint16_t *rl = a.rel + h_roff[k];
      for (int bb = 0; bb < bc; bb++) {
        const int tc = rl[bb];
        if (tc >= s) break;
        if ((tc & (NWV - 1)) != wv) continue;
        const int cb = cs(tc), ub = bb * (2 * bc - bb - 1) / 2;
        for (int r = bb + ln; r < bc; r += 64) { const int tr = rl[r]; if (tr < s) F[cb + tr] += Uc[ub + r]; }
      }
    }
  }
  __syncthreads();
  if (tid == 0) sa.nwide[J] = n_wide;
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
    for (int64_t t = a.Lp[k] + tid % 16; t < a.Lp[k + 1]; t += 16) { const int r = a.loc[t]; if (r >= s) break; a.Lx[t] = F[cc + r] * dinv; }
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
  // 6. W to the arrays of the solves, and -- whole, zeros above the diagonal -- to the block rows of the panel scratch for k_mfb_rows
  for (int j = tb; j < s; j += CW) {
    const int cj = cs(j), cw = j * (2 * s - j - 1) / 2;
    for (int i = ta; i < s; i += RW) {
      const double v = i == j ? 1.0 : (i > j ? -F[cj + i] : 0.0);
      Pn[(int64_t)j * f + i] = v;
      if (a.Wc && i >= j) { a.Wc[a.woff[J] + cw + i] = v; a.Wr[a.woff[J] + i * (i + 1) / 2 + j] = v; }
    }
  }
  if (pos) atomicAdd(&blk_pos, pos);
  __syncthreads();
  if (tid == 0) {
    if (blk_bad) atomicOr(&a.status[0], 1);
    if (blk_pos) atomicAdd(&a.status[1], blk_pos);
  }
}

// first position in [t0, t1) whose 16-bit value is >= key (ascending list)
__device__ __forceinline__ int64_t mfb_lower64(const uint16_t *v, int64_t t0, int64_t t1, int key) {
  while (t0 < t1) { const int64_t mid = (t0 + t1) >> 1; if ((int)v[mid] < key) t0 = mid + 1; else t1 = mid; }
  return t0;
}

// setup: thread per (piece, column): first entry of column c of L whose row of the front is >= the piece's first row (a bisection
// of ~11 dependent loads -- done inside k_mfb_rows, sixteen columns per wavefront one after the other, it was most of its 100 us)
__global__ __launch_bounds__(256) void k_mfb_chunk_t0(int nchunks, MfbSplitArgs sa) {
  const MfArgs &a = sa.g.a;
  const int i = blockIdx.x * 256 + threadIdx.x, w = i / kMfbSmax, c = i % kMfbSmax;
  if (w >= nchunks) return;
  const int J = a.list[sa.chunks[2 * w]], ch = sa.chunks[2 * w + 1];
  const int q0 = a.ptr[J], s = a.ptr[J + 1] - q0;
  if (c >= s) { sa.ct0[i] = 0; return; }
  const int k = a.piv[q0 + c];
  sa.ct0[i] = mfb_lower64(a.loc, a.Lp[k], a.Lp[k + 1], s + ch * 64);
}

__global__ __launch_bounds__(256) void k_mfb_rows(MfbSplitArgs sa) {
  extern __shared__ __attribute__((aligned(16))) double mfb_rows_lds[];  // 2 x 32 KB (above the static limit together with the maps)
  double *X = mfb_rows_lds;                  // the piece: X[c * 64 + r], row r of the piece, column c of the panel
  double *Wl = mfb_rows_lds + kMfbSmax * 64; // W(c, k) at Wl[k * 64 + c] (zeros above the diagonal, rows / columns beyond s zero)
  __shared__ double dinv[kMfbSmax];
  __shared__ long long tlo[kMfbSmax], thi[kMfbSmax];
  __shared__ unsigned short invR[64], invC[kMfbSmax];
  __shared__ int bnd[3];
  const MfbArgs &g = sa.g;
  const MfArgs &a = g.a;
  const int tid = threadIdx.x, wv = tid / 64, ln = tid % 64;
  const int li = sa.chunks[2 * blockIdx.x], ch = sa.chunks[2 * blockIdx.x + 1];
  const int J = a.list[li];
  const int q0 = a.ptr[J], s = a.ptr[J + 1] - q0, b = a.bsz[J], f = s + b;
  const int R0 = s + ch * 64, nr = f - R0 < 64 ? f - R0 : 64;
  double *Pn = g.panel + g.poff[J];
  for (int e = tid; e < kMfbSmax * 64; e += 256) {
    const int k = e / 64, c = e % 64;
    X[e] = 0.0;
    Wl[e] = (k < s && c < s) ? Pn[(int64_t)k * f + c] : 0.0;
  }
  if (tid < kMfbSmax) dinv[tid] = tid < s ? a.Dinv[a.piv[q0 + tid]] : 0.0;
  __syncthreads();
  // 1. the entries of K in the piece: in column c of L the entries of the rows R0 .. R0 + 64 are one run (the rows ascend)
  if (tid < s) {  // (the pieces of a front follow each other in the list: the next piece's start is this one's end)
    const bool last = R0 + 64 >= f;
    tlo[tid] = sa.ct0[(int64_t)blockIdx.x * kMfbSmax + tid];
    thi[tid] = last ? a.Lp[a.piv[q0 + tid] + 1] : sa.ct0[((int64_t)blockIdx.x + 1) * kMfbSmax + tid];
  }
  __syncthreads();
  for (int c = wv; c < s; c += 4)
    for (int64_t t = tlo[c] + ln; t < thi[c]; t += 64) X[c * 64 + (a.loc[t] - R0)] = a.Lx[t];
  // 2. the children that reach below the pivot block, ascending: entry (R, C) of the piece is U_child(row of R, row of C)
  const int c0 = a.chp[J], nw = sa.nwide[J];
  for (int w = 0; w < nw; w++) {
    const int cn = sa.wide[c0 + w], bc = a.bsz[cn];
    const uint16_t *rl = a.rel + a.reloff[cn];
    if ((int)rl[bc - 1] < R0) continue;  // (uniform over the workgroup)
    __syncthreads();                     // (the previous child's maps and bounds are no longer read)
    if (tid < 3) bnd[tid] = mfb_lower(rl, bc, tid == 0 ? R0 : (tid == 1 ? R0 + 64 : s));  // three bisections side by side
    if (tid < 64) invR[tid] = 0xFFFF;
    if (tid < kMfbSmax) invC[tid] = 0xFFFF;
    __syncthreads();
    const int i0 = bnd[0], i1 = bnd[1], ncol = bnd[2];
    if (i0 == i1) continue;
    for (int i = i0 + tid; i < i1; i += 256) invR[rl[i] - R0] = (unsigned short)i;
    for (int i = tid; i < ncol; i += 256) invC[rl[i]] = (unsigned short)i;
    __syncthreads();
    const double *Uc = a.U + a.uoff[cn];
    const unsigned ir = invR[ln];
    if (ir != 0xFFFFu)
      for (int C = wv; C < s; C += 4) {
        const unsigned ic = invC[C];
        if (ic != 0xFFFFu) X[C * 64 + ln] += Uc[(int64_t)ic * (2 * bc - (int64_t)ic - 1) / 2 + ir];
      }
  }
  __syncthreads();
  // 3. Y = X W': lane = row of the piece, the wavefront's columns c = wv + 4 j; the sums over k ascending
  double acc[kMfbSmax / 4];
#pragma unroll
  for (int j = 0; j < kMfbSmax / 4; j++) acc[j] = 0.0;
  for (int k = 0; k < s; k++) {
    const double xk = X[k * 64 + ln];
    const double *wk = Wl + k * 64 + wv;
#pragma unroll
    for (int j = 0; j < kMfbSmax / 4; j++) acc[j] = __builtin_fma(wk[4 * j], xk, acc[j]);
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kMfbSmax / 4; j++) X[(wv + 4 * j) * 64 + ln] = acc[j];
  __syncthreads();
  // 4. Y to the panel scratch (what k_mfb_update and the solves' top part read), Y D^-1 to the columns of L
  for (int e = tid; e < s * 64; e += 256) {
    const int c = e / 64, r = e % 64;
    if (r < nr) Pn[(int64_t)c * f + R0 + r] = X[e];
  }
  for (int c = wv; c < s; c += 4) {
    const double di = dinv[c];
    for (int64_t t = tlo[c] + ln; t < thi[c]; t += 64) a.Lx[t] = X[c * 64 + (a.loc[t] - R0)] * di;
  }
}


constexpr int kMfbPass = 32;  // pivots of the rank-s product staged per pass (two slabs of 32 x 64 doubles: 32 KB of LDS)

__global__ __launch_bounds__(256) void k_mfb_update(MfbArgs g) {
  __shared__ double Yr[kMfbPass][kMfbTile], Yc[kMfbPass][kMfbTile];  // [pivot][row of the tile]: a lane group reads consecutive rows
  __shared__ uint16_t invR[kMfbTile], invC[kMfbTile];
  const MfArgs &a = g.a;
  const int tid = threadIdx.x, tx = tid % 16, ty = tid / 16;
  const int li = g.tiles[3 * blockIdx.x], ti = g.tiles[3 * blockIdx.x + 1], tj = g.tiles[3 * blockIdx.x + 2];
  const int J = a.list[li];
  const int q0 = a.ptr[J], s = a.ptr[J + 1] - q0, b = a.bsz[J], f = s + b;
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
      const int p = e / kMfbTile, r = e % kMfbTile;  // (lanes along the rows: a column of the panel is contiguous)
      Yr[p][r] = r0 + r < b ? Pn[(int64_t)(p0 + p) * f + s + r0 + r] : 0.0;
      Yc[p][r] = cc0 + r < b ? Pn[(int64_t)(p0 + p) * f + s + cc0 + r] * a.Dinv[a.piv[q0 + p0 + p]] : 0.0;
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
