// mfront.hpp -- numeric LDL' by supernodes (row K2 of SURVEY.md section 8a), included by direct.hip.
//
// The level-by-level factorisation (k_ldl_diag_* / k_ldl_entries_*) needs two launches per PIVOT level and every entry of
// a level pays the latency of a sparse row intersection: control-1e6 has 519 levels, 84 ms per refactorisation = 85 ADMM
// iterations of time, and every rho update / matrix update pays it [REF src/interface.jl:330-406, 539-550].
// The supernode partition of the triangular solves (symbolic.hpp, Supernodes) supports a MULTIFRONTAL factorisation with
// one launch per SUPERNODE level and size class (11 levels there).  A supernode J is a connected piece of the elimination
// tree with a single top node, so every row its columns touch outside J lies in the pattern of the top node's column
// (col struct(v) \ {parent} is a subset of struct(parent), by induction along the piece): its FRONT is the dense symmetric
// matrix over   rows(J) = [ the s pivots of J in slot order | the b rows of the top column, ascending ]   (f = s + b).
//   1. F = entries of K in the columns of J (they sit in Lx / D after the scatter kernels of the assembly);
//   2. F += the update matrices of the children of J (extend-add, children in ascending order: a fixed order of sums --
//      no floating-point atomics, two factorisations of the same data give the same bits);
//   3. s pivots eliminated inside LDS (right-looking over the f x s panel, then the b x b Schur complement as ONE rank-s
//      product with the sums in registers);
//   4. columns of L back to Lx (sparse positions through loc[]), pivots to D / Dinv, the Schur complement = update matrix
//      U_J to a global array for the parent.
// The front lives in LDS as a packed lower triangle, column-major: entry (i, j), i >= j, at j (2f - j - 1) / 2 + i.
// Three size classes: fronts of at most 16 rows take 16 lanes (sixteen to a workgroup, no workgroup barrier), at most 48
// rows a wavefront, larger ones a 256-thread workgroup with the slab sized by the largest front of the launch.
#pragma once
#include "engine.hpp"

namespace oq {
namespace {

struct MfArgs {
  const int *list;            // supernodes of this launch
  int count;
  int fcap;                   // every front of the launch has at most this many rows
  const int *ptr, *piv;       // supernode -> slots, slot -> pivot
  const int *bsz;             // rows of the border (pattern of the top node's column)
  const int64_t *uoff;        // update matrix of supernode J: U + uoff[J], packed lower triangle of order bsz[J]
  const int64_t *reloff;      // rel + reloff[J]: row of the PARENT's front for each border row of J
  const uint16_t *rel;
  const int *chp, *chl;       // children lists
  const int64_t *Lp;
  const uint16_t *loc;        // per entry of L: row of its column's front
  double *Lx, *D, *Dinv, *U;
  int *status;                // [0] |= 1: zero / NaN pivot; [1] += positive pivots
  // the inverse of the supernode's own unit lower triangular block, for the supernodal solves (direct.hip k_sn_*): packed
  // by columns (Wc) and by rows (Wr) at woff[J]; null: not wanted
  double *Wc, *Wr;
  const int64_t *woff;
};

template <int TF>
__device__ __forceinline__ void mf_sync() {
  if constexpr (TF > 64) __syncthreads();
  else {  // the lanes of a front share a wavefront: its LDS writes are drained before any of its lanes reads them
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  }
}

#ifdef OQ_MF_PROFILE  // experiment build: wall-clock stamps (100 MHz) per phase of k_mf_front, printed by the first front of a launch
#define MF_T0 long long mt0 = wall_clock64(), macc[8] = {0};
#define MF_T(k) { __syncthreads(); long long mt1 = wall_clock64(); macc[k] += mt1 - mt0; mt0 = mt1; }
#define MF_PRINT if (blockIdx.x == 0 && threadIdx.x == 0) printf("mf TF %d count %d f %d s %d children %d: zero+K %lld extend %lld pivots %lld U %lld L %lld W %lld Wout %lld (x10 ns)\n", TF, a.count, f, s, a.chp[J + 1] - a.chp[J], macc[0], macc[1], macc[2], macc[3], macc[4], macc[5], macc[6]);
#else
#define MF_T0
#define MF_T(k)
#define MF_PRINT
#endif
constexpr int kMfBlock = 256;
__host__ __device__ inline size_t mf_slab_doubles(int fcap) { return (size_t)fcap * (fcap + 1) / 2 + (size_t)(fcap < 64 ? fcap : 64); }

template <int TF>
__global__ __launch_bounds__(TF > kMfBlock ? TF : kMfBlock) void k_mf_front(MfArgs a) {
  extern __shared__ __attribute__((aligned(16))) double mf_lds[];
  __shared__ int blk_pos;
  constexpr int NFB = TF > kMfBlock ? 1 : kMfBlock / TF;  // fronts per workgroup (TF = 1024: a level of few large fronts, where
                                                          // the latency of ONE front is the level's time)
  constexpr int RW = TF >= 256 ? 32 : 16;    // lanes along the rows of a tile of work, CW along its columns
  constexpr int CW = TF / RW;
  constexpr int G = TF >= 64 ? 16 : 1;       // lanes per column when the columns of L are read / written
  const int fr = threadIdx.x / TF, tid = threadIdx.x % TF;
  const int ta = tid % RW, tb = tid / RW;
  const int li = blockIdx.x * NFB + fr;
  const bool live = li < a.count;
  const int J = live ? a.list[li] : 0;
  const int q0 = live ? a.ptr[J] : 0, s = live ? a.ptr[J + 1] - q0 : 0, b = live ? a.bsz[J] : 0, f = s + b;
  double *F = mf_lds + (size_t)fr * mf_slab_doubles(a.fcap);
  double *dv = F + (size_t)a.fcap * (a.fcap + 1) / 2;  // reciprocal pivots
  auto cs = [&](int j) { return j * (2 * f - j - 1) / 2; };
  MF_T0
  if (threadIdx.x == 0) blk_pos = 0;
  for (int e = tid; e < f * (f + 1) / 2; e += TF) F[e] = 0.0;
  mf_sync<TF>();
  if (TF <= 64) __syncthreads();  // blk_pos
  // 1. the entries of K in the columns of the supernode
  for (int c = tid / G; c < s; c += TF / G) {
    const int k = a.piv[q0 + c], cc = cs(c);
    if (tid % G == 0) F[cc + c] = a.D[k];
    for (int64_t t = a.Lp[k] + tid % G; t < a.Lp[k + 1]; t += G) F[cc + a.loc[t]] = a.Lx[t];
  }
  mf_sync<TF>();
  MF_T(0)
  // 2. extend-add of the children's update matrices.  No barrier between children: a target column belongs to ONE wavefront
  //    (column mod 4 of the front for the workgroup form; the only wavefront otherwise), and the LDS operations of a wavefront
  //    run in program order, so every entry of the front receives its children's terms in ascending order of the children --
  //    a fixed order of sums without the child-after-child barriers that made a front with 58 children ~100 us of latency.
  if constexpr (TF >= 256) {
    // (round 6) A front of a problem with a constraint row per variable has one ONE-ROW child per pivot -- 58 children for the 84-row
    // fronts of the control family, 70 for a 119-row front of a grid -- and walking them one after the other was four dependent trips
    // to memory each: 59 of the 183 us of such a front, as much as its pivots.  As in k_mfb_panel: the headers of up to 64 children
    // at a time come to LDS in one round, a one-row child is one number added to the diagonal entry of its row (the row's thread adds
    // them up in the order of the children), the others follow with their headers at hand.  (Fixed order of sums: the one-row
    // children of a batch first, then the others, ascending.)
    __shared__ int h_bc[64], h_first[64];
    __shared__ long long h_uoff[64], h_roff[64];
    __shared__ double h_val[64];
    constexpr int NWV = TF / 64;
    const int wvf = tid / 64, ln = tid % 64;
    const int c0 = a.chp[J], c1 = a.chp[J + 1];
    for (int cbase = c0; cbase < c1; cbase += 64) {
      const int nc = c1 - cbase < 64 ? c1 - cbase : 64;
      __syncthreads();
      if (tid < nc) {
        const int cn = a.chl[cbase + tid], bc = a.bsz[cn];
        const long long uo = a.uoff[cn], ro = a.reloff[cn];
        h_bc[tid] = bc; h_uoff[tid] = uo; h_roff[tid] = ro;
        h_first[tid] = bc ? a.rel[ro] : 0;
        h_val[tid] = bc == 1 ? a.U[uo] : 0.0;
      }
      __syncthreads();
      if (tid < f) {
        double d = F[cs(tid) + tid];
        for (int k = 0; k < nc; k++) if (h_bc[k] == 1 && h_first[k] == tid) d += h_val[k];
        F[cs(tid) + tid] = d;
      }
      __syncthreads();
      for (int k = 0; k < nc; k++) {
        const int bc = h_bc[k];
        if (bc <= 1) continue;
        const double *Uc = a.U + h_uoff[k];
        const uint16_t *rl = a.rel + h_roff[k];
        for (int bb = 0; bb < bc; bb++) {
          const int tc = rl[bb];
          if ((tc & (NWV - 1)) != wvf) continue;
          const int cb = cs(tc), ub = bb * (2 * bc - bb - 1) / 2;
          for (int r = bb + ln; r < bc; r += 64) F[cb + rl[r]] += Uc[ub + r];
        }
      }
    }
    mf_sync<TF>();
  } else {
    constexpr int NWV = TF >= 64 ? TF / 64 : 1;                 // wavefronts of the front
    constexpr int LW = TF >= 64 ? 64 : TF;                      // lanes that share the rows of one child column
    const int wvf = TF >= 64 ? tid / 64 : 0, ln = tid % LW;
    const int c0 = live ? a.chp[J] : 0, c1 = live ? a.chp[J + 1] : 0;
    int cn = c0 < c1 ? a.chl[c0] : 0;
    int bn = c0 < c1 ? a.bsz[cn] : 0;
    int64_t un = c0 < c1 ? a.uoff[cn] : 0, rn = c0 < c1 ? a.reloff[cn] : 0;
    for (int ci = c0; ci < c1; ci++) {
      const int bc = bn;
      const double *Uc = a.U + un;
      const uint16_t *rl = a.rel + rn;
      if (ci + 1 < c1) { cn = a.chl[ci + 1]; bn = a.bsz[cn]; un = a.uoff[cn]; rn = a.reloff[cn]; }  // the next child's header rides along
      for (int bb = 0; bb < bc; bb++) {
        const int tc = rl[bb];
        if (NWV > 1 && (tc & (NWV - 1)) != wvf) continue;
        const int cb = cs(tc), ub = bb * (2 * bc - bb - 1) / 2;
        for (int r = bb + ln; r < bc; r += LW) F[cb + rl[r]] += Uc[ub + r];
      }
    }
    mf_sync<TF>();
  }
  MF_T(1)
  // 3. the pivots of the supernode: right-looking over the columns of the panel
  int bad = 0;
  for (int p = 0; p < s; p++) {
    const int cp = cs(p);
    const double d = F[cp + p];
    const double dinv = 1.0 / d;
    if (tid == 0) { dv[p] = dinv; bad |= (d == 0.0) || (d != d); }
    for (int j = p + 1 + tb; j < s; j += CW) {
      const double w = F[cp + j] * dinv;
      const int cj = cs(j);
      int i = j + ta;
      // four rows of the strip at a time, every LDS read issued before the first write (one read-modify-write per loop step
      // is one LDS round trip of latency per step: the pivot loop of a 96-row front was ~2 us per pivot)
      for (; i + 3 * RW < f; i += 4 * RW) {
        const double a0 = F[cp + i], a1 = F[cp + i + RW], a2 = F[cp + i + 2 * RW], a3 = F[cp + i + 3 * RW];
        const double c0 = F[cj + i], c1 = F[cj + i + RW], c2 = F[cj + i + 2 * RW], c3 = F[cj + i + 3 * RW];
        F[cj + i] = c0 - a0 * w; F[cj + i + RW] = c1 - a1 * w; F[cj + i + 2 * RW] = c2 - a2 * w; F[cj + i + 3 * RW] = c3 - a3 * w;
      }
      if (i + RW < f) {
        const double a0 = F[cp + i], a1 = F[cp + i + RW], c0 = F[cj + i], c1 = F[cj + i + RW];
        F[cj + i] = c0 - a0 * w; F[cj + i + RW] = c1 - a1 * w;
        i += 2 * RW;
      }
      if (i < f) F[cj + i] -= F[cp + i] * w;
    }
    mf_sync<TF>();
  }
  MF_T(2)
  // 4. update matrix: U(r, c) = F(s + r, s + c) - sum_p F(s + r, p) dinv_p F(s + c, p), sums in registers
  if (b > 0) {
    double *Uj = a.U + a.uoff[J];
    for (int c = tb; c < b; c += CW) {
      const int cc = cs(s + c), uc = c * (2 * b - c - 1) / 2;
      for (int r = c + ta; r < b; r += RW) {
        double acc = F[cc + s + r];
        int p = 0;
        for (; p + 3 < s; p += 4) {  // the twelve reads of four pivots in flight together, the sum in the order of the plain loop
          const int c0 = cs(p), c1 = cs(p + 1), c2 = cs(p + 2), c3 = cs(p + 3);
          const double x0 = F[c0 + s + r], x1 = F[c1 + s + r], x2 = F[c2 + s + r], x3 = F[c3 + s + r];
          const double y0 = F[c0 + s + c] * dv[p], y1 = F[c1 + s + c] * dv[p + 1], y2 = F[c2 + s + c] * dv[p + 2], y3 = F[c3 + s + c] * dv[p + 3];
          acc -= x0 * y0; acc -= x1 * y1; acc -= x2 * y2; acc -= x3 * y3;
        }
        for (; p < s; p++) { const int cp = cs(p); acc -= F[cp + s + r] * (F[cp + s + c] * dv[p]); }
        Uj[uc + r] = acc;
      }
    }
  }
  MF_T(3)
  // 5. the columns of L, the pivots, the inertia
  int pos = 0;
  for (int c = tid / G; c < s; c += TF / G) {
    const int k = a.piv[q0 + c], cc = cs(c);
    const double dinv = dv[c];
    if (tid % G == 0) { const double d = F[cc + c]; a.D[k] = d; a.Dinv[k] = dinv; pos += d > 0.0; }
    for (int64_t t = a.Lp[k] + tid % G; t < a.Lp[k + 1]; t += G) a.Lx[t] = F[cc + a.loc[t]] * dinv;
  }
  // 6. W = L_JJ^-1 in place.  With B = -W below the diagonal, eliminating column p of the unit lower triangular block from
  //    the rows below it is   B(i, c) -= B(i, p) B(p, c),  c < p < i   (B(i, p) is still the entry of L when step p reads it:
  //    it only becomes a target in later steps) -- every update of a step is independent, one barrier per step.
  MF_T(4)
  if (a.Wc) {
    mf_sync<TF>();
    for (int j = tb; j < s; j += CW) {
      const int cj = cs(j);
      const double dj = dv[j];
      for (int i = j + 1 + ta; i < s; i += RW) F[cj + i] *= dj;
    }
    mf_sync<TF>();
    for (int p = 1; p + 1 < s; p++) {
      const int cp = cs(p);
      for (int c = tb; c < p; c += CW) {
        const int cc = cs(c);
        const double w = F[cc + p];
        int i = p + 1 + ta;
        if (i + RW < s) {  // (s <= 64: at most two rows per lane with 32 lanes along the rows, four with 16)
          const double a0 = F[cp + i], a1 = F[cp + i + RW], c0 = F[cc + i], c1 = F[cc + i + RW];
          F[cc + i] = c0 - a0 * w; F[cc + i + RW] = c1 - a1 * w;
          i += 2 * RW;
        }
        for (; i < s; i += RW) F[cc + i] -= F[cp + i] * w;
      }
      mf_sync<TF>();
    }
    MF_T(5)
    double *Wc = a.Wc + (live ? a.woff[J] : 0), *Wr = a.Wr + (live ? a.woff[J] : 0);
    for (int j = tb; j < s; j += CW) {
      const int cj = cs(j), cw = j * (2 * s - j - 1) / 2;
      for (int i = j + ta; i < s; i += RW) {
        const double v = i == j ? 1.0 : -F[cj + i];
        Wc[cw + i] = v;
        Wr[i * (i + 1) / 2 + j] = v;
      }
    }
  }
  MF_T(6)
  MF_PRINT
  if (bad) atomicOr(&a.status[0], 1);
  if (pos) atomicAdd(&blk_pos, pos);
  __syncthreads();
  if (threadIdx.x == 0 && blk_pos) atomicAdd(&a.status[1], blk_pos);
}

// setup: row of the PARENT's front for every border row of every supernode (one thread per supernode)
__global__ __launch_bounds__(kBlock) void k_mf_rel(int count, const int *__restrict__ ptr, const int *__restrict__ piv,
                                                   const int *__restrict__ up, const int *__restrict__ snof, const int *__restrict__ slot,
                                                   const int64_t *__restrict__ Lp, const int *__restrict__ Li,
                                                   const int64_t *__restrict__ reloff, uint16_t *__restrict__ rel, int *__restrict__ err) {
  const int J = blockIdx.x * kBlock + threadIdx.x;
  if (J >= count) return;
  const int P = up[J];
  const int top = piv[ptr[J + 1] - 1];
  const int64_t t0 = Lp[top], t1 = Lp[top + 1];
  if (P < 0) { if (t1 > t0) atomicOr(err, 1); return; }
  const int q0 = ptr[P], sP = ptr[P + 1] - q0, topP = piv[ptr[P + 1] - 1];
  const int64_t p0 = Lp[topP], p1 = Lp[topP + 1];
  int64_t lo = p0;  // the border rows ascend, so does their place in the parent's list
  for (int64_t t = t0; t < t1; t++) {
    const int i = Li[t];
    int r;
    if (snof[i] == P) r = slot[i] - q0;
    else {
      int64_t l = lo, h = p1;
      while (l < h) { const int64_t mid = (l + h) >> 1; if (Li[mid] < i) l = mid + 1; else h = mid; }
      if (l >= p1 || Li[l] != i) { atomicOr(err, 2); l = p0; }
      lo = l;
      r = sP + (int)(l - p0);
    }
    rel[reloff[J] + (t - t0)] = (uint16_t)r;
  }
}
// setup: row of its column's front for every entry of L (one thread per entry)
__global__ __launch_bounds__(kBlock) void k_mf_loc(int64_t nnzL, const int *__restrict__ Lcol, const int *__restrict__ Li,
                                                   const int64_t *__restrict__ Lp, const int *__restrict__ ptr, const int *__restrict__ piv,
                                                   const int *__restrict__ snof, const int *__restrict__ slot, uint16_t *__restrict__ loc,
                                                   int *__restrict__ err) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= nnzL) return;
  const int k = Lcol[t], i = Li[t], J = snof[k];
  const int q0 = ptr[J], s = ptr[J + 1] - q0;
  int r;
  if (snof[i] == J) r = slot[i] - q0;
  else {
    const int top = piv[ptr[J + 1] - 1];
    const int64_t p0 = Lp[top], p1 = Lp[top + 1];
    int64_t l = p0, h = p1;
    while (l < h) { const int64_t mid = (l + h) >> 1; if (Li[mid] < i) l = mid + 1; else h = mid; }
    if (l >= p1 || Li[l] != i) { atomicOr(err, 4); l = p0; }
    r = s + (int)(l - p0);
  }
  loc[t] = (uint16_t)r;
}

// ---- device side of a lean analysis (direct.hip LdlFactor::lean_device_*) ------------------------------------------------
// where entry k of triu(P) (row_offset 0: KKT nodes rowidx, colid) or of A (row_offset n: nodes colid = the variable,
// n + rowidx = the constraint row) sits in L: position in Lx, or -(pivot) - 1 for a diagonal entry
__global__ __launch_bounds__(kBlock) void k_lean_map(int64_t nnz, const int *__restrict__ colid, const int *__restrict__ rowidx, int row_offset,
                                                     const int *__restrict__ pinv, const int64_t *__restrict__ Lp, const int *__restrict__ Li,
                                                     int64_t *__restrict__ map) {
  const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k >= nnz) return;
  const int a = pinv[colid[k]], b = pinv[rowidx[k] + row_offset];
  if (a == b) { map[k] = -(int64_t)a - 1; return; }
  const int c = a < b ? a : b, r = a < b ? b : a;
  int64_t l = Lp[c], h = Lp[c + 1];
  while (l < h) { const int64_t mid = (l + h) >> 1; if (Li[mid] < r) l = mid + 1; else h = mid; }
  map[k] = l;  // present by construction: every entry of K is an entry of L
}
// sort keys of the entries of L outside the diagonal blocks: pass 0 (slot of the row, slot of the column), pass 1 the other
// way round; row -1 = inside a block (dropped by the sort)
__global__ __launch_bounds__(kBlock) void k_lean_keys(int64_t nnzL, const int *__restrict__ Lcol, const int *__restrict__ Li,
                                                      const int *__restrict__ snof, const int *__restrict__ slot, int pass,
                                                      int *__restrict__ er, int *__restrict__ ec) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (t >= nnzL) return;
  const int j = Lcol[t], k = Li[t];
  const bool inside = snof[j] == snof[k];
  const int sr = slot[k], sc = slot[j];
  er[t] = inside ? -1 : (pass == 0 ? sr : sc);
  ec[t] = pass == 0 ? sc : sr;
}
__global__ __launch_bounds__(kBlock) void k_lean_rowid(int64_t n, const int *__restrict__ walk, const int *__restrict__ rowid, int *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) out[i] = rowid[walk[i]];
}
__global__ __launch_bounds__(kBlock) void k_widen(int64_t n, const int *__restrict__ in, int64_t *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) out[i] = in[i];
}
// per forward row: the first entry that points at a slot >= q_upper
__global__ __launch_bounds__(kBlock) void k_lean_split(int N, const int64_t *__restrict__ Fp, const int *__restrict__ Fj, int q_upper,
                                                       int64_t *__restrict__ Fsplit) {
  const int q = blockIdx.x * kBlock + threadIdx.x;
  if (q >= N) return;
  int64_t l = Fp[q], h = Fp[q + 1];
  while (l < h) { const int64_t mid = (l + h) >> 1; if (Fj[mid] < q_upper) l = mid + 1; else h = mid; }
  Fsplit[q] = l;
}

}  // namespace
}  // namespace oq
