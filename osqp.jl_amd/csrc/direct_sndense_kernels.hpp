// direct_sndense_kernels.hpp -- a dense top block OVER a supernodal factor (round 6), included by direct.hip.
//
// The top of a nested-dissection tree of a 2-D structure is a chain: the root separator of a 700 x 700 grid is eleven
// supernodes of 64 pivots one above the other, its two children six each, ... -- 23 of the 49 supernode levels hold seven
// separators, and each level is one dependent hand-over per direction in the solves (k_sn_tree: ~17 + 11 us a level there)
// and one launch of a handful of large fronts in the factorisation (k_mfb_panel: ~300 us a level).  The level-scheduled
// factor has had an answer to top chains since round 1 -- the last kD pivots are not factorised, their Schur complement is
// inverted explicitly and a solve replaces both chains by one dense product (direct_dense_kernels.hpp) -- but a supernodal
// factor switched it off (direct.hip: `if (sn) kD = 0`).  This file is the same idea on the supernode partition:
//
//   D = the supernodes of the levels [Ld, nlev) -- upward closed, K pivots in the slots [q0, N).
//   factorisation   the fronts stop below D; S = K_DD + (update matrices of the BOUNDARY children: supernodes below D whose
//                   parent is in D) is assembled as a dense K x K array -- k_snd_init (entries of K, and the one-row children:
//                   the pendant constraint row of a variable adds one number to one diagonal entry), k_snd_extend (a workgroup
//                   per 64 x 64 tile of the lower triangle, the children that reach its rows in ascending order: a fixed order
//                   of sums, no atomics) -- and inverted by the block Gauss-Jordan sweeps on the matrix cores (k_gj_*), whose
//                   pivots are the ones LDL' would meet: the inertia count is unchanged.
//   solve           the levels below D as before; then  t = b_D - (the entries of the rows of D that point below the top
//                   part, gathered) - (the front vectors of the boundary children inside the top part, children in ascending
//                   order),  x_D = S^-1 t  as one symmetric product (k_dense_apply_sym), and the backward sweep below D
//                   starts from there.
#pragma once
#include "direct_dense_kernels.hpp"

namespace oq {
namespace {

// setup: slot (relative to q0) of every border row of every boundary child; block per child
// (the rows and columns of S are the pivots of D in an order of their own, `dpos`: slot - q0 -> position -- the separators one
// after the other, children before parents, instead of interleaved level by level as the slots are: the block pattern of the
// sweeps, direct.hip gj_symbolic, is only sparse when a 64-block holds ONE separator's pivots)
__global__ __launch_bounds__(256) void k_snd_slots(const int *__restrict__ bch, const int *__restrict__ ptr, const int *__restrict__ piv,
                                                   const int64_t *__restrict__ Lp, const int *__restrict__ Li, const int *__restrict__ slot,
                                                   int q0, const int *__restrict__ dpos, const int64_t *__restrict__ boff,
                                                   int *__restrict__ bslot, int *__restrict__ err) {
  const int k = blockIdx.x, J = bch[k];
  const int top = piv[ptr[J + 1] - 1];
  const int64_t t0 = Lp[top];
  const int b = (int)(Lp[top + 1] - t0);
  for (int i = threadIdx.x; i < b; i += 256) {
    const int r = slot[Li[t0 + i]] - q0;
    if (r < 0) atomicOr(err, 8);  // a border row of a child of D below D: the set would not be upward closed
    bslot[boff[k] + i] = r < 0 ? 0 : dpos[r];
  }
}

// entries of K in the columns of D (they sit in Lx / D after the scatter kernels of the assembly: no front touches these
// columns), both triangles; the one-row children of a pivot onto its diagonal entry, in ascending order.  Wavefront per slot.
__global__ __launch_bounds__(64) void k_snd_init(int q0, int ld, const int *__restrict__ piv, const int *__restrict__ slot,
                                                 const int *__restrict__ dpos, const int64_t *__restrict__ Lp, const int *__restrict__ Li,
                                                 const double *__restrict__ Lx, const double *__restrict__ D, const int *__restrict__ pend_ptr,
                                                 const int64_t *__restrict__ pend_src, const double *__restrict__ U, double *__restrict__ S) {
  const int k = piv[q0 + blockIdx.x], c = dpos[blockIdx.x];
  if (threadIdx.x == 0) {
    double d = D[k];
    for (int p = pend_ptr[c]; p < pend_ptr[c + 1]; p++) d += U[pend_src[p]];
    S[(size_t)c * (ld + 1)] = d;
  }
  for (int64_t e = Lp[k] + threadIdx.x; e < Lp[k + 1]; e += 64) {
    const int r = dpos[slot[Li[e]] - q0];
    const double v = Lx[e];
    S[(size_t)r + (size_t)c * ld] = v;
    S[(size_t)c + (size_t)r * ld] = v;
  }
}

// extend-add of the boundary children's update matrices into S: workgroup per 64 x 64 tile (ti >= tj), thread (ty, tx) a
// 4 x 4 piece.  The children that have a border row in tile row ti come from a host-built list (ascending); a child is
// skipped when none of its rows falls into the tile's columns.  Per child: the inverse maps  row / column of the tile ->
// border row of the child  in LDS (the border rows of a child are distinct slots), then every entry of the tile is summed by
// its own thread.  The tile is written to both triangles (the block sweeps read the diagonal tiles whole).
struct SndExtArgs {
  const int *tiles;           // (ti, tj) per workgroup
  const int *trow_ptr, *trow_list;  // per tile row: indices into bch of the children with a border row there
  const int *bch, *bsz;
  const int64_t *uoff, *boff;
  const int *bslot, *crange;  // crange[2k], [2k+1]: smallest / largest slot of child k
  const double *U;
  double *S;
  int ld;
};
__global__ __launch_bounds__(256) void k_snd_extend(SndExtArgs a) {
  __shared__ unsigned short invR[64], invC[64];
  const int tid = threadIdx.x, tx = tid % 16, ty = tid / 16;
  const int ti = a.tiles[2 * blockIdx.x], tj = a.tiles[2 * blockIdx.x + 1];
  const int r0 = ti * 64, c0 = tj * 64;
  double acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = a.S[(size_t)(r0 + ty * 4 + i) + (size_t)(c0 + tx * 4 + j) * a.ld];
  for (int h = a.trow_ptr[ti]; h < a.trow_ptr[ti + 1]; h++) {
    const int k = a.trow_list[h];
    if (a.crange[2 * k] >= c0 + 64 || a.crange[2 * k + 1] < c0) continue;  // (uniform over the workgroup)
    const int J = a.bch[k], b = a.bsz[J];
    const int *bs = a.bslot + a.boff[k];
    const double *Uc = a.U + a.uoff[J];
    __syncthreads();
    if (tid < 64) { invR[tid] = 0xFFFF; invC[tid] = 0xFFFF; }
    __syncthreads();
    for (int i = tid; i < b; i += 256) {
      const int r = bs[i];
      if (r >= r0 && r < r0 + 64) invR[r - r0] = (unsigned short)i;
      if (r >= c0 && r < c0 + 64) invC[r - c0] = (unsigned short)i;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const unsigned ri = invR[ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const unsigned cj = invC[tx * 4 + j];
        if (ri == 0xFFFFu || cj == 0xFFFFu || r0 + ty * 4 + i < c0 + tx * 4 + j) continue;
        const unsigned hi = ri > cj ? ri : cj, lo = ri > cj ? cj : ri;  // the child's triangle is packed in ITS row order
        acc[i][j] += Uc[(int64_t)lo * (2 * b - (int64_t)lo - 1) / 2 + hi];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int R = r0 + ty * 4 + i, C = c0 + tx * 4 + j;
      if (R < C) continue;
      a.S[(size_t)R + (size_t)C * a.ld] = acc[i][j];
      a.S[(size_t)C + (size_t)R * a.ld] = acc[i][j];
    }
}

// t = b_D - (entries of the row before `Fd`: what lies below the part that hands front vectors up -- or below D itself)
//         - (the entries of the boundary children's front vectors that belong to this row, children in ascending order:
//            `vptr` / `vsrc`, positions in uvec; vptr == nullptr: no vectors);
// wavefront per row of D.  (The vectors were subtracted child after child by ONE workgroup at first -- a fixed order of sums --
// which cost 21 us of a 500 us iteration for twenty children: twenty dependent trips to memory.  A row's own list has the same
// order and a handful of entries, loaded by as many lanes at once.)
__global__ __launch_bounds__(kBlock) void k_snd_rhs(int q0, int K, const int *__restrict__ dpos, const int64_t *__restrict__ Fp,
                                                    const int64_t *__restrict__ Fd, const int *__restrict__ Fj, const double *__restrict__ Fx,
                                                    const double *__restrict__ b, const int *__restrict__ vptr, const int64_t *__restrict__ vsrc,
                                                    const double *__restrict__ uvec, double *__restrict__ t) {
  const int lane = threadIdx.x & 63;
  const int r = (int)(((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6);
  if (r >= K) return;
  const int q = q0 + r, a = dpos[r];
  double acc = gather_dot(Fp[q] + lane, Fd[q], 64, Fj, Fx, b);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  double v = b[q] - acc;
  if (vptr) {
    const int v0 = vptr[a], nv = vptr[a + 1] - v0;
    for (int base = 0; base < nv; base += 64) {
      const double u = base + lane < nv ? uvec[vsrc[v0 + base + lane]] : 0.0;
      const int cnt = nv - base < 64 ? nv - base : 64;
      for (int i = 0; i < cnt; i++) v -= __shfl(u, i, 64);
    }
  }
  if (lane == 0) t[a] = v;
}

}  // namespace
}  // namespace oq
