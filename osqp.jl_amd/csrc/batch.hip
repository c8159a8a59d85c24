// batch.hip -- batched small-QP path (rows K11/K12 of SURVEY.md section 8a,
// BASELINE.json config 5: 4096 independent MPC QPs, n = 100, m = 200).
//
// Two kernels behind one launcher (launch_batch):
//   * k_batch_quad (batch_quad.hpp, round 4) -- one QP per FOUR wavefronts, three QPs per compute unit, the inverse of the
//     reduced KKT matrix in registers as four quadrants, formed there by Gauss-Jordan sweeps, applied with
//     v_fmac_f64_dpp row_newbcast; nothing of a factorisation touches global memory.  Takes the patterns that fit its
//     compile-time bounds -- the MPC family of the benchmark;
//   * k_batch_solve (this file, rounds 1-3) -- one QP per 512-thread workgroup, two per compute unit, for every other
//     pattern with n <= 128 (run-time shapes, dense P, long rows):
//       - LDS (~75 KB at n = 100, m = 200): the instance's values of A (shared sparsity pattern, CSC order) and of the full
//         symmetric P, q, l, u, the Ruiz scalings, all ADMM iterates, and the thread's share of the pattern of A in both
//         orientations as packed (value offset, operand offset) words;
//       - REGISTERS: the inverse of the reduced KKT matrix M = P + sigma I + A' diag(rho) A -- thread 4 i + c of the 512
//         holds M^-1[i, c*n/4 .. (c+1)*n/4) (25 doubles at n = 100): the four column parts of a row sit in neighbouring
//         lanes, so the dense product ends in two DPP adds instead of a trip through LDS;
//       - M is assembled from host-precomputed term lists in a per-instance n x n scratch in GLOBAL memory (L2-resident) and
//         inverted there by Gauss-Jordan block sweeps on the fp64 matrix cores (invert_mfma), then loaded into the register
//         tiles.  (The on-chip variant of this factorisation was built in round 3 and measured slower for this
//         decomposition: profiles/r03_batch_experiments.md.)
// Per iteration, three barrier-separated phases: b = sigma x - q + A'(rho z - y) (4 lanes per column of A);
// x~ = M^-1 b (registers x LDS broadcast) with the x update; z~ = A x~ consumed row by row by the z / y update (2 lanes
// per row); every `check_termination` iterations the same residual / infeasibility tests as the large-problem path.
// The 512-thread kernel is issue-bound, not latency-bound (round 3: 16 waves per CU, every vector instruction of a wavefront is four
// cycles of its SIMD): the iteration is written for instruction count -- operand addresses come ready-made out of one
// packed word (two instructions per entry), short columns / rows are padded with a zero-valued entry instead of
// branching, nothing the loop needs lives in spilled registers.
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
int g_batch_last_kernel = -2;  // what launch_batch launched last: -1 the 512-thread kernel, >= 0 the entry of DevicePattern::kQuadCfg
inline bool batch_quad_enabled() {  // OSQP_AMD_BATCH_QUAD=0: the MPC family on the 512-thread kernel (A/B runs, tests)
  const char *e = getenv("OSQP_AMD_BATCH_QUAD");
  return !e || atoi(e) != 0;
}
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
  int max_col, max_row;             // longest column / row of A
};

// LDS pointers carry their address space in the type: a plain `double *` into LDS is a 64-bit generic pointer whose
// accesses compile to flat_load / flat_store (the slow path into LDS, and two registers per pointer) -- with these
// every access is a ds_read / ds_write on a 32-bit offset.
typedef __attribute__((address_space(3))) double ldouble;
typedef __attribute__((address_space(3))) unsigned short lshort;
typedef __attribute__((address_space(3))) int lint;
typedef __attribute__((address_space(3))) char lchar;

__device__ __forceinline__ double nmax(double a, double b) { return (a > b || a != a) ? a : b; }
__device__ __forceinline__ double lim(double v) { v = v < B_MIN_SCALING ? 1.0 : v; return v > B_MAX_SCALING ? B_MAX_SCALING : v; }

// a value every lane holds identically, moved to scalar registers (the compiler cannot see that an LDS broadcast is uniform)
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ double uni(double v) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readfirstlane(u.i[0]);
  u.i[1] = __builtin_amdgcn_readfirstlane(u.i[1]);
  return u.d;
}

// Hides where a (wave-uniform) pointer comes from: address arithmetic on it cannot be hoisted out of the enclosing loop.
// The rarely taken phases of the ADMM loop (factorisation, residual evaluation) would otherwise park dozens of
// precomputed addresses in registers -- or spill them -- across the iterations that never use them.
template <typename T>
__device__ __forceinline__ T *opaque(T *p) {
  asm volatile("" : "+s"(p));
  return p;
}
// The same for per-thread values: the thread id as a value the optimiser cannot trace (every use site gets its own copy, so
// nothing derived from it -- row ids, LDS addresses, predicates -- is a loop invariant worth keeping).
__device__ __forceinline__ int mytid() {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}

// lane `lane` (a compile-time constant) of v, as a wave-uniform value: two v_readlane into scalar registers.  A vector of up
// to 64 doubles that every lane of a wavefront needs (a pivot row, a right-hand side) is read from LDS ONCE -- lane l takes
// element l -- and handed round this way: no vector registers, no further LDS traffic.
__device__ __forceinline__ double lane_bcast(double v, int lane) {
  union { double d; int i[2]; } a;
  a.d = v;
  a.i[0] = __builtin_amdgcn_readlane(a.i[0], lane);
  a.i[1] = __builtin_amdgcn_readlane(a.i[1], lane);
  return a.d;
}

// v of the lane whose id differs in bit 0 (kXor = 1) or bit 1 (kXor = 2): a DPP quad permutation on the two halves -- a
// plain VALU move, where __shfl_xor goes through the LDS crossbar (ds_bpermute) and its queue
template <int kXor>
__device__ __forceinline__ double quad_xor(double v) {
  constexpr int ctrl = kXor == 1 ? 0xB1 : 0x4E;  // quad_perm [1,0,3,2] / [2,3,0,1]
  union { double d; int i[2]; } a;
  a.d = v;
  a.i[0] = __builtin_amdgcn_mov_dpp(a.i[0], ctrl, 0xF, 0xF, true);
  a.i[1] = __builtin_amdgcn_mov_dpp(a.i[1], ctrl, 0xF, 0xF, true);
  return a.d;
}

// K simultaneous block reductions (max for op 0, sum for op 1); result broadcast to every thread
template <int K>
__device__ __forceinline__ void block_reduce(double *v, int op, ldouble *red) {
#pragma unroll
  for (int k = 0; k < K; k++) {
    double a = v[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { double b = __shfl_xor(a, o, 64); a = op ? a + b : nmax(a, b); }
    v[k] = a;
  }
  __syncthreads();
  if ((mytid() & 63) == 0)
    for (int k = 0; k < K; k++) red[(mytid() >> 6) * K + k] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; k++) {
    double a = red[k];
#pragma unroll
    for (int w = 1; w < NW; w++) a = op ? a + red[w * K + k] : nmax(a, red[w * K + k]);
    v[k] = a;
  }
}

typedef __attribute__((address_space(3))) unsigned luint;
constexpr int KT = 4, KR = 6;  // entries per lane: 4 lanes per column of A (A' v), 2 lanes per row (A v)
constexpr int KRW = 8;         // dwords per thread of the row-side words (6 used: a 16-byte and an 8-byte read)

struct Lds {
  ldouble *gjc;  // 2 x (4 x 128): the four pivot rows of a block step of the MFMA inversion, double-buffered; behind them
                 // 2 x 32: the inverse of the 4 x 4 pivot block and its positive-definite flag
  ldouble *bb, *Av, *Pv, *q, *l, *u, *D, *E, *rho, *rhoi, *x, *z, *y, *xp, *zp, *xt, *zt, *dx, *dy, *Ax, *Px, *Aty, *tn, *tm, *red, *ldinv, *nrm;
  lint *ctype;
  lshort *Ap, *Ai, *Rp, *Rc, *Rmap, *Fp, *Fc;  // shared pattern, staged into LDS as 16-bit indices
  luint *cw, *rw;  // the thread's entries of A, column side (KT words per thread) and row side (KRW per thread): (byte offset of
                   // the value in Av) << 16 | byte offset of the operand; only when the pattern fits (sparse_fits)
  int ld;
};
// Av carries one more double than A has entries: a zero that padded entries point at
__host__ __device__ inline size_t lds_doubles(int n, int m, int nnzA, int nnzF) {
  return (size_t)n + nnzA + 1 + nnzF + 10 * (size_t)n + 12 * (size_t)m + 16 * NW + 24 + 2 * 4 * 128 + 64 + 1;  // + 1: alignment slack
}
__host__ __device__ inline size_t lds_shorts(int n, int m, int nnzA, int nnzF) {
  return 2 * ((size_t)n + 1) + ((size_t)m + 1) + 3 * (size_t)nnzA + (size_t)nnzF + 8;
}
__host__ __device__ inline size_t lds_bytes(int n, int m, int nnzA, int nnzF, bool words) {
  return lds_doubles(n, m, nnzA, nnzF) * 8 + (((size_t)m * 4 + 15) / 16) * 16 + (words ? (size_t)NT * (KT + KRW) * 4 : 0) +
         lds_shorts(n, m, nnzA, nnzF) * 2 + 16;
}
__device__ __forceinline__ Lds carve(ldouble *base, const Pattern &P, bool words) {
  Lds s;
  const int n = P.n, m = P.m;
  s.ld = n + 1;
  ldouble *p = base;
  s.bb = p; p += n;  // the right-hand side of the reduced system
  s.Av = p; p += P.nnzA + 1; s.Pv = p; p += P.nnzF;
  s.q = p; p += n; s.D = p; p += n; s.x = p; p += n; s.xp = p; p += n; s.xt = p; p += n; s.dx = p; p += n;
  s.Px = p; p += n; s.Aty = p; p += n; s.tn = p; p += n; s.ldinv = p; p += n;
  s.l = p; p += m; s.u = p; p += m; s.E = p; p += m; s.rho = p; p += m; s.rhoi = p; p += m; s.z = p; p += m; s.y = p; p += m;
  s.zp = p; p += m; s.zt = p; p += m; s.dy = p; p += m; s.Ax = p; p += m; s.tm = p; p += m;
  s.red = p; p += 16 * NW;  // NW * K doubles of block_reduce, K <= 14
  s.nrm = p; p += 24;
  s.gjc = p; p += 2 * 4 * 128 + 64;
  p += (p - base) & 1;  // what follows starts on a 16-byte boundary
  s.ctype = (lint *)p;
  luint *w = (luint *)((lchar *)p + (((size_t)m * 4 + 15) / 16) * 16);  // 16-byte aligned: the words are read four at a time
  s.cw = w; s.rw = w + (words ? NT * KT : 0);
  lshort *h = (lshort *)(w + (words ? NT * (KT + KRW) : 0));
  s.Ap = h; h += n + 1; s.Fp = h; h += n + 1; s.Rp = h; h += m + 1;
  s.Ai = h; h += P.nnzA; s.Rc = h; h += P.nnzA; s.Rmap = h; h += P.nnzA; s.Fc = h;
  return s;
}

// y = A x (CSR), y = A' x (CSC), y = P x (full symmetric CSR); no barriers inside.  L lanes share a row (the index ->
// value -> operand chain of LDS reads is latency-bound: 8 entries walked by one lane cost 8 round trips, by 4 lanes 2)
// and add up with xor shuffles, so every thread of the workgroup reaches the shuffles whether it has a row or not.
// finish(r, sum) runs on one lane per row.
template <int L, typename F, typename G>
__device__ __forceinline__ void rows_dot(int rows, const lshort *ptr, F term, G finish) {
  const int lane = mytid() & (L - 1);
  for (int base = 0; base < rows; base += NT / L) {
    const int r = base + mytid() / L;
    double a = 0.0;
    if (r < rows)
      for (int q = ptr[r] + lane; q < ptr[r + 1]; q += L) a += term(q);
#pragma unroll
    for (int o = L >> 1; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if (lane == 0 && r < rows) finish(r, a);
  }
}
__device__ __forceinline__ void mul_P(const Pattern &P, const Lds &s, const ldouble *x, ldouble *y) {
  rows_dot<4>(P.n, s.Fp, [&](int q) { return s.Pv[q] * x[s.Fc[q]]; }, [&](int r, double a) { y[r] = a; });
}

__device__ __forceinline__ void set_rho(const Pattern &P, const Lds &s, double rho, bool classify) {
  for (int i = mytid(); i < P.m; i += NT) {
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

// ---------------------------------------------------------------------------------------------------------
// The reduced KKT matrix and its inverse, in registers.
//
// Thread t = 4 * row + part owns row `row` (rows >= n idle) and the NC = ceil(n / PARTS) columns
// [part * NC, (part + 1) * NC) of the n x n array: T.v[u] = M^-1[row, j0 + u].  NCT is the compile-time bound of NC.
// The four parts of a row are neighbouring lanes: the dense product M^-1 b ends in two quad-permute adds.
//
// Why an explicit inverse: it is applied to thousands of right-hand sides (one per ADMM iteration) between two rho
// updates, and a dense product M^-1 b keeps every row independent while a triangular solve is a chain of 2n dependent
// steps.  Why Gauss-Jordan sweeps (Goodnight's sweep operator): a block sweep is a rank-4 update of the whole array on the
// matrix cores (invert_mfma below) -- no serial column loop anywhere.  The pivots are the Schur complements of M, so the
// positive-definiteness test of the Cholesky factorisation carries over unchanged.
// ---------------------------------------------------------------------------------------------------------
constexpr int PARTS = 4;
static_assert(NT == 512, "the register tiles assume 128 rows x 4 column parts");

template <int NCT>
struct MTile {
  double v[NCT];
};

// Addressing rule of the routines below: ONE base register per array and compile-time offsets u / u ld.  (Written with the
// global column j = j0 + u, the 25 column ids and 25 clamped addresses are loop invariants that the compiler precomputes
// and then has to keep in -- or spill from -- registers across the whole ADMM loop.)  EXACT: PARTS * NCT == n, no
// column of a tile lies outside the matrix; otherwise columns u >= ncv of the last part are masked.
// the thread's tile (row = tid >> 2, NCT columns of part tid & 3) of the n x n array in the instance's scratch
template <int NCT, bool EXACT>
__device__ __forceinline__ void load_tile(int n, const double *scratch, MTile<NCT> &T) {
  scratch = opaque(scratch);
  const int ld = n;
  const int row = mytid() >> 2, part = mytid() & 3;
  const int nc = EXACT ? NCT : (n + PARTS - 1) / PARTS, j0 = part * nc;
  const int ncv = EXACT ? NCT : max(0, min(nc, n - j0));
  const bool live = row < n;
  const double *src = scratch + (live ? row : 0) + (size_t)min(j0, n - 1) * ld;
#pragma unroll
  for (int u = 0; u < NCT; u++) T.v[u] = (live && (EXACT || u < ncv)) ? src[(size_t)u * ld] : 0.0;
}
// M = P + sigma I + A' diag(rho) A: the lower triangle from the host-precomputed term lists (the intersections of the
// columns of A do not depend on the instance), both triangles into the instance's scratch (global memory, n x n,
// column-major)
__device__ __forceinline__ void assemble_scratch(const Pattern &P, const Lds &s, double sigma, double *__restrict__ scratch) {
  const int n = P.n, ld = n;
  scratch = opaque(scratch);
  for (int e = mytid(); e < n * ld; e += NT) scratch[e] = 0.0;
  __syncthreads();
  for (int t = mytid(); t < P.npair; t += NT) {
    double acc = 0.0;
    const int q1 = P.Tp[t + 1];
    int q = P.Tp[t];
    for (; q + 4 <= q1; q += 4) {  // the index triples of four terms first (global memory), then their LDS operands, then the sum in order
      unsigned short tr[4], ta[4], tb[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { tr[u] = P.Tr[q + u]; ta[u] = P.Ta[q + u]; tb[u] = P.Tb[q + u]; }
      double r[4], a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { r[u] = s.rho[tr[u]]; a[u] = s.Av[ta[u]]; b[u] = s.Av[tb[u]]; }
#pragma unroll
      for (int u = 0; u < 4; u++) acc += r[u] * a[u] * b[u];
    }
    for (; q < q1; q++) acc += s.rho[P.Tr[q]] * s.Av[P.Ta[q]] * s.Av[P.Tb[q]];
    const int i = P.Ti[t], j = P.Tj[t];
    scratch[i + j * ld] = acc;
    scratch[j + i * ld] = acc;
  }
  __syncthreads();
  for (int i = mytid(); i < n; i += NT) scratch[i + i * ld] += sigma;
  __syncthreads();
  for (int r = mytid(); r < n; r += NT)
    for (int q = s.Fp[r]; q < s.Fp[r + 1]; q++) scratch[r + s.Fc[q] * ld] += s.Pv[q];  // full symmetric P: both triangles
  __syncthreads();
}


// ---- the same inverse by block sweeps on the matrix cores ---------------------------------------------------------
// Four pivots at a time: with K the pivot indices, C = M[K, :] (4 x n) and G = M[K, K]^-1 the sweep operator is
//   M <- M - C' (G C),  then  M[K, R] <- G C (and its mirror),  M[K, K] <- -G,
// i.e. one rank-4 update of the whole array -- v_mfma_f64_16x16x4_f64 per 16 x 16 tile -- and a patch of four rows and
// columns, instead of four rank-1 updates whose pivot row has to be broadcast element by element (invert_tile: 50
// v_readlane per 25 multiply-adds, ~10 % of the fp64 rate).  The array stays symmetric, so only the tiles on and below
// the diagonal are kept, TPW per wavefront in accumulator layout (lane: column lane & 15, rows (lane >> 4) + 4 r);
// indices >= n are padded with the identity and never swept.  One barrier per block step: the pivot rows go through two
// alternating 4 x 128 LDS buffers.  In: M in the instance's scratch (both triangles, ld = n); out: M^-1 there.
typedef double d4_t __attribute__((ext_vector_type(4)));
template <int TPW>
__device__ __forceinline__ bool invert_mfma(int n, double *scratch, ldouble *cb) {
  scratch = opaque(scratch);
  const int lane = mytid() & 63, wave = uni(mytid() >> 6);
  const int lr = lane >> 4, lc = lane & 15;
  const int NB = (n + 15) >> 4, ntiles = NB * (NB + 1) / 2, steps = (n + 3) >> 2;
  d4_t acc[TPW];
  int TI[TPW], TJ[TPW];
#pragma unroll
  for (int s = 0; s < TPW; s++) {
    const int t = wave + s * NW;
    int I = -1, J = -1;
    if (t < ntiles) { I = 0; while ((I + 1) * (I + 2) / 2 <= t) I++; J = t - I * (I + 1) / 2; }
    TI[s] = uni(I); TJ[s] = uni(J);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = I * 16 + lr + 4 * r, col = J * 16 + lc;
      double v = (row == col) ? 1.0 : 0.0;
      if (I >= 0 && row < n && col < n) v = scratch[row + (size_t)col * n];
      acc[s][r] = v;
    }
  }
  bool ok = true;
  for (int tb = 0; tb < NB; tb++) {
#pragma unroll
    for (int tq = 0; tq < 4; tq++) {
      const int step = tb * 4 + tq;
      if (step < steps) {
        ldouble *C = cb + (step & 1) * 512;  // C[a][j] at a * 128 + j
        ldouble *Gs = cb + 1024 + (step & 1) * 32;
        // publish rows k0 .. k0 + 3: left of and inside block tb from the tiles of row block tb (register tq of every
        // lane), right of it from column k0 + a of the tiles below (the array is symmetric).  The wavefront that owns the
        // diagonal tile also inverts the 4 x 4 pivot block -- it sits in register tq of its lanes 16 a + 4 tq + b -- while
        // the others wait at the barrier: one Gauss-Jordan per block step instead of one per wavefront.
#pragma unroll
        for (int s = 0; s < TPW; s++) {
          if (TI[s] == tb) C[lr * 128 + TJ[s] * 16 + lc] = acc[s][tq];
          if (TJ[s] == tb && TI[s] > tb && (lc >> 2) == tq) {
#pragma unroll
            for (int r = 0; r < 4; r++) C[(lc & 3) * 128 + TI[s] * 16 + lr + 4 * r] = acc[s][r];
          }
          if (TI[s] == tb && TJ[s] == tb) {
            double g[4][4];
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
              for (int b = 0; b < 4; b++) g[a][b] = lane_bcast(acc[s][tq], 16 * a + 4 * tq + b);
            bool pd = true;
#pragma unroll
            for (int p = 0; p < 4; p++) {
              if (!(g[p][p] > 0.0)) pd = false;
              double d = __builtin_amdgcn_rcp(g[p][p]);                 // v_rcp_f64 and two Newton steps instead of the
              d = __builtin_fma(__builtin_fma(-g[p][p], d, 1.0), d, d);  // ~40-instruction chain of the IEEE division
              d = __builtin_fma(__builtin_fma(-g[p][p], d, 1.0), d, d);
#pragma unroll
              for (int j = 0; j < 4; j++) if (j != p) g[p][j] *= d;
#pragma unroll
              for (int i = 0; i < 4; i++) if (i != p) {
                const double f = g[i][p];
#pragma unroll
                for (int j = 0; j < 4; j++) if (j != p) g[i][j] = __builtin_fma(-f, g[p][j], g[i][j]);
                g[i][p] = -f * d;
              }
              g[p][p] = d;
            }
            if (lane == 0) {
#pragma unroll
              for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) Gs[a * 4 + b] = g[a][b];
              Gs[16] = pd ? 1.0 : 0.0;
            }
          }
        }
        __syncthreads();
        if (Gs[16] == 0.0) ok = false;
        double gl[4], gc[4];  // rows lane >> 4 and lane & 3 of G
#pragma unroll
        for (int b = 0; b < 4; b++) { gl[b] = Gs[lr * 4 + b]; gc[b] = Gs[(lc & 3) * 4 + b]; }
        const double gdiag = Gs[lr * 4 + (lc & 3)];
#pragma unroll
        for (int s = 0; s < TPW; s++) {
          if (TI[s] < 0) continue;
          const double aop = -C[lr * 128 + TI[s] * 16 + lc];  // A[i = lane & 15][k = lane >> 4] = -C[k][row i of block I]
          const ldouble *cj = C + TJ[s] * 16 + lc;
          double bop = gl[0] * cj[0];                          // B[k = lane >> 4][j = lane & 15] = (G C)[k][column j of block J]
          bop = __builtin_fma(gl[1], cj[128], bop);
          bop = __builtin_fma(gl[2], cj[256], bop);
          bop = __builtin_fma(gl[3], cj[384], bop);
          acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc[s], 0, 0, 0);
          if (TI[s] == tb) acc[s][tq] = (TJ[s] == tb && (lc >> 2) == tq) ? -gdiag : bop;  // rows K: G C, and -G inside the block
          if (TJ[s] == tb && (lc >> 2) == tq) {                                              // columns K: the mirror
#pragma unroll
            for (int r = 0; r < 4; r++) {
              if (TI[s] == tb && r == tq) continue;  // rows K of the diagonal tile were set above
              const ldouble *ci = C + TI[s] * 16 + lr + 4 * r;
              double w = gc[0] * ci[0];
              w = __builtin_fma(gc[1], ci[128], w);
              w = __builtin_fma(gc[2], ci[256], w);
              w = __builtin_fma(gc[3], ci[384], w);
              acc[s][r] = w;
            }
          }
        }
      }
    }
  }
  // the sweeps leave -M^-1; both triangles back into the scratch
#pragma unroll
  for (int s = 0; s < TPW; s++) {
    if (TI[s] < 0) continue;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = TI[s] * 16 + lr + 4 * r, col = TJ[s] * 16 + lc;
      if (row < n && col < n) {
        const double v = -acc[s][r];
        scratch[row + (size_t)col * n] = v;
        scratch[col + (size_t)row * n] = v;
      }
    }
  }
  __syncthreads();
  return ok;
}

// LDS through a byte offset: with the arrays at compile-time addresses (the shape-specialised kernel) the base is an
// immediate of the ds_read and the offset register goes in as it comes out of the packed word
__device__ __forceinline__ double lds_at(const ldouble *base, unsigned byte_off) {
  return *(const ldouble *)((const lchar *)base + byte_off);
}

// x~ = M^-1 b with b in s.bb: the thread's NCT entries of row `row` against its stretch of b (every lane of a quad reads
// its own part: four addresses per wavefront, each a broadcast), then the four parts of the row add up inside the quad.
// The lane of part 0 finishes the row on the spot: x~ to s.xt (the row-side product reads it), x and delta_x.
template <int NCT, bool EXACT>
__device__ __forceinline__ void apply_tile(int n, const MTile<NCT> &T, const Lds &s, double alpha, const ldouble *xp, ldouble *x) {
  const int row = mytid() >> 2, part = mytid() & 3;
  const int nc = EXACT ? NCT : (n + PARTS - 1) / PARTS, j0 = part * nc;
  const int ncv = EXACT ? NCT : max(0, min(nc, n - j0));
  const ldouble *bp = s.bb + j0;
  double bj[NCT];
#pragma unroll
  for (int u = 0; u < NCT; u++) bj[u] = (EXACT || u < ncv) ? bp[u] : 0.0;
  const double xo = xp[row < n ? row : 0];
  double a0 = 0.0, a1 = 0.0;
#pragma unroll
  for (int u = 0; u < NCT; u++) {
    if (u & 1) a1 = __builtin_fma(T.v[u], bj[u], a1); else a0 = __builtin_fma(T.v[u], bj[u], a0);
  }
  double a = a0 + a1;
  a += quad_xor<2>(a);
  a += quad_xor<1>(a);
  if (part == 0 && row < n) {
    s.xt[row] = a;
    const double xn = alpha * a + (1.0 - alpha) * xo;
    x[row] = xn;
    s.dx[row] = xn - xo;
  }
}

// ---------------------------------------------------------------------------------------------------------
// The thread's share of the pattern of A, both orientations, as packed words in LDS (s.cw, s.rw):
//   column side (A' v):  4 lanes per column, lane l of column c walks entries Ap[c] + l, + 4, ...   (KT per lane)
//   row side    (A v):   2 lanes per row,    lane l of row r    walks entries Rp[r] + l, + 2, ...   (KR per lane)
// A word holds (byte offset of the value inside Av) << 16 | (byte offset of the operand inside its vector): both LDS reads
// of an entry take their address from one shift / one mask, where the walk through the pattern arrays is a chain of three
// dependent reads.  A lane with fewer entries than KT / KR is padded with words that point at the zero behind the values
// of A (and at operand 0): the loops have no tails and no predicates, the padded terms add +0.0 and change nothing.
// The words live in LDS, not in registers: the hot loop fetches them with one 16-byte read (two on the row side) -- held
// in registers across the ADMM loop they were the first thing the allocator spilled, and a spilled word came back from
// scratch memory once per entry and iteration (round 2: ~10 dependent round trips to memory per iteration).
// Same lane-strided order and the same quad reduction as rows_dot<4> / rows_dot<2>: bit-identical sums.  Used when the
// pattern fits (at most 4 KT per column, 2 KR per row, 4 n and 2 m threads); rows_dot on the LDS copy otherwise.
// ---------------------------------------------------------------------------------------------------------
__host__ __device__ inline bool sparse_fits(const Pattern &P) {
  // m > 0: a padded entry reads operand 0 of the vector it is multiplied into -- there has to be one (0 x garbage is not 0)
  return P.m > 0 && P.nnzA > 0 && 4 * P.n <= NT && 2 * P.m <= NT && P.max_col <= 4 * KT && P.max_row <= 2 * KR && (size_t)(P.nnzA + 1) * 8 < 65536;
}
__device__ __forceinline__ void store_sparse(const Pattern &P, const Lds &s) {
  const int t = mytid();
  const unsigned pad = (unsigned)(P.nnzA * 8) << 16;  // the zero behind the values, operand 0
  const int col = (t >> 2) < P.n ? (t >> 2) : -1, row = (t >> 1) < P.m ? (t >> 1) : -1;
#pragma unroll
  for (int e = 0; e < KT; e++) {
    unsigned w = pad;
    if (col >= 0) {
      const int k = s.Ap[col] + (t & 3) + 4 * e;
      if (k < s.Ap[col + 1]) w = ((unsigned)(k * 8) << 16) | (unsigned)(s.Ai[k] * 8);
    }
    s.cw[t * KT + e] = w;
  }
#pragma unroll
  for (int e = 0; e < KRW; e++) {
    unsigned w = pad;
    if (row >= 0 && e < KR) {
      const int q = s.Rp[row] + (t & 1) + 2 * e;
      if (q < s.Rp[row + 1]) w = ((unsigned)(s.Rmap[q] * 8) << 16) | (unsigned)(s.Rc[q] * 8);
    }
    s.rw[t * KRW + e] = w;
  }
}
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
typedef unsigned uint2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) uint4_t luint4;
typedef __attribute__((address_space(3))) uint2_t luint2;
struct ColWords { unsigned w[KT]; };
struct RowWords { unsigned w[KR]; };
__device__ __forceinline__ ColWords col_words(const Lds &s) {
  const uint4_t v = *(const luint4 *)(s.cw + mytid() * KT);
  ColWords c;
  c.w[0] = v.x; c.w[1] = v.y; c.w[2] = v.z; c.w[3] = v.w;
  return c;
}
__device__ __forceinline__ RowWords row_words(const Lds &s) {
  const luint *p = s.rw + mytid() * KRW;
  const uint4_t v = *(const luint4 *)p;
  const uint2_t v2 = *(const luint2 *)(p + 4);
  RowWords r;
  r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z; r.w[3] = v.w; r.w[4] = v2.x; r.w[5] = v2.y;
  return r;
}
// sum over the lane's entries of value * v[operand]; the value is read through `val(byte offset)` so that the scaling
// passes can look at |value| with the same walk
// finish(col, sum over the column of A of value * v[row]) on one lane per column
// All LDS reads of a phase are issued before the first one is consumed (values and operands into arrays first, the
// arithmetic in a second loop, in entry order): one round trip to LDS per phase instead of one per entry.
// pre(index) reads what finish() will need about the column / row (every lane: the reads join the batch above, clamped
// index for lanes without one); finish(index, sum, what pre returned) runs on one lane per column / row.
template <typename PRE, typename G>
__device__ __forceinline__ void col_dot(const Lds &s, const ldouble *v, int n, PRE pre, G finish) {
  const ColWords c = col_words(s);
  const int t = mytid(), j = t >> 2;
  double av[KT], ov[KT];
#pragma unroll
  for (int e = 0; e < KT; e++) { av[e] = lds_at(s.Av, c.w[e] >> 16); ov[e] = lds_at(v, c.w[e] & 0xFFFFu); }
  const auto ops = pre(j < n ? j : 0);
  double a = 0.0;
#pragma unroll
  for (int e = 0; e < KT; e++) a += av[e] * ov[e];
  a += quad_xor<2>(a);
  a += quad_xor<1>(a);
  if ((t & 3) == 0 && j < n) finish(j, a, ops);
}
template <typename PRE, typename G>
__device__ __forceinline__ void row_dot(const Lds &s, const ldouble *v, int m, PRE pre, G finish) {
  const RowWords r = row_words(s);
  const int t = mytid(), i = t >> 1;
  double av[KR], ov[KR];
#pragma unroll
  for (int e = 0; e < KR; e++) { av[e] = lds_at(s.Av, r.w[e] >> 16); ov[e] = lds_at(v, r.w[e] & 0xFFFFu); }
  const auto ops = pre(i < m ? i : 0);
  double a = 0.0;
#pragma unroll
  for (int e = 0; e < KR; e++) a += av[e] * ov[e];
  a += quad_xor<1>(a);
  if ((t & 1) == 0 && i < m) finish(i, a, ops);
}
// a wave-uniform double the optimiser cannot see through: what is derived from it (1 - alpha ...) is recomputed where it is
// used instead of being kept in -- or spilled from -- a register across the ADMM loop
__device__ __forceinline__ double opaque_s(double v) {
  asm volatile("" : "+s"(v));
  return v;
}
struct Ops2 { double a, b; };
struct Ops6 { double a, b, c, d, e, f; };

#ifdef OQ_BATCH_PROFILE
#define PROF_DECL long long pt0 = clock64(), pacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define PROF(k) { long long pt1 = clock64(); pacc[k] += pt1 - pt0; pt0 = pt1; }
#define PROF_PRINT if (inst == 0 && tid == 0) printf("cycles: load %lld scale %lld assemble %lld invert %lld rhs %lld solve %lld mulA+upd %lld check %lld rho %lld iters %d\n", pacc[0], pacc[1], pacc[8], pacc[2], pacc[3], pacc[4], pacc[5], pacc[6], pacc[7], iter);
#else
#define PROF_DECL
#define PROF(k)
#define PROF_PRINT
#endif

// ---------------------------------------------------------------------------------------------------------
// Residual evaluation + termination tests, every `check_termination` iterations.  NOT inlined on purpose: the ADMM
// loop holds the inverse in registers; as a separate function this phase gets its own register allocation (the call
// saves / restores what is live around it -- once per 25 iterations) instead of dragging 100+ temporaries into the
// allocation of the hot loop.  Everything it needs is in LDS (products walk the LDS copy of A); results go back through
// the nrm[] block: the 14 norms (Slot order of the large-problem path), then pri_res, dua_res, obj and the status code
// (0: keep iterating).
// ---------------------------------------------------------------------------------------------------------
enum { N_PRI = 14, N_DUA = 15, N_OBJ = 16, N_STATUS = 17, N_COUNT = 24 };
struct CheckArgs {
  int n, m, nnzA, nnzF, swapped, uns, passes, last, words;
  double ea, er, epi, edi, c, cinv;
};
extern __shared__ __attribute__((aligned(16))) double lds_raw[];

// K values per thread -> K block results in out[0..K) (max for op 0, sum for op 1); two barriers, a few registers
template <int K>
__device__ __forceinline__ void block_reduce_to(double *v, int op, ldouble *red, ldouble *out) {
#pragma unroll
  for (int k = 0; k < K; k++) {
    double a = v[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { double b = __shfl_xor(a, o, 64); a = op ? a + b : nmax(a, b); }
    v[k] = a;
  }
  __syncthreads();
  const int t = mytid();
  if ((t & 63) == 0) {
#pragma unroll
    for (int k = 0; k < K; k++) red[(t >> 6) * K + k] = v[k];
  }
  __syncthreads();
  if (t < K) {
    double a = red[t];
#pragma unroll 1
    for (int w = 1; w < NW; w++) a = op ? a + red[w * K + t] : nmax(a, red[w * K + t]);
    out[t] = a;
  }
  __syncthreads();
}

template <int CN, int CM, int CA, int CF>
__device__ __noinline__ void residual_phase(CheckArgs a) {
  Pattern P;
  P.n = CN ? CN : a.n; P.m = CN ? CM : a.m; P.nnzA = CN ? CA : a.nnzA; P.nnzF = CN ? CF : a.nnzF;
  const Lds s = carve((ldouble *)lds_raw, P, a.words != 0);
  const int n = P.n, m = P.m, tid = mytid();
  ldouble *x = a.swapped ? s.xp : s.x, *z = a.swapped ? s.zp : s.z;
  ldouble *nrm = s.nrm, *tmp = s.nrm + 18;  // tmp: 6 scratch results of the small reductions
  const bool uns = a.uns;
  const double c = a.c, cinv = a.cinv;
  auto a_rows = [&](const ldouble *v, auto finish) {
    rows_dot<2>(m, s.Rp, [&](int q) { return s.Av[s.Rmap[q]] * v[s.Rc[q]]; }, finish);
  };
  auto a_cols = [&](const ldouble *v, auto finish) {
    rows_dot<4>(n, s.Ap, [&](int k) { return s.Av[k] * v[s.Ai[k]]; }, finish);
  };
  // ---- residual evaluation (K8): nrm[0..14), pri_res, dua_res, obj ----
  {
    a_rows(x, [&](int r, double v) { s.Ax[r] = v; });
    mul_P(P, s, x, s.Px);
    a_cols(s.y, [&](int j, double v) { s.Aty[j] = v; });
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
    block_reduce_to<14>(v, 0, s.red, nrm);
    block_reduce_to<2>(sm, 1, s.red, tmp);
    if (tid == 0) {
      nrm[N_PRI] = m == 0 ? 0.0 : (uns ? nrm[1] : nrm[0]);
      nrm[N_DUA] = uns ? cinv * nrm[7] : nrm[6];
      nrm[N_OBJ] = cinv * (0.5 * tmp[0] + tmp[1]);
      nrm[N_STATUS] = 0.0;
    }
    __syncthreads();
  }
  const double pri_res = nrm[N_PRI], dua_res = nrm[N_DUA];
  // ---- termination tests: the requested accuracy when a check is due or the iteration limit is reached,
  //      then -- at the limit only -- the 10x-relaxed ones ----
  int status = 0;
  for (int pass = 0; pass < a.passes && status == 0; pass++) {
    const bool approx = pass == 1;
    double ea = a.ea, er = a.er, epi = a.epi, edi = a.edi;
    if (!(pri_res <= OSQP_INFTY) || !(dua_res <= OSQP_INFTY)) { status = OSQP_NON_CVX; break; }
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
        block_reduce_to<1>(v, 0, s.red, tmp);
        block_reduce_to<1>(sm, 1, s.red, tmp + 1);
        const double nv = tmp[0], lhs = tmp[1];
        if (nv > epi && lhs < -epi * nv) {
          a_cols(s.dy, [&](int j, double t) { s.tn[j] = t; });
          __syncthreads();
          double w[1] = {0.0};
          for (int j = tid; j < n; j += NT) w[0] = nmax(w[0], fabs(uns ? s.tn[j] / s.D[j] : s.tn[j]));
          block_reduce_to<1>(w, 0, s.red, tmp + 2);
          pinf = tmp[2] < epi * nv;
        }
      }
    }
    double eps_d = ea + er * (uns ? cinv * nmax(nrm[11], nmax(nrm[12], nrm[13])) : nmax(nrm[8], nmax(nrm[9], nrm[10])));
    if (dua_res < eps_d) dc = true;
    else {  // dual infeasibility on delta_x
      double v[1] = {0.0}, sm[1] = {0.0};
      for (int j = tid; j < n; j += NT) { v[0] = nmax(v[0], fabs(uns ? s.D[j] * s.dx[j] : s.dx[j])); sm[0] += s.q[j] * s.dx[j]; }
      block_reduce_to<1>(v, 0, s.red, tmp + 3);
      block_reduce_to<1>(sm, 1, s.red, tmp + 4);
      const double nv = tmp[3], qdx = tmp[4];
      double cs = uns ? c : 1.0;
      if (nv > edi && qdx < -cs * edi * nv) {
        mul_P(P, s, s.dx, s.tn);
        __syncthreads();
        double w[1] = {0.0};
        for (int j = tid; j < n; j += NT) w[0] = nmax(w[0], fabs(uns ? s.tn[j] / s.D[j] : s.tn[j]));
        block_reduce_to<1>(w, 0, s.red, tmp + 5);
        if (tmp[5] < cs * edi * nv) {
          a_rows(s.dx, [&](int r, double t) { s.tm[r] = t; });
          __syncthreads();
          double bad[1] = {0.0};
          for (int i = tid; i < m; i += NT) {
            double t = uns ? s.tm[i] / s.E[i] : s.tm[i];
            if ((s.u[i] < B_INF && t > edi * nv) || (s.l[i] > -B_INF && t < -edi * nv) || t != t) bad[0] = 1.0;
          }
          block_reduce_to<1>(bad, 0, s.red, tmp + 5);
          dinf = tmp[5] == 0.0;
        }
      }
    }
    if (pc && dc) status = approx ? OSQP_SOLVED_INACCURATE : OSQP_SOLVED;
    else if (pinf) status = approx ? OSQP_PRIMAL_INFEASIBLE_INACCURATE : OSQP_PRIMAL_INFEASIBLE;
    else if (dinf) status = approx ? OSQP_DUAL_INFEASIBLE_INACCURATE : OSQP_DUAL_INFEASIBLE;
  }
  if (a.last && a.passes && status == 0) status = OSQP_MAX_ITER_REACHED;
  __syncthreads();
  if (tid == 0) nrm[N_STATUS] = (double)status;
  __syncthreads();
}


#include "batch_quad.hpp"

// CN > 0: the instance shape (n, m, nnz(A), nnz(P full)) = (CN, CM, CA, CF) is known at compile time -- every LDS address
// becomes an immediate and every vector loop a fixed trip count (the registers otherwise spent on ~35 LDS pointers are
// what the inverse needs); CN = 0: the same source with the shape read from the pattern at run time.
template <int NCT, int CN, int CM, int CA, int CF>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_batch_solve(Pattern Pin, OSQPSettings st, int count, double *__restrict__ scratch_all,
                                                    const double *__restrict__ Px_all,
                                                    const double *__restrict__ Ax_all, const double *__restrict__ q_all,
                                                    const double *__restrict__ l_all, const double *__restrict__ u_all,
                                                    double *__restrict__ x_out, double *__restrict__ y_out,
                                                    double *__restrict__ info_out, int x_stride, int y_stride, int info_stride,
                                                    int info_cols) {
  const int inst = blockIdx.x;
  if (inst >= count) return;
  Pattern P = Pin;
  if (CN > 0) { P.n = CN; P.m = CM; P.nnzA = CA; P.nnzF = CF; }
  constexpr bool EXACT = CN > 0 && PARTS * NCT == CN;
  const int n = P.n, m = P.m, tid = mytid();
  // the thread's share of the entries of A as packed words in LDS (both orientations; the positions do not change under
  // scaling, so the walks of the scaling passes use them too: 4 lanes per column, 2 per row)
  const bool regs = sparse_fits(P);
  Lds s = carve((ldouble *)lds_raw, P, regs);
  PROF_DECL
  // ---- stage the shared pattern (16-bit) and load the instance -----------------
  for (int k = tid; k <= n; k += NT) { s.Ap[k] = (unsigned short)P.Ap[k]; s.Fp[k] = (unsigned short)P.Fp[k]; }
  for (int k = tid; k <= m; k += NT) s.Rp[k] = (unsigned short)P.Rp[k];
  for (int k = tid; k < P.nnzA; k += NT) { s.Ai[k] = (unsigned short)P.Ai[k]; s.Rc[k] = (unsigned short)P.Rc[k]; s.Rmap[k] = (unsigned short)P.Rmap[k]; }
  for (int k = tid; k < P.nnzF; k += NT) s.Fc[k] = (unsigned short)P.Fc[k];
  for (int k = tid; k < P.nnzA; k += NT) s.Av[k] = Ax_all[(size_t)inst * P.nnzA + k];
  if (tid == 0) s.Av[P.nnzA] = 0.0;  // what padded entries of the packed words point at
  for (int k = tid; k < P.nnzF; k += NT) s.Pv[k] = Px_all[(size_t)inst * P.nnzP + P.Fmap[k]];
  for (int j = tid; j < n; j += NT) { s.q[j] = q_all[(size_t)inst * n + j]; s.D[j] = 1.0; s.x[j] = 0.0; s.xp[j] = 0.0; s.dx[j] = 0.0; }
  for (int i = tid; i < m; i += NT) {
    s.l[i] = fmax(l_all[(size_t)inst * m + i], -OSQP_INFTY); s.u[i] = fmin(u_all[(size_t)inst * m + i], OSQP_INFTY);
    s.E[i] = 1.0; s.z[i] = 0.0; s.y[i] = 0.0; s.zp[i] = 0.0; s.dy[i] = 0.0;
  }
  __syncthreads();
  PROF(0)
  // ---- K0: Ruiz equilibration + cost scaling --------------------------------
  if (regs) store_sparse(P, s);
  __syncthreads();
  double c = 1.0;
  for (int it = 0; it < st.scaling; it++) {
    if (regs) {
      const ColWords cwd = col_words(s);
      double mx = 0.0;
#pragma unroll
      for (int e = 0; e < KT; e++) mx = fmax(mx, fabs(lds_at(s.Av, cwd.w[e] >> 16)));  // padded entries: |0|
      mx = fmax(mx, quad_xor<2>(mx));
      mx = fmax(mx, quad_xor<1>(mx));
      if ((tid & 3) == 0 && (tid >> 2) < n) {
        const int j = tid >> 2;
        for (int q = s.Fp[j]; q < s.Fp[j + 1]; q++) mx = fmax(mx, fabs(s.Pv[q]));
        s.tn[j] = 1.0 / sqrt(lim(mx));
      }
      const RowWords rwd = row_words(s);
      double mr = 0.0;
#pragma unroll
      for (int e = 0; e < KR; e++) mr = fmax(mr, fabs(lds_at(s.Av, rwd.w[e] >> 16)));
      mr = fmax(mr, quad_xor<1>(mr));
      if ((tid & 1) == 0 && (tid >> 1) < m) s.tm[tid >> 1] = 1.0 / sqrt(lim(mr));
    } else {
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
    }
    __syncthreads();
    for (int r = tid; r < n; r += NT)
      for (int q = s.Fp[r]; q < s.Fp[r + 1]; q++) {
        int cc = s.Fc[q];
        int lo = cc < r ? cc : r, hi = cc < r ? r : cc;
        s.Pv[q] = (s.Pv[q] * s.tn[lo]) * s.tn[hi];
      }
    if (regs) {
      if ((tid >> 2) < n) {
        const double tj = s.tn[tid >> 2];
        const ColWords cwd = col_words(s);
        const unsigned padw = (unsigned)(P.nnzA * 8);
#pragma unroll
        for (int e = 0; e < KT; e++) {
          const unsigned vo = cwd.w[e] >> 16;
          if (vo != padw) *(ldouble *)((lchar *)s.Av + vo) = (lds_at(s.Av, vo) * lds_at(s.tm, cwd.w[e] & 0xFFFFu)) * tj;
        }
      }
      for (int j = tid; j < n; j += NT) { s.q[j] *= s.tn[j]; s.D[j] *= s.tn[j]; }
    } else {
    for (int j = tid; j < n; j += NT) {
      for (int k = s.Ap[j]; k < s.Ap[j + 1]; k++) s.Av[k] = (s.Av[k] * s.tm[s.Ai[k]]) * s.tn[j];
      s.q[j] *= s.tn[j];
      s.D[j] *= s.tn[j];
    }
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
    {  // the sum and the maximum in one exchange: wavefront reductions, one barrier, the two halves of the upper part of
       // s.red taken in turn so that the pass after next may overwrite what this one reads
      double sm = w[0], mq = v[0];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { sm += __shfl_xor(sm, o, 64); mq = nmax(mq, __shfl_xor(mq, o, 64)); }
      ldouble *rr = s.red + 8 * NW + (it & 1) * 2 * NW;
      if ((tid & 63) == 0) { rr[2 * (tid >> 6)] = sm; rr[2 * (tid >> 6) + 1] = mq; }
      __syncthreads();
      sm = rr[0]; mq = rr[1];
#pragma unroll
      for (int wv = 1; wv < NW; wv++) { sm += rr[2 * wv]; mq = nmax(mq, rr[2 * wv + 1]); }
      w[0] = sm; v[0] = mq;
    }
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
  double rho = uni(fmin(fmax(st.rho, B_RHO_MIN), B_RHO_MAX));
  set_rho(P, s, rho, true);
  int status = OSQP_UNSOLVED;
  double *scratch = scratch_all + (size_t)inst * n * n;
  MTile<NCT> Minv;
  // y = A v / y = A' v through whichever walk of A applies
  auto a_rows = [&](const ldouble *v, auto pre, auto finish) {
    if (regs) row_dot(s, v, m, pre, finish);
    else rows_dot<2>(m, s.Rp, [&](int q) { return s.Av[s.Rmap[q]] * v[s.Rc[q]]; }, [&](int r, double a) { finish(r, a, pre(r)); });
  };
  auto a_cols = [&](const ldouble *v, auto pre, auto finish) {
    if (regs) col_dot(s, v, n, pre, finish);
    else rows_dot<4>(n, s.Ap, [&](int k) { return s.Av[k] * v[s.Ai[k]]; }, [&](int r, double a) { finish(r, a, pre(r)); });
  };
  PROF(2)
  const bool uns = st.scaling && !st.scaled_termination;
  const int check = (int)st.check_termination;
  const int rho_interval = st.adaptive_rho ? (st.adaptive_rho_interval ? (int)st.adaptive_rho_interval : 100) : 0;
  const double alpha = st.alpha, sigma = st.sigma;
  double pri_res = 0.0, dua_res = 0.0, obj = 0.0;
  ldouble *nrm = s.nrm;  // the 14 norms of the last residual evaluation (the same in every thread: kept in LDS, not in registers)
  int iter = 0, rho_updates = 0;
  ldouble *x = s.x, *xp = s.xp, *z = s.z, *zp = s.zp;

  // Every phase below appears ONCE in the code (the loop is arranged around that): the kernel is one long function
  // whose register allocation has to hold the inverse (2 NCT registers) across all of it.
  // ---- ADMM loop --------------------------------------------------------------
  const int max_iter = (int)st.max_iter;
  bool need_factor = true;
  for (int i = tid; i < m; i += NT) s.zt[i] = s.rho[i] * z[i] - s.y[i];
  __syncthreads();
  for (iter = 1; iter <= max_iter; iter++) {
    if (need_factor) {  // first iteration and after every rho update
      assemble_scratch(P, s, st.sigma, scratch);
      PROF(8)
      const bool pd = invert_mfma<(NCT <= 16 ? 2 : (NCT <= 25 ? 4 : 5))>(n, scratch, s.gjc);
      load_tile<NCT, EXACT>(n, scratch, Minv);
      if (!pd) { status = OSQP_NON_CVX; iter--; break; }
      need_factor = false;
      PROF(2)
    }
    { ldouble *t = x; x = xp; xp = t; t = z; z = zp; zp = t; }
    // b = sigma x_prev - q + A'(rho z_prev - y); s.zt = rho z_prev - y was left behind by the previous z / y update
    a_cols(s.zt, [&](int j) { return Ops2{xp[j], s.q[j]}; }, [&](int j, double a, const Ops2 &o) { s.bb[j] = sigma * o.a - o.b + a; });
    __syncthreads();
    PROF(3)
    // x~ = M^-1 b: the thread's tile of the inverse against its stretch of b, the four parts of a row add up inside a
    // quad of lanes; the lane that ends up with the row writes x~, x and delta_x
    apply_tile<NCT, EXACT>(n, Minv, s, opaque_s(alpha), xp, x);
    __syncthreads();
    PROF(4)
    // z~ = A x~ row by row, each row finished on the spot: z, y, delta_y and s.zt = rho z - y for the next right-hand side
    a_rows(s.xt, [&](int i) { return Ops6{zp[i], s.y[i], s.rhoi[i], s.l[i], s.u[i], s.rho[i]}; },
           [&](int i, double zt, const Ops6 &o) {
      const double al = opaque_s(alpha);
      const double zh = al * zt + (1.0 - al) * o.a;
      const double yo = o.b;
      const double zn = fmin(fmax(zh + o.c * yo, o.d), o.e);
      z[i] = zn;
      const double d = o.f * (zh - zn);
      s.dy[i] = d; s.y[i] = yo + d;
      s.zt[i] = o.f * zn - (yo + d);
    });
    __syncthreads();
    PROF(5)
    const bool last = iter == max_iter;
    const bool due = check && (iter % check == 0);
    const bool rho_due = rho_interval && (iter % rho_interval == 0);
    if (!(due || rho_due || last)) continue;

    // ---- residual evaluation (K8) and termination tests: a call, not inlined (see residual_phase) ----
    {
      CheckArgs ca;
      ca.n = n; ca.m = m; ca.nnzA = P.nnzA; ca.nnzF = P.nnzF;
      ca.swapped = (x != s.x); ca.uns = uns; ca.words = regs; ca.passes = (due || last) ? (last ? 2 : 1) : 0; ca.last = last;
      ca.ea = st.eps_abs; ca.er = st.eps_rel; ca.epi = st.eps_prim_inf; ca.edi = st.eps_dual_inf; ca.c = c; ca.cinv = cinv;
      residual_phase<CN, CM, CA, CF>(ca);
    }
    bool done = false;
    {
      const int code = (int)nrm[N_STATUS];
      if (code != 0) { status = code; done = true; }
      pri_res = uni(nrm[N_PRI]); dua_res = uni(nrm[N_DUA]); obj = uni(nrm[N_OBJ]);
    }
    PROF(6)
    if (done) break;
    // ---- adaptive rho ----
    if (rho_due) {
      double pr = m == 0 ? 0.0 : nrm[0] / (nmax(nrm[2], nrm[3]) + 1e-10);
      double du = nrm[6] / (nmax(nmax(nrm[8], nrm[9]), nrm[10]) + 1e-10);
      double est = uni(fmin(fmax(rho * sqrt(pr / (du + 1e-10)), B_RHO_MIN), B_RHO_MAX));
      if (est > rho * st.adaptive_rho_tolerance || est < rho / st.adaptive_rho_tolerance) {
        rho = est; rho_updates++;
        set_rho(P, s, rho, false);
        for (int i = mytid(); i < m; i += NT) s.zt[i] = s.rho[i] * z[i] - s.y[i];  // the carried vector follows rho
        __syncthreads();
        need_factor = true;  // picked up at the top of the next iteration
      }
    }
    PROF(7)
  }
  if (iter > max_iter) iter = max_iter;
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
__host__ __device__ constexpr int mpc_nnzA() {
  int c = 0;
  for (int t = 0; t < TT; t++) c += NX * (2 + (t + 1 < TT ? NX : 0)) + NU * (NX + 2 + (t + 1 < TT ? 1 : 0));
  return c;
}
constexpr int kMpcNnzA = mpc_nnzA();
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
  mutable DevBuf<double> scratch;  // [instances x n x n]: where a workgroup assembles its reduced KKT matrix (launch_batch sizes it)
  // schedule of the four-wavefront kernel (batch_quad.hpp); quad_ok: the pattern fits its compile-time bounds
  bool quad_ok = false;
  quad::Sched QS;
  DevBuf<unsigned short> qs_colstart, qs_collist;
  DevBuf<unsigned> qs_roww, qs_meta;
  DevBuf<unsigned long long> qs_stream;
  DevBuf<int> qs_Fp, qs_Fc, qs_Fmap;   // the full symmetric P in the kernel's numbering of the variables
  DevBuf<unsigned short> qs_perm;
  std::vector<int> hFmap;
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
    int max_col = 0, max_row = 0;
    for (int j = 0; j < n; j++) max_col = std::max(max_col, hAp[j + 1] - hAp[j]);
    for (int i = 0; i < m; i++) max_row = std::max(max_row, rp[i + 1] - rp[i]);
    P = Pattern{n, m, nnzA, nnzP, (int)fc.size(), Ap.get(), Ai.get(), Rp.get(), Rc.get(), Rmap.get(), Fp.get(), Fc.get(), Fmap.get(),
                (int)ti.size(), Tp.get(), Ti.get(), Tj.get(), Tr.get(), Ta.get(), Tb.get(), max_col, max_row};
    hFmap = fmap;
    build_quad(n, m, hAp, hAi, rp, rc, rmap, fp, fc, tp, ti, tj, tr, ta, tb, s);
  }

  // ---- schedule of the four-wavefront kernel ------------------------------------------------------------------------
  // The instantiations of the kernel (launch_batch holds the same table): quadrant size NH (n <= 2 NH), compile-time bounds
  // KC / KE of the longest column / row of A, CH rows per assembly window.  Entry 0 is the MPC family of BASELINE.json config
  // 5 with its shape compiled in (every LDS offset an immediate); the others take the shape at run time.  A pattern takes the
  // first entry it fits (round 5: up to round 4 entry 0 was the only one, every other pattern ran the 512-thread kernel with
  // its n x n global scratch).
  struct QuadCfg { int NH, KC, KE, CH; bool mpc; };
  static constexpr int kQuadCfgs = 11;
  static constexpr QuadCfg kQuadCfg[kQuadCfgs] = {{50, 9, 11, 16, true},   {16, 16, 16, 16, false}, {32, 16, 16, 16, false},
                                                  {48, 16, 16, 16, false}, {50, 16, 16, 16, false}, {64, 16, 16, 16, false},
                                                  {16, 32, 32, 16, false}, {32, 32, 32, 16, false}, {48, 32, 32, 16, false},
                                                  {64, 32, 32, 16, false}, {50, 12, 12, 16, false}};
  // the order patterns try the entries in: the MPC sizes with short columns / rows first look at the entry whose LDS layout
  // still holds THREE QPs per compute unit (bounds 12 / 12: 53 KB; 16 / 16 is 59 KB, two per unit)
  static constexpr int kQuadOrder[kQuadCfgs] = {0, 1, 2, 3, 10, 4, 5, 6, 7, 8, 9};
  int quad_cfg = -1;
  int kNH = 50, kKC = 9, kKE = 11, kCH = 16;  // of the entry taken
  void build_quad(int n, int m, const std::vector<int> &hAp, const std::vector<int> &hAi, const std::vector<int> &rp,
                  const std::vector<int> &rc, const std::vector<int> &rmap, const std::vector<int> &fp, const std::vector<int> &fc,
                  const std::vector<int> &tp, const std::vector<unsigned short> &ti, const std::vector<unsigned short> &tj,
                  const std::vector<unsigned short> &tr, const std::vector<unsigned short> &ta, const std::vector<unsigned short> &tb,
                  hipStream_t s) {
    quad_ok = false; quad_cfg = -1;
    const int only = getenv("OSQP_AMD_BATCH_QUAD_CFG") ? atoi(getenv("OSQP_AMD_BATCH_QUAD_CFG")) : -1;  // experiments: one entry by number
    for (int oi = 0; oi < kQuadCfgs && !quad_ok; oi++) {
      const int c = kQuadOrder[oi];
      if (only >= 0 && c != only) continue;
      const QuadCfg &q = kQuadCfg[c];
      if (q.mpc && !(n == MPC_N && m == MPC_M && hAp[n] == kMpcNnzA && (int)fc.size() == MPC_N)) continue;
      kNH = q.NH; kKC = q.KC; kKE = q.KE; kCH = q.CH;
      build_quad_with(n, m, hAp, hAi, rp, rc, rmap, fp, fc, tp, ti, tj, tr, ta, tb, s);
      if (quad_ok) quad_cfg = c;
    }
  }
  void build_quad_with(int n, int m, const std::vector<int> &hAp, const std::vector<int> &hAi, const std::vector<int> &rp,
                       const std::vector<int> &rc, const std::vector<int> &rmap, const std::vector<int> &fp, const std::vector<int> &fc,
                       const std::vector<int> &tp, const std::vector<unsigned short> &ti, const std::vector<unsigned short> &tj,
                       const std::vector<unsigned short> &tr, const std::vector<unsigned short> &ta, const std::vector<unsigned short> &tb,
                       hipStream_t s) {
    using namespace quad;
    quad_ok = false;
    const int nnzA = hAp[n], nnzF = (int)fc.size();
    static const bool trace = getenv("OSQP_AMD_BATCH_TRACE") && atoi(getenv("OSQP_AMD_BATCH_TRACE")) == 1;
    auto refuse = [&](const char *why, int a, int b) {
      if (trace) fprintf(stderr, "[batch] quadrants of %d (columns <= %d, rows <= %d): not taken, %s (%d > %d)\n", kNH, kKC, kKE, why, a, b);
    };
    if (n > 2 * kNH) { refuse("n", n, 2 * kNH); return; }
    if (m > QT || m == 0 || nnzA == 0) { refuse("rows", m, QT); return; }
    int kc = 0;
    for (int j = 0; j < n; j++) kc = std::max(kc, hAp[j + 1] - hAp[j]);
    // rows by length, longest first (stable): lane L holds row order[L], so the long rows share wavefronts
    std::vector<int> order(m);
    for (int i = 0; i < m; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return rp[a + 1] - rp[a] > rp[b + 1] - rp[b]; });
    const int ke = rp[order[0] + 1] - rp[order[0]];
    if (kc > kKC) { refuse("longest column", kc, kKC); return; }
    if (ke > kKE) { refuse("longest row", ke, kKE); return; }
    const Layout L = make_layout(n, m, nnzA, nnzF, kNH, kKC, kKE, kCH);
    // the term words and row words carry 16-bit LDS offsets of values, row records and operands: everything up to the
    // pattern tables (colstart, collist, meta, row words -- addressed with 32-bit arithmetic) must lie below 64 KB
    if (L.colstart > 65535 || (size_t)(nnzA + 1) * 8 > 65535) { refuse("LDS bytes below the pattern tables (16-bit offsets)", L.colstart, 65535); return; }
    if (L.total > 80 * 1024) { refuse("LDS bytes (two QPs per compute unit)", L.total, 80 * 1024); return; }
    const int kch = L.kch, kep = L.kep;
    // ---- the pattern of M = P + sigma I + A' rho A and the kernel's numbering of the variables ---------------------------
    // perm[j]: the caller's index of the kernel's variable j.  Row half 1 is numbered in reverse when that lets the first
    // phase of the sweeps (batch_quad.hpp: two-ended) take pivots from both ends of a banded pattern: the kernel's pivots
    // 0, 1, ... of half 0 stay inside the quadrant (0, 0), its pivots kNH, kNH + 1, ... of half 1 -- the caller's n - 1,
    // n - 2, ... -- inside (1, 1).  Counted by symbolic elimination: a pivot reaches at most what the pivots before it reached.
    std::vector<std::vector<int>> pair_of(n, std::vector<int>(n, -1)), pent0(n, std::vector<int>(n, -1));
    for (size_t t = 0; t < ti.size(); t++) { pair_of[ti[t]][tj[t]] = (int)t; pair_of[tj[t]][ti[t]] = (int)t; }
    for (int r = 0; r < n; r++) for (int q = fp[r]; q < fp[r + 1]; q++) pent0[r][fc[q]] = q;
    auto mnz = [&](int i, int j) { return i == j || pair_of[i][j] >= 0 || pent0[i][j] >= 0; };
    std::vector<int> perm(n), inv(n);
    int p1_top = 0, p1_bot = 0, bw = 0;
    {
      for (int i = 0; i < n; i++) perm[i] = i < kNH ? i : n - 1 - (i - kNH);
      auto count = [&](int first, int last, int &reach_over) {  // pivots first, first + 1, ... whose reach stays in [.., last]
        int reach = -1, cnt = 0;
        for (int a = first; a <= last; a++) {
          for (int j = last + 1; j < n; j++) if (mnz(perm[a], perm[j])) return cnt;      // (half 0 only: a column of half 1)
          for (int j = 0; j < first; j++) if (mnz(perm[a], perm[j])) return cnt;         // (half 1 only: a column of half 0)
          for (int j = a; j <= last; j++) if (mnz(perm[a], perm[j])) { reach = std::max(reach, j); reach_over = std::max(reach_over, j - a); }
          cnt = a - first + 1;
        }
        return cnt;
      };
      int over = 0;
      p1_top = count(0, std::min(n, kNH) - 1, over);
      if (n > kNH) p1_bot = count(kNH, n - 1, over);
      bw = over;  // the furthest a first-phase pivot reaches beyond itself
      const int nb = (kNH + 15) / 16, cap = nb > 1 ? (nb - 1) * 16 : kNH;  // the kernel compiles the phase for its first nb - 1 pivot blocks
      p1_top = std::min(p1_top, cap); p1_bot = std::min(p1_bot, cap);
      static const bool off = getenv("OSQP_AMD_BATCH_TWO_ENDED") && atoi(getenv("OSQP_AMD_BATCH_TWO_ENDED")) == 0;  // A/B runs
      const bool entry0 = kNH == kQuadCfg[0].NH && kKC == kQuadCfg[0].KC && kKE == kQuadCfg[0].KE;  // compiles the reach in: 19
      if (off || p1_top + p1_bot < 8 || (entry0 && bw > 19)) {
        p1_top = p1_bot = 0;
        for (int i = 0; i < n; i++) perm[i] = i;
      }
      for (int i = 0; i < n; i++) inv[perm[i]] = i;
      if (trace) fprintf(stderr, "[batch] pattern of M: first-phase pivots %d + %d of %d (reach %d)\n", p1_top, p1_bot, n, bw);
    }
    // the full symmetric P in the kernel's numbering (entries of a row in the caller's order)
    std::vector<int> fp2(n + 1, 0), fc2(nnzF), fmap2(nnzF);
    std::vector<std::vector<int>> pent(n, std::vector<int>(n, -1));
    for (int i = 0, k = 0; i < n; i++) {
      for (int q = fp[perm[i]]; q < fp[perm[i] + 1]; q++, k++) { fc2[k] = inv[fc[q]]; fmap2[k] = hFmap[q]; pent[i][fc2[k]] = k; }
      fp2[i + 1] = k;
    }
    std::vector<unsigned short> colstart(QT), collist((size_t)QT * kch);
    for (int t = 0; t < QT; t++) {
      const int wv = t >> 6, cbk = wv & 1, hb = wv >> 1, cl = t & 63, jk = cbk * kNH + cl;
      const bool col = cl < kNH && jk < n;
      const int j = col ? perm[jk] : 0;
      colstart[t] = (unsigned short)(col ? hAp[j] : nnzA);
      for (int e = 0; e < kch; e++) {
        const bool real = col && hAp[j] + 2 * e + hb < hAp[j + 1];
        collist[(size_t)t * kch + e] = (unsigned short)((real ? hAi[hAp[j] + 2 * e + hb] : m) * RECB);
      }
      collist[(size_t)t * kch + kch - 1] = colstart[t];  // the first value of the column rides in the last slot
    }
    std::vector<unsigned> roww((size_t)QT * kep, (unsigned)(nnzA * 8) << 16), meta((size_t)QT, 0xFFFFFFFFu);
    int kew[4] = {0, 0, 0, 0};
    for (int k = 0; k < m; k++) {
      const int row = order[k];
      meta[k] = (unsigned)row * RECB;
      kew[k >> 6] = std::max(kew[k >> 6], rp[row + 1] - rp[row]);
      for (int q = rp[row], e = 0; q < rp[row + 1]; q++, e++) roww[(size_t)k * kep + e] = ((unsigned)(rmap[q] * 8) << 16) | (unsigned)(inv[rc[q]] * 16);  // operands: 16 bytes per column
    }
    // terms of M, grouped by position (i, j) of a window (rows k kCH / 2 + r of either row half), the groups of a window
    // dealt to the threads (longest first)
    const int ch2 = kCH / 2, nwin = (kNH + ch2 - 1) / ch2;
    struct Term { unsigned short r, a, b; };
    std::vector<std::vector<std::vector<Term>>> groups(nwin);
    std::vector<std::vector<unsigned short>> targets(nwin);
    {
      for (int i = 0; i < n; i++) {
        const int wh = i / kNH, il = i - wh * kNH, cw = il / ch2, wrow = wh * ch2 + il % ch2;
        for (int j = 0; j < n; j++) {
          std::vector<Term> g;
          const int io = perm[i], jo = perm[j];
          if (pair_of[io][jo] >= 0) {
            const int t = pair_of[io][jo];
            for (int q = tp[t]; q < tp[t + 1]; q++)
              g.push_back(Term{(unsigned short)(L.rec + tr[q] * RECB + F_RHO), (unsigned short)(L.Av + 8 * ta[q]), (unsigned short)(L.Av + 8 * tb[q])});
          }
          if (pent[i][j] >= 0) g.push_back(Term{(unsigned short)L.cst, (unsigned short)L.cst, (unsigned short)(L.Pv + 8 * pent[i][j])});
          if (i == j) g.push_back(Term{(unsigned short)L.cst, (unsigned short)L.cst, (unsigned short)(L.cst + 8)});
          if (g.empty()) continue;
          groups[cw].push_back(g);
          targets[cw].push_back((unsigned short)(wrow * n + j));
        }
      }
    }
    std::vector<std::vector<std::vector<int>>> deal(nwin, std::vector<std::vector<int>>(QT));
    int ns = 0;
    for (int cw = 0; cw < nwin; cw++) {
      std::vector<int> idx(groups[cw].size());
      for (size_t g = 0; g < idx.size(); g++) idx[g] = (int)g;
      std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return groups[cw][a].size() > groups[cw][b].size(); });
      std::vector<int> load(QT, 0);
      for (int g : idx) {
        int best = 0;
        for (int t = 1; t < QT; t++) if (load[t] < load[best]) best = t;
        deal[cw][best].push_back(g);
        load[best] += (int)groups[cw][g].size();
      }
      for (int t = 0; t < QT; t++) ns = std::max(ns, load[t]);
    }
    ns = std::max(4, (ns + 3) & ~3);
    if (ns > 64) { refuse("assembly terms per thread and window", ns, 64); return; }  // (the kernel prefetches 12 slots -- its NSM -- and reads further ones in place)
    if (trace) fprintf(stderr, "[batch] quadrants of %d (columns <= %d, rows <= %d): taken, %d bytes of LDS, %d term slots\n", kNH, kKC, kKE, L.total, ns);
    const unsigned long long pad = (unsigned long long)(kCH * n) | 0x8000ull | ((unsigned long long)(L.rec + m * RECB + F_RHO) << 16) |
                                   ((unsigned long long)L.cst << 32) | ((unsigned long long)L.cst << 48);  // 0 * 1 * 1 into the spare position
    std::vector<unsigned long long> stream((size_t)nwin * ns * QT, pad);
    for (int cw = 0; cw < nwin; cw++)
      for (int t = 0; t < QT; t++) {
        int slot = 0;
        for (int g : deal[cw][t]) {
          const auto &G = groups[cw][g];
          for (size_t k = 0; k < G.size(); k++, slot++) {
            unsigned long long w = (unsigned long long)targets[cw][g] | ((unsigned long long)G[k].r << 16) | ((unsigned long long)G[k].a << 32) |
                                   ((unsigned long long)G[k].b << 48);
            if (k + 1 == G.size()) w |= 0x8000ull;
            stream[((size_t)cw * ns + slot) * QT + t] = w;
          }
        }
      }
    auto up16 = [&](DevBuf<unsigned short> &d, const std::vector<unsigned short> &h) { d.alloc(h.size()); d.upload(h.data(), h.size(), s); };
    auto up32 = [&](DevBuf<unsigned> &d, const std::vector<unsigned> &h) { d.alloc(h.size()); d.upload(h.data(), h.size(), s); };
    up16(qs_colstart, colstart); up16(qs_collist, collist); up32(qs_roww, roww); up32(qs_meta, meta);
    qs_stream.alloc(stream.size()); qs_stream.upload(stream.data(), stream.size(), s);
    HIP_CHECK(hipStreamSynchronize(s));
    std::vector<unsigned short> perm16(perm.begin(), perm.end());
    up16(qs_perm, perm16);
    auto upi = [&](DevBuf<int> &d, const std::vector<int> &h) { d.alloc(h.size()); d.upload(h.data(), h.size(), s); };
    upi(qs_Fp, fp2); upi(qs_Fc, fc2); upi(qs_Fmap, fmap2);
    HIP_CHECK(hipStreamSynchronize(s));
    QS = Sched{n, m, nnzA, P.nnzP, nnzF, {kew[0], kew[1], kew[2], kew[3]}, ns, p1_top, p1_bot, bw, getenv("OSQP_AMD_BATCH_ROT") ? atoi(getenv("OSQP_AMD_BATCH_ROT")) : 0, qs_perm.get(), qs_colstart.get(),
               qs_collist.get(), qs_roww.get(), qs_meta.get(), qs_stream.get(), qs_Fp.get(), qs_Fc.get(), qs_Fmap.get()};
    quad_ok = true;
  }
};

// outputs: row i of x / y / info at x + i * x_stride etc. (packed layouts put all three in one row); info_cols 4 or 6
void launch_batch(const DevicePattern &dp, const OSQPSettings &st, int count, const double *Px, const double *Ax, const double *q,
                  const double *l, const double *u, double *x, double *y, double *info, int x_stride, int y_stride, int info_stride,
                  int info_cols, hipStream_t s) {
  const Pattern &P = dp.P;
  size_t bytes = lds_bytes(P.n, P.m, P.nnzA, P.nnzF, sparse_fits(P));
  if (P.n > 128 || P.m > 65535 || P.nnzA > 65535 || P.nnzF > 65535) throw Error(1, "the batched path supports n <= 128 and fewer than 65536 rows / non-zeros");
  if (bytes > 160 * 1024) throw Error(1, "instance too large for the LDS-resident batched path (needs " + std::to_string(bytes) + " bytes of LDS)");
  const bool quad_path = dp.quad_ok && dp.quad_cfg >= 0 && batch_quad_enabled();
  const size_t need = quad_path ? 0 : (size_t)count * P.n * P.n;  // the four-wavefront kernel has no global scratch
  if (dp.scratch.n < need) { HIP_CHECK(hipStreamSynchronize(s)); dp.scratch.alloc(need); }
  const int nc = (P.n + PARTS - 1) / PARTS;  // columns of the inverse per thread: the register tile is sized at compile time
#define OQ_BATCH_LAUNCH(...)                                                                                                          \
  do {                                                                                                                                 \
    HIP_CHECK(hipFuncSetAttribute((const void *)k_batch_solve<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));  \
    OQ_LAUNCH((k_batch_solve<__VA_ARGS__>), dim3(count), dim3(NT), bytes, s, P, st, count, dp.scratch.get(), Px, Ax, q, l, u, x, y, info, \
              x_stride, y_stride, info_stride, info_cols);                                                                             \
  } while (0)
  // shapes compiled in (same source, constants folded): the MPC family of BASELINE.json config 5
  const bool mpc = P.n == MPC_N && P.m == MPC_M && P.nnzA == kMpcNnzA && P.nnzF == MPC_N;
  const bool use_quad = batch_quad_enabled();
  if (quad_path && use_quad) {
    // one QP per four wavefronts, three (two from 64-column quadrants on) QPs per compute unit, the factorisation on chip
    // (batch_quad.hpp); the instantiation the pattern's schedule was built for (DevicePattern::kQuadCfg)
    const quad::Layout L = quad::make_layout(P.n, P.m, P.nnzA, P.nnzF, dp.kNH, dp.kKC, dp.kKE, dp.kCH);
#define OQ_QUAD_LAUNCH(KERN, ...)                                                                                                      \
  do {                                                                                                                                 \
    auto kern = quad::KERN<__VA_ARGS__>;                                                                                               \
    HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, L.total));                           \
    OQ_LAUNCH(kern, dim3(count), dim3(quad::QT), (size_t)L.total, s, dp.QS, st, count, Px, Ax, q, l, u, x, y, info, x_stride, y_stride, \
              info_stride, info_cols);                                                                                                 \
  } while (0)
    g_batch_last_kernel = dp.quad_cfg;
    switch (dp.quad_cfg) {
    case 0: OQ_QUAD_LAUNCH(k_batch_quad, 50, 9, 11, 16, MPC_N, MPC_M, kMpcNnzA, MPC_N); break;
    case 1: OQ_QUAD_LAUNCH(k_batch_quad, 16, 16, 16, 16, 0, 0, 0, 0); break;
    case 2: OQ_QUAD_LAUNCH(k_batch_quad, 32, 16, 16, 16, 0, 0, 0, 0); break;
    case 3: OQ_QUAD_LAUNCH(k_batch_quad, 48, 16, 16, 16, 0, 0, 0, 0); break;
    case 4: OQ_QUAD_LAUNCH(k_batch_quad, 50, 16, 16, 16, 0, 0, 0, 0); break;   // the MPC sizes with another pattern (a non-diagonal P ...): no padding of the quadrants
    case 5: OQ_QUAD_LAUNCH(k_batch_quad2, 64, 16, 16, 16, 0, 0, 0, 0); break;
    case 6: OQ_QUAD_LAUNCH(k_batch_quad, 16, 32, 32, 16, 0, 0, 0, 0); break;
    case 7: OQ_QUAD_LAUNCH(k_batch_quad, 32, 32, 32, 16, 0, 0, 0, 0); break;
    case 8: OQ_QUAD_LAUNCH(k_batch_quad2, 48, 32, 32, 16, 0, 0, 0, 0); break;
    case 9: OQ_QUAD_LAUNCH(k_batch_quad2, 64, 32, 32, 16, 0, 0, 0, 0); break;
    default: OQ_QUAD_LAUNCH(k_batch_quad, 50, 12, 12, 16, 0, 0, 0, 0); break;
    }
#undef OQ_QUAD_LAUNCH
    return;
  }
  g_batch_last_kernel = -1;
  if (mpc) OQ_BATCH_LAUNCH(25, MPC_N, MPC_M, kMpcNnzA, MPC_N);
  else if (nc <= 16) OQ_BATCH_LAUNCH(16, 0, 0, 0, 0);
  else if (nc <= 25) OQ_BATCH_LAUNCH(25, 0, 0, 0, 0);
  else OQ_BATCH_LAUNCH(32, 0, 0, 0, 0);
#undef OQ_BATCH_LAUNCH
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

c_int osqp_amd_batch_last_kernel(void) { return g_batch_last_kernel; }

c_int osqp_amd_batch_solve(c_int count, c_int n, c_int m, const c_int *Pp, const c_int *Pi, const c_float *Px_all, const c_int *Ap,
                           const c_int *Ai, const c_float *Ax_all, const c_float *q_all, const c_float *l_all, const c_float *u_all,
                           const OSQPSettings *settings, c_float *x_out, c_float *y_out, OSQPInfo *info_out, c_int device) {
  try {
    // the same checks osqp_setup makes [REF src/interface.jl:47-100 + the C side's validate_data / validate_settings]
    if (count <= 0 || n <= 0 || m < 0 || !Pp || !Pi || !Ap || !Ai || !q_all || (m > 0 && (!l_all || !u_all)) || !settings || !x_out ||
        !info_out || (m > 0 && !y_out)) { set_last_error("invalid batch data"); return 1; }
    if (validate_settings(settings)) { set_last_error("invalid settings"); return 2; }
    if (n > 128 || m > 65535 || Pp[0] != 0 || Ap[0] != 0 || Pp[n] < 0 || Ap[n] < 0 || Pp[n] > 65535 || Ap[n] > 65535) {
      set_last_error("the batched path supports n <= 128 and fewer than 65536 rows / non-zeros"); return 1;
    }
    for (c_int j = 0; j < n; j++) {
      if (Pp[j + 1] < Pp[j] || Ap[j + 1] < Ap[j]) { set_last_error("column pointers must not decrease"); return 1; }
      for (c_int k = Pp[j]; k < Pp[j + 1]; k++) if (Pi[k] < 0 || Pi[k] > j) { set_last_error("P must be upper triangular with row indices in range"); return 1; }
      for (c_int k = Ap[j]; k < Ap[j + 1]; k++) if (Ai[k] < 0 || Ai[k] >= m) { set_last_error("row index of A out of range"); return 1; }
      // the term lists of A' rho A are built by merging sorted columns (DevicePattern::build): unsorted or repeated rows
      // would silently drop terms
      for (c_int k = Ap[j] + 1; k < Ap[j + 1]; k++) if (Ai[k] <= Ai[k - 1]) { set_last_error("the rows of every column of A must be sorted and unique"); return 1; }
      for (c_int k = Pp[j] + 1; k < Pp[j + 1]; k++) if (Pi[k] <= Pi[k - 1]) { set_last_error("the rows of every column of P must be sorted and unique"); return 1; }
    }
    if ((Pp[n] > 0 && !Px_all) || (Ap[n] > 0 && !Ax_all)) { set_last_error("invalid batch data"); return 1; }
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
