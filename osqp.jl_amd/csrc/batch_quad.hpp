// batch_quad.hpp -- the batched small-QP kernel, round 4: ONE QP PER FOUR WAVEFRONTS, THE INVERSE IN REGISTERS AS FOUR
// QUADRANTS (row K11 of SURVEY.md section 8a).  Included by batch.hip inside namespace oq::{anonymous} (it uses the LDS
// pointer types and the small helpers defined there).
//
// Why a second decomposition (the 512-thread kernel in batch.hip stays for the patterns this one does not take): with one
// QP per 512 threads every phase of an ADMM iteration gives a wavefront ~20 useful multiply-adds between two
// eight-wavefront barriers, two QPs fit a compute unit, and the factorisation goes through an n x n scratch in global
// memory (5.5 GB of HBM / L2 traffic per 4096 QPs).  Here a QP is a 256-thread workgroup, THREE are resident per compute
// unit (<= 53 KB of LDS, 168 vector registers: three wavefronts per SIMD), nothing of a factorisation leaves the chip.
//
//   * wavefront w = 2 hb + cb holds the quadrant (row half hb, column half cb) of the inverse of the reduced KKT matrix
//     M = P + sigma I + A' diag(rho) A: lane c owns column j = cb NH + c, register r of the lane is row hb NH + r (U[0..NH),
//     2 NH vector registers; NH = 50 for n = 100);
//   * x~ = M^-1 b: NH `v_fmac_f64_dpp ... row_newbcast` instructions per lane -- b sits in NH / 16 registers, lane l holding
//     b[hb NH + 16 k + (l & 15)], and the DPP control broadcasts element i & 15 of the sixteen-lane row to the row: no LDS
//     and no scalar-register traffic per multiply-add (the DPP form issues at the rate of the plain v_fma_f64; a
//     ds_read_b128 broadcast per two multiply-adds measures 6x slower: tools/micro/dpp_probe.hip); the two row halves of a
//     column leave two partial sums, added by whoever reads x~;
//   * the inverse is formed IN THE REGISTERS by symmetric Gauss-Jordan sweeps (Goodnight's sweep operator), one pivot per
//     step: the wavefronts of the pivot's row half publish, per lane, ONE number -- their column's element of the pivot row, a
//     register picked by a branch tree on the wave-uniform pivot index -- the pivot column is the pivot row by symmetry and
//     comes back through the same DPP broadcast, the rank-1 update is NH multiply-adds per lane; the scaling of the pivot's
//     own column is deferred as a per-column factor (the update is linear in a column, the factor carries through);
//   * M is assembled from a host-precomputed stream of terms through an LDS window of a few rows at a time (it aliases
//     the row-side pattern words, which are re-staged from L2 afterwards);
//   * the constraint rows live in LDS as 64-byte records {z, y | l, u | rho, 1 / rho | rho z - y, E}: the row finish of an
//     iteration is three 16-byte reads and two writes; lane L owns row order[L] (longest rows first, so that the lockstep
//     length of a wavefront is the longest row IT holds: 11 entries for the first wavefront of the MPC pattern, 2 for the
//     others); a column of A is walked by the two lanes of its variable, one parity each.
// Per iteration three four-wavefront barriers: partial b | partial x~ | z, y.  Same algorithm and the same arithmetic per
// element as the 512-thread kernel except for the order of the sums inside a sparse row / column and the route to the
// inverse.
#pragma once
#include <type_traits>
#include <utility>

namespace quad {

#ifdef OQ_BATCH_PROFILE  // experiment build (make prof): clock64() stamps per phase, printed by instance 0
#define QPROF_DECL long long qt0 = clock64(), qacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define QPROF(k) { long long qt1 = clock64(); qacc[k] += qt1 - qt0; qt0 = qt1; }
#define QPROF_PRINT if (inst == 0 && threadIdx.x == 0) printf("quad cycles: load %lld scale %lld assemble %lld invert %lld rhs %lld dense %lld rows %lld check %lld rho %lld iters %d\n", qacc[0], qacc[1], qacc[2], qacc[3], qacc[4], qacc[5], qacc[6], qacc[7], qacc[8], iter);
#else
#define QPROF_DECL
#define QPROF(k)
#define QPROF_PRINT
#endif

constexpr int QT = 256;    // threads per QP
constexpr int RECB = 64;   // bytes of a row record
enum { F_Z = 0, F_Y = 8, F_L = 16, F_U = 24, F_RHO = 32, F_RHOI = 40, F_ZT = 48, F_E = 56 };

struct Sched {  // device pointers, shared by all instances (built on the host from the shared pattern)
  int n, m, nnzA, nnzP, nnzF;
  int kew[4];                       // longest row held by each wavefront
  int ns;                           // term slots per thread and window (a multiple of 4)
  // two-ended sweeps (round 6): the first p1_top pivots (0, 1, ...) fill in only inside the quadrant (0, 0), the last p1_bot
  // (n - 1, n - 2, ...) only inside (1, 1) -- found on the host by symbolic elimination of the pattern of M; bw: its half
  // bandwidth (the MPC family: 19, two stages)
  int p1_top, p1_bot, bw;
  int rot;                          // > 0: the roles of the four wavefronts rotate with (workgroup id >> rot) & 3 (which SIMD carries the one-wavefront sweeps)
  const unsigned short *perm;       // [n]: the caller's index of the kernel's variable j (identity, or row half 1 reversed)
  const unsigned short *colstart;   // [QT]: first value of the lane's column (lanes without a column: nnzA)
  const unsigned short *collist;    // [QT][kch]: byte offset of the row record of entries hb, hb + 2, ... of the lane's column;
                                    // padding: the zero record (m)
  const unsigned *roww;             // [QT][kep]: (byte offset of the value) << 16 | 8 * column, entries of the lane's row
  const unsigned *meta;             // [QT]: byte offset of the record of the lane's row, 0xFFFFFFFF: none
  // the terms of M, window by window: stream[(window * ns + slot) * QT + thread] = target | last << 15 | r << 16 | a << 32 |
  // b << 48: acc += lds[r] * lds[a] * lds[b] (absolute LDS byte offsets: rho of a row record and two values of A -- or the
  // constant one twice and a value of P / sigma); `last` closes the sum of a position: window[target] = acc, acc = 0.
  // All terms of a position belong to one thread, in the order of the sparse dot product of the two columns.
  const unsigned long long *stream;
  const int *Fp, *Fc, *Fmap;        // full symmetric P, CSR; Fmap -> position in the caller's triu(P) values
};

struct Layout {  // byte offsets into the workgroup's LDS
  int Av, Pv, cst, vec, bp, xp, xs, tmp, cx, cq, cD, cdx, rec, dy, ax, red, nrm, ctype, Fp, Fc, colstart, collist, meta, roww, total;
  int nh2, kch, kep, pbstride;
};
// NH: rows / columns per quadrant; KC: longest column (entries), KE: longest row; CH: rows per assembly window
__host__ __device__ inline Layout make_layout(int n, int m, int nnzA, int nnzF, int NH, int KC, int KE, int CH) {
  Layout L;
  L.nh2 = 2 * ((NH + 15) & ~15);  // the broadcast registers of a row half cover 16 ceil(NH / 16) indices: both halves padded
  L.kch = (((KC + 1) / 2) + 1 + 3) & ~3;  // + the slot that carries the first value of the column
  L.kep = (KE + 3) & ~3;
  L.pbstride = L.nh2 + 2;
  int o = 0;
  L.Av = o; o += (nnzA + KC + 2) * 8;   // + the zero that padded row words point at, + what a padded column walk reads
  L.Pv = o; o += nnzF * 8;
  L.cst = o; o += 16;                   // the constants 1.0 and sigma (operands of the P / sigma terms of the assembly)
  o = (o + 15) & ~15;
  // two partial b (nh2 each, indexed hb NHP + r), two partial x~ (n each), a copy of x / delta_x (n), a column temporary (n);
  // the same stretch holds the pivot-row buffers of the inversion: 2 x (scaled, unscaled) x (nh2 + 2)
  const int vecd = (2 * L.nh2 + 4 * n) > 4 * L.pbstride ? (2 * L.nh2 + 4 * n) : 4 * L.pbstride;
  L.vec = o; L.bp = o; L.xp = o + 2 * L.nh2 * 8; L.xs = L.xp + 2 * n * 8; L.tmp = L.xs + n * 8; o += vecd * 8;
  L.cx = o; L.cq = o + n * 8; L.cD = o + 2 * n * 8; L.cdx = o + 3 * n * 8; o += 4 * n * 8;  // x, q, D, delta_x of the columns
  o = (o + 15) & ~15;
  L.rec = o; o += (m + 1) * RECB;       // + the zero record
  L.dy = o; o += m * 8;
  L.ax = o; o += m * 8;
  L.red = o; o += 2 * 4 * 8 * 8;        // two halves x four wavefronts x up to eight values
  L.nrm = o; o += 24 * 8;
  L.ctype = o; o += ((m + 3) & ~3);
  L.Fp = o; o += ((n + 1) * 2 + 3) & ~3;
  L.Fc = o; o += (nnzF * 2 + 3) & ~3;
  L.colstart = o; o += QT * 2;
  o = (o + 7) & ~7;
  L.collist = o; o += QT * L.kch * 2;
  L.meta = o; o += QT * 4;
  o = (o + 15) & ~15;
  const int words = QT * L.kep * 4, window = (CH * n + 1) * 8;  // + the position padded terms are written to
  L.roww = o; o += words > window ? words : window;
  L.total = o;
  return L;
}

// ---- small device helpers --------------------------------------------------------------------------------------
template <int LN>
__device__ __forceinline__ void fmac_bc(double &acc, double b, double m) {
  // acc += (b of lane LN of the sixteen-lane row) * m
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(m), "n"(LN));
}
__device__ __forceinline__ double ld(const lchar *base, unsigned off) { return *(const ldouble *)(base + off); }
__device__ __forceinline__ void sd(lchar *base, unsigned off, double v) { *(ldouble *)(base + off) = v; }
typedef double d2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) d2_t ld2;
__device__ __forceinline__ d2_t ld2at(const lchar *base, unsigned off) { return *(const ld2 *)(base + off); }
__device__ __forceinline__ void st2at(lchar *base, unsigned off, double a, double b) { d2_t v; v.x = a; v.y = b; *(ld2 *)(base + off) = v; }

// Wavefront reduction of a non-negative double (max: OP 0, NaN-propagating as nmax; sum: OP 1; the identity of both is 0,
// which is what a lane outside a DPP shift reads): row shifts by 1, 2, 4, 8 inside the sixteen-lane rows, then the row
// totals across rows (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3) -- plain VALU moves, where __shfl_xor
// goes through the LDS crossbar and its queue six times.  The total arrives in lane 63 and is returned as a wave-uniform
// value (v_readlane).
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_get(double v) {
  union { double d; int i[2]; } a, r;
  a.d = v;
  r.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], CTRL, ROWMASK, 0xF, true);
  r.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], CTRL, ROWMASK, 0xF, true);
  return r.d;
}
template <int OP>
__device__ __forceinline__ double red_op(double a, double b) { return OP ? a + b : nmax(a, b); }
template <int OP>
__device__ __forceinline__ double wave_red(double a) {
  a = red_op<OP>(a, dpp_get<0x111, 0xF>(a));  // row_shr:1
  a = red_op<OP>(a, dpp_get<0x112, 0xF>(a));  // row_shr:2
  a = red_op<OP>(a, dpp_get<0x114, 0xF>(a));  // row_shr:4
  a = red_op<OP>(a, dpp_get<0x118, 0xF>(a));  // row_shr:8
  a = red_op<OP>(a, dpp_get<0x142, 0xA>(a));  // row_bcast:15 into rows 1, 3
  a = red_op<OP>(a, dpp_get<0x143, 0xC>(a));  // row_bcast:31 into rows 2, 3
  union { double d; int i[2]; } u;
  u.d = a;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], 63);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], 63);
  return u.d;
}
// K per-thread values -> K workgroup results (wavefront 0's part first ... wavefront 3's last: a fixed order) in every
// thread; one barrier.  Consecutive calls alternate halves of `red` so that a fast wavefront cannot overwrite what a slow
// one still reads.  wvs: the wavefront's index (scalar).
template <int K, int OP>
__device__ __forceinline__ void quad_reduce(double *v, lchar *lds, int redoff, int &flip, int wvs) {
  const int base = redoff + flip * 4 * 8 * 8;
#pragma unroll
  for (int k = 0; k < K; k++) {
    const double r = wave_red<OP>(v[k]);
    sd(lds, base + (wvs * 8 + k) * 8, r);  // every lane the same value to the same address
    asm volatile("" ::: "memory");
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; k++) {
    double a = ld(lds, base + k * 8);
#pragma unroll
    for (int w = 1; w < 4; w++) { const double b = ld(lds, base + (w * 8 + k) * 8); a = OP ? a + b : nmax(a, b); }
    v[k] = a;
  }
  flip ^= 1;
}

// the same for one sum (v[0]) and one maximum (v[1]) in one exchange
__device__ __forceinline__ void quad_reduce2(double *v, lchar *lds, int redoff, int &flip, int wvs) {
  const int base = redoff + flip * 4 * 8 * 8;
  const double r0 = wave_red<1>(v[0]), r1 = wave_red<0>(v[1]);
  st2at(lds, base + wvs * 64, r0, r1);
  __syncthreads();
  d2_t a = ld2at(lds, base);
#pragma unroll
  for (int w = 1; w < 4; w++) { const d2_t b = ld2at(lds, base + w * 64); a.x += b.x; a.y = nmax(a.y, b.y); }
  v[0] = a.x; v[1] = a.y;
  flip ^= 1;
}

#define OQ_FENCE() asm volatile("" ::: "memory")
// a zero of its own: as a literal the second half of a 16-byte store shares ONE loop-invariant register tuple with every other
// zero of the kernel (the base of the 16-bit offset extractions among them), and under pressure that tuple is spilled and
// reloaded from scratch memory in every phase that needs a zero
__device__ __forceinline__ double fresh_zero() {
  double z;
  asm volatile("v_mov_b64 %0, 0" : "=v"(z));
  return z;
}

// register file of the lane's column: U[p] for a wave-uniform p, through a tree of scalar branches (a select chain would be
// two v_cndmask per register and access).  The asm at the END of a leaf keeps the optimiser from sinking the leaves' loads /
// stores into one access through a phi of addresses -- a dynamically indexed array lives in scratch memory.
template <int LO, int HI, int NR>
__device__ __forceinline__ double reg_get(const double (&U)[NR], int p) {
  if constexpr (HI - LO == 1) {
    double v = U[LO];
    asm volatile("" : "+v"(v));
    return v;
  } else {
    constexpr int MID = (LO + HI) / 2;
    if (p < MID) return reg_get<LO, MID, NR>(U, p);
    return reg_get<MID, HI, NR>(U, p);
  }
}
template <int LO, int HI, int NR>
__device__ __forceinline__ void reg_put(double (&U)[NR], int p, double v) {
  if constexpr (HI - LO == 1) {
    U[LO] = v;
    asm volatile("" : "+v"(U[LO]));
  } else {
    constexpr int MID = (LO + HI) / 2;
    if (p < MID) reg_put<LO, MID, NR>(U, p, v);
    else reg_put<MID, HI, NR>(U, p, v);
  }
}
// U[k * CH + r] = w[r] for a wave-uniform window index k
template <int LO, int HI, int NR, int CH>
__device__ __forceinline__ void chunk_store(double (&U)[NR], int k, const double (&w)[CH]) {
  if constexpr (HI - LO == 1) {
#pragma unroll
    for (int r = 0; r < CH; r++) {
      if constexpr (LO * CH + CH <= NR) U[LO * CH + r] = w[r];
      else if (LO * CH + r < NR) U[(LO * CH + r) < NR ? (LO * CH + r) : 0] = w[r];
    }
    asm volatile("" : "+v"(U[LO * CH]));
  } else {
    constexpr int MID = (LO + HI) / 2;
    if (k < MID) chunk_store<LO, MID, NR, CH>(U, k, w);
    else chunk_store<MID, HI, NR, CH>(U, k, w);
  }
}

// U[i] += b(i) * g over i in [I0, I1), b(i) = element i & 15 of the sixteen-lane row of bk
template <int I0, int I1, int NR>
__device__ __forceinline__ void rank1_16(double (&U)[NR], double bk, double g) {
  if constexpr (I0 < I1) {
    fmac_bc<I0 & 15>(U[I0], bk, g);
    rank1_16<I0 + 1, I1, NR>(U, bk, g);
  }
}
template <int I0, int I1, int NR, int NACC>
__device__ __forceinline__ void dot_16(const double (&U)[NR], double bk, double (&acc)[NACC]) {
  if constexpr (I0 < I1) {
    fmac_bc<I0 & 15>(acc[I0 % NACC], bk, U[I0]);
    dot_16<I0 + 1, I1, NR, NACC>(U, bk, acc);
  }
}
template <int K, int NB, int NR>
__device__ __forceinline__ void rank1_all(double (&U)[NR], const double (&B)[NB], double g) {
  if constexpr (K < NB) {
    rank1_16<K * 16, (K * 16 + 16 < NR ? K * 16 + 16 : NR), NR>(U, B[K], g);
    rank1_all<K + 1, NB, NR>(U, B, g);
  }
}
// U[i] += b(i) * g over i in [R0, R1) only (the rows a pivot of a banded array reaches)
template <int K, int R0, int R1, int NB, int NR>
__device__ __forceinline__ void rank1_range(double (&U)[NR], const double (&B)[NB], double g) {
  if constexpr (K < NB) {
    constexpr int A0 = K * 16 > R0 ? K * 16 : R0, A1 = K * 16 + 16 < R1 ? K * 16 + 16 : R1;
    if constexpr (A0 < A1) rank1_16<A0, A1, NR>(U, B[K], g);
    rank1_range<K + 1, R0, R1, NB, NR>(U, B, g);
  }
}
template <int K, int NB, int NR, int NACC>
__device__ __forceinline__ void dot_all(const double (&U)[NR], const double (&B)[NB], double (&acc)[NACC]) {
  if constexpr (K < NB) {
    dot_16<K * 16, (K * 16 + 16 < NR ? K * 16 + 16 : NR), NR, NACC>(U, B[K], acc);
    dot_all<K + 1, NB, NR, NACC>(U, B, acc);
  }
}

// v of lane 16 k + (l & 15) in every lane l: one pass through the LDS crossbar (ds_bpermute, no memory), where a write and a
// read back are two
__device__ __forceinline__ double row16_of(double v, int addr4) {
  union { double d; int i[2]; } a, r;
  a.d = v;
  r.i[0] = __builtin_amdgcn_ds_bpermute(addr4, a.i[0]);
  r.i[1] = __builtin_amdgcn_ds_bpermute(addr4, a.i[1]);
  return r.d;
}
template <int LN>
__device__ __forceinline__ double lane_of(double v) {  // v of lane LN, wave-uniform
  union { double d; int i[2]; } a, r;
  a.d = v;
  r.i[0] = __builtin_amdgcn_readlane(a.i[0], LN);
  r.i[1] = __builtin_amdgcn_readlane(a.i[1], LN);
  return r.d;
}

// One QP per 256-thread workgroup.  NH: rows / columns per quadrant (n <= 2 NH); KC, KE: compile-time bounds of the longest
// column / row of A; CH: rows per assembly window; CN > 0: the shape is compiled in (the MPC family), every LDS offset an
// immediate.
struct Me {  // what a lane is, recomputed at the head of every phase from an untraceable copy of the thread id (mytid()):
             // lane ids, LDS addresses and predicates cost a few instructions per phase instead of registers -- or spills --
             // across the ADMM loop, next to the inverse
  int t, lane16, wv, cb, hb, cl, j, jp;
  bool col, owner;
};
// lane id without a register that holds it: two mbcnt instructions wherever it is needed (the thread id the hardware leaves in
// v0 is otherwise a cold value the allocator spills -- and reloads from scratch memory once per phase)
__device__ __forceinline__ int lane_now() {
  int l;  // volatile asm: neither hoisted out of the ADMM loop nor merged across phases
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
template <int NH, bool FULL>
__device__ __forceinline__ Me make_me(int n, int wvs) {  // wvs: the wavefront's index in the workgroup (a scalar register)
  Me me;
  me.cl = lane_now(); me.wv = wvs; me.t = (wvs << 6) | me.cl; me.lane16 = me.cl & 15; me.cb = me.wv & 1; me.hb = me.wv >> 1;
  me.j = me.cb * NH + me.cl;                                      // the lane's column
  me.col = FULL ? me.cl < NH : (me.cl < NH && me.j < n);          // the lane holds a column
  me.owner = me.col && me.hb == 0;                                // ... and does the per-column work (x, q, D)
  me.jp = me.col ? me.j : -1;                                     // the column as a pivot index
  return me;
}

template <int NH, int KC, int KE, int CH, int CN, int CM, int CA, int CF>
__device__ __forceinline__ void quad_body(
    const Sched &S, const OSQPSettings &st, int count, const double *__restrict__ Px_all, const double *__restrict__ Ax_all,
    const double *__restrict__ q_all, const double *__restrict__ l_all, const double *__restrict__ u_all,
    double *__restrict__ x_out, double *__restrict__ y_out, double *__restrict__ info_out, int x_stride, int y_stride,
    int info_stride, int info_cols) {
  const int inst = blockIdx.x;
  if (inst >= count) return;
  const int n = CN ? CN : S.n, m = CN ? CM : S.m, nnzA = CN ? CA : S.nnzA, nnzF = CN ? CF : S.nnzF;
  const Layout L = make_layout(n, m, nnzA, nnzF, NH, KC, KE, CH);
  constexpr int NB = (NH + 15) / 16;       // broadcast registers of a row half
  constexpr int NHP = NB * 16;             // a row half, padded
  constexpr int KCE = (KC + 1) / 2;        // entries of a column per lane (one parity)
  constexpr int KCH = (KCE + 1 + 3) & ~3;  // the last slot of a lane's list carries the first value of its column
  constexpr int KEP = (KE + 3) & ~3;
  constexpr int NCH = (NH + CH / 2 - 1) / (CH / 2);  // assembly windows: CH / 2 rows of either row half each
  constexpr bool FULL = CN == 2 * NH;      // every lane below NH of a quadrant has a column
  lchar *lds = (lchar *)lds_raw;
  const int wvs = uni(((int)threadIdx.x >> 6) ^ (S.rot > 0 ? ((int)blockIdx.x >> S.rot) & 3 : 0));
  auto tid = [&]() { return (wvs << 6) | lane_now(); };
#define ME const Me me = make_me<NH, FULL>(n, wvs)
  const unsigned ZREC = (unsigned)m * RECB;  // the zero record
  int flip = 0;
  QPROF_DECL

  // ---- stage the schedule and load the instance -------------------------------------------------------------
  {
    ME;
    const int t = me.t;
    for (int k = t; k <= n; k += QT) *(lshort *)(lds + L.Fp + 2 * k) = (unsigned short)S.Fp[k];
    for (int k = t; k < nnzF; k += QT) *(lshort *)(lds + L.Fc + 2 * k) = (unsigned short)S.Fc[k];
    *(lshort *)(lds + L.colstart + 2 * t) = S.colstart[t];
#pragma unroll
    for (int e = 0; e < KCH; e++) *(lshort *)(lds + L.collist + (t * KCH + e) * 2) = S.collist[t * KCH + e];
    *(luint *)(lds + L.meta + t * 4) = S.meta[t];
    for (int k = t; k < nnzA; k += QT) sd(lds, L.Av + k * 8, Ax_all[(size_t)inst * nnzA + k]);
    for (int k = nnzA + t; k < nnzA + KC + 2; k += QT) sd(lds, L.Av + k * 8, 0.0);
    for (int k = t; k < nnzF; k += QT) sd(lds, L.Pv + k * 8, Px_all[(size_t)inst * S.nnzP + S.Fmap[k]]);
    for (int k = t; k < 2 * L.nh2 + 4 * n; k += QT) sd(lds, L.vec + k * 8, 0.0);
    for (int i = t; i <= m; i += QT) {
      const unsigned r = (unsigned)i * RECB;
      const bool real = i < m;
      st2at(lds, L.rec + r + F_Z, 0.0, 0.0);
      st2at(lds, L.rec + r + F_L, real ? fmax(l_all[(size_t)inst * m + i], -OSQP_INFTY) : 0.0, real ? fmin(u_all[(size_t)inst * m + i], OSQP_INFTY) : 0.0);
      st2at(lds, L.rec + r + F_RHO, 0.0, 0.0);
      st2at(lds, L.rec + r + F_ZT, 0.0, real ? 1.0 : 0.0);
      if (real) { sd(lds, L.dy + i * 8, 0.0); sd(lds, L.ax + i * 8, 0.0); }
    }
    // x_j, q_j, D_j, delta_x_j live in LDS, not in registers of the column's owner: they are touched once or twice per
    // iteration, and every register next to the inverse counts
    if (me.owner) { sd(lds, L.cx + me.j * 8, 0.0); sd(lds, L.cq + me.j * 8, q_all[(size_t)inst * n + S.perm[me.j]]); sd(lds, L.cD + me.j * 8, 1.0); sd(lds, L.cdx + me.j * 8, 0.0); }
  }
  auto stage_words = [&]() {
    const int t = tid();
#pragma unroll
    for (int e = 0; e < KEP; e++) *(luint *)(lds + L.roww + (t * KEP + e) * 4) = S.roww[t * KEP + e];
  };
  stage_words();
  __syncthreads();

  // ---- walks of the pattern -----------------------------------------------------------------------------------
  // the lane's half of column j of A (entries hb, hb + 2, ...): f(e, LDS offset of the value, record offset of its row)
  auto col_walk = [&](const Me &me, auto f) {
    unsigned short ro[KCH];
#pragma unroll
    for (int e4 = 0; e4 < KCH; e4 += 4) {
      const uint2_t w = *(const luint2 *)(lds + L.collist + (me.t * KCH + e4) * 2);
      ro[e4] = (unsigned short)(w.x & 0xFFFFu); ro[e4 + 1] = (unsigned short)(w.x >> 16);
      ro[e4 + 2] = (unsigned short)(w.y & 0xFFFFu); ro[e4 + 3] = (unsigned short)(w.y >> 16);
    }
    const unsigned vbase = L.Av + ((unsigned)ro[KCH - 1] + me.hb) * 8;
#pragma unroll
    for (int e = 0; e < KCE; e++) f(e, vbase + 16 * e, (unsigned)ro[e]);
  };
  // partial sum over the lane's half of column j of value * (field FIELD of the entry's row record)
  auto col_dot = [&](const Me &me, int field) -> double {
    double av[KCE], ov[KCE];
    col_walk(me, [&](int e, unsigned voff, unsigned ro) { av[e] = ld(lds, voff); ov[e] = ld(lds, L.rec + field + ro); });
    double a = av[0] * ov[0];
#pragma unroll
    for (int e = 1; e < KCE; e++) a = __builtin_fma(av[e], ov[e], a);
    return a;
  };
  // the lane's row: sum over its entries of value * (v0[col] + v1[col]), the two operands interleaved at xp (16 bytes per
  // column: the two partial sums of x~, or a vector and a zero); words padded with (zero value, operand 0).  All words
  // first, then the values and operands of four entries at a time: a wavefront of short rows (<= 4 entries) makes one
  // trip to LDS, the wavefront of the long rows three.
  auto row_batch = [&](const unsigned *w, auto nb_tag, double &a0, double &a1) {
    constexpr int NBE = decltype(nb_tag)::value;
    double av[NBE];
    d2_t op[NBE];
#pragma unroll
    for (int e = 0; e < NBE; e++) { av[e] = ld(lds, L.Av + (w[e] >> 16)); op[e] = ld2at(lds, L.xp + (w[e] & 0xFFFFu)); }
#pragma unroll
    for (int e = 0; e < NBE; e++) {
      const double o = op[e].x + op[e].y;
      if (e & 1) a1 = __builtin_fma(av[e], o, a1); else a0 = __builtin_fma(av[e], o, a0);
    }
  };
  // hook(): called once, before the loads of the last batch are issued -- where the caller's own LDS reads (the row record of
  // the finish) join the queue without adding to the registers the first batch holds
  auto row_dot_h = [&](const Me &me, auto hook) -> double {
    const int kew = uni(S.kew[me.wv]);
    double a0 = 0.0, a1 = 0.0;
    unsigned w[KEP];
    if (kew <= 4) {
      const uint4_t w4 = *(const luint4 *)(lds + L.roww + (me.t * KEP) * 4);
      w[0] = w4.x; w[1] = w4.y; w[2] = w4.z; w[3] = w4.w;
      hook();
      row_batch(w, std::integral_constant<int, 4>{}, a0, a1);
    } else {
      constexpr int NBT = KEP / 4;
      uint4_t w4 = *(const luint4 *)(lds + L.roww + (me.t * KEP) * 4);
#pragma unroll
      for (int bt = 0; bt < NBT; bt++) {
        const unsigned wb[4] = {w4.x, w4.y, w4.z, w4.w};
        if (bt + 1 < NBT) w4 = *(const luint4 *)(lds + L.roww + (me.t * KEP + 4 * bt + 4) * 4);  // the next batch's words ride along
        if (bt == NBT - 1) hook();
        if (bt * 4 < kew) row_batch(wb, std::integral_constant<int, 4>{}, a0, a1);
      }
    }
    return a0 + a1;
  };
  auto row_dot = [&](const Me &me) -> double { return row_dot_h(me, [] {}); };
  auto row_absmax = [&](const Me &me) -> double {
    const int kew = uni(S.kew[me.wv]);
    double mx = 0.0;
#pragma unroll
    for (int e0 = 0; e0 < KEP; e0 += 4) {
      if (e0 < kew) {
        const uint4_t w4 = *(const luint4 *)(lds + L.roww + (me.t * KEP + e0) * 4);
        const unsigned w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int e = 0; e < 4; e++) mx = fmax(mx, fabs(ld(lds, L.Av + (w[e] >> 16))));
      }
    }
    return mx;
  };
  auto my_rec = [&](const Me &me) -> unsigned { return *(const luint *)(lds + L.meta + me.t * 4); };  // 0xFFFFFFFF: no row
  // row j of the full symmetric P times a vector in LDS (owner lanes)
  auto p_row_dot = [&](const Me &me, int vecoff) -> double {
    double a = 0.0;
    if (me.owner) {
      const int q0 = *(const lshort *)(lds + L.Fp + 2 * me.j), q1 = *(const lshort *)(lds + L.Fp + 2 * me.j + 2);
      for (int q = q0; q < q1; q++) a += ld(lds, L.Pv + q * 8) * ld(lds, vecoff + 8 * *(const lshort *)(lds + L.Fc + 2 * q));
    }
    return a;
  };
  auto p_col_absmax = [&](const Me &me) -> double {
    double mx = 0.0;
    if (me.owner) {
      const int q0 = *(const lshort *)(lds + L.Fp + 2 * me.j), q1 = *(const lshort *)(lds + L.Fp + 2 * me.j + 2);
      for (int q = q0; q < q1; q++) mx = fmax(mx, fabs(ld(lds, L.Pv + q * 8)));
    }
    return mx;
  };

  QPROF(0)
  // ---- K0: Ruiz equilibration + cost scaling (the arithmetic of batch.hip / oracle scale_data, element for element) ----
  double c = 1.0;
  for (int it = 0; it < st.scaling; it++) {
    ME;
    const unsigned myrec = my_rec(me);
    const bool hasrow = myrec != 0xFFFFFFFFu;
    const int j = me.j;
    {  // column maxima: the two halves of a column meet in tmp
      double mx = 0.0;
      col_walk(me, [&](int e, unsigned voff, unsigned ro) { const double v = fabs(ld(lds, voff)); mx = fmax(mx, ro != ZREC ? v : 0.0); });
      if (me.col && me.hb == 1) sd(lds, L.tmp + j * 8, mx);
      if (hasrow) sd(lds, L.ax + (myrec / RECB) * 8, 1.0 / sqrt(lim(row_absmax(me))));  // row factors into ax (free outside an evaluation)
      __syncthreads();
      if (me.owner) {
        mx = fmax(fmax(mx, ld(lds, L.tmp + j * 8)), p_col_absmax(me));
        sd(lds, L.xs + j * 8, 1.0 / sqrt(lim(mx)));
      }
    }
    __syncthreads();
    const double tnj = me.col ? ld(lds, L.xs + j * 8) : 1.0;
    if (me.owner) {
      const int q0 = *(const lshort *)(lds + L.Fp + 2 * j), q1 = *(const lshort *)(lds + L.Fp + 2 * j + 2);
      for (int q = q0; q < q1; q++) {
        const int cc = *(const lshort *)(lds + L.Fc + 2 * q);
        const int lo = cc < j ? cc : j, hi = cc < j ? j : cc;
        sd(lds, L.Pv + q * 8, (ld(lds, L.Pv + q * 8) * ld(lds, L.xs + lo * 8)) * ld(lds, L.xs + hi * 8));
      }
      sd(lds, L.cq + j * 8, ld(lds, L.cq + j * 8) * tnj); sd(lds, L.cD + j * 8, ld(lds, L.cD + j * 8) * tnj);
    }
    if (me.col)
      col_walk(me, [&](int e, unsigned voff, unsigned ro) {
        if (ro != ZREC) sd(lds, voff, (ld(lds, voff) * ld(lds, L.ax + (ro / RECB) * 8)) * tnj);
      });
    if (hasrow) sd(lds, L.rec + myrec + F_E, ld(lds, L.rec + myrec + F_E) * ld(lds, L.ax + (myrec / RECB) * 8));
    __syncthreads();
    double sq[2] = {p_col_absmax(me), me.owner ? fabs(ld(lds, L.cq + j * 8)) : 0.0};  // sum of the column maxima of P, max |q|
    quad_reduce2(sq, lds, L.red, flip, wvs);
    double c_temp = sq[0] / (double)n;
    c_temp = lim(fmax(c_temp, lim(sq[1])));
    c_temp = uni(1.0 / c_temp);
    for (int k = me.t; k < nnzF; k += QT) sd(lds, L.Pv + k * 8, ld(lds, L.Pv + k * 8) * c_temp);
    if (me.owner) sd(lds, L.cq + j * 8, ld(lds, L.cq + j * 8) * c_temp);
    c = uni(c * c_temp);
    __syncthreads();
  }
  const double cinv = uni(1.0 / c);
  // ---- scaled bounds, K1: classes and rho ------------------------------------------------------------------------
  double rho = uni(fmin(fmax(st.rho, B_RHO_MIN), B_RHO_MAX));
  auto set_rho = [&](bool classify) {
    for (int i = tid(); i < m; i += QT) {
      const unsigned r = (unsigned)i * RECB;
      int ty;
      if (classify) {
        const double e = ld(lds, L.rec + r + F_E);
        const double lo = ld(lds, L.rec + r + F_L) * e, up = ld(lds, L.rec + r + F_U) * e;
        st2at(lds, L.rec + r + F_L, lo, up);
        if (lo < -B_INF && up > B_INF) ty = -1;
        else if (up - lo < 1e-4) ty = 1;
        else ty = 0;
        *(lchar *)(lds + L.ctype + i) = (char)ty;
      } else ty = *(const lchar *)(lds + L.ctype + i);
      const double rr = ty == -1 ? B_RHO_MIN : (ty == 1 ? 1e3 * rho : rho);
      st2at(lds, L.rec + r + F_RHO, rr, 1.0 / rr);
      sd(lds, L.rec + r + F_ZT, rr * ld(lds, L.rec + r + F_Z) - ld(lds, L.rec + r + F_Y));  // the carried vector follows rho
    }
    __syncthreads();
  };
  set_rho(true);

  QPROF(1)
  const bool uns = st.scaling && !st.scaled_termination;
  const int check = (int)st.check_termination;
  const int rho_interval = st.adaptive_rho ? (st.adaptive_rho_interval ? (int)st.adaptive_rho_interval : 100) : 0;
  const double sigma = st.sigma;
  const int max_iter = (int)st.max_iter;
  double pri_res = 0.0, dua_res = 0.0, obj = 0.0;
  int status = OSQP_UNSOLVED, iter = 0, rho_updates = 0;
  bool need_factor = true;
  double U[NH], nsc = 0.0;

  for (iter = 1; iter <= max_iter; iter++) {
    if (need_factor) {
      // ---- K2: M = P + sigma I + A' diag(rho) A through an LDS window, into the registers ------------------------------------
      // A window holds CH / 2 rows of EITHER row half (rows k CH/2 + r of half 0 and of half 1; all columns): every lane
      // reads -- and clears -- its own column of the rows of its half, registers k CH/2 + r.  The windows are unrolled:
      // compile-time register indices, every register assigned exactly once.  The window aliases the row words.
      for (int e = tid(); e < CH * n + 1; e += QT) sd(lds, L.roww + e * 8, 0.0);
      if (tid() == 0) st2at(lds, L.cst, 1.0, sigma);
      __syncthreads();
      // the term words of a window are fetched (L2) behind the terms of the window before, so that their latency runs under
      // the two barriers and the read-out of that window; NSM: compile-time bound of the slots per thread
      constexpr int NSM = 12, CH2 = CH / 2;
      const int NS = S.ns;
      unsigned long long tw[NSM];
      auto fetch_words = [&](int cw) {
        const unsigned long long *sp = S.stream + (size_t)cw * NS * QT + tid();
#pragma unroll
        for (int k = 0; k < NSM; k++) tw[k] = k < NS ? sp[(size_t)k * QT] : 0ull;
      };
      fetch_words(0);
      auto window = [&](auto k_tag) {
        constexpr int K = decltype(k_tag)::value;
        {
          double acc = 0.0;
          auto four = [&](unsigned long long w0, unsigned long long w1, unsigned long long w2, unsigned long long w3) {
            const unsigned long long w[4] = {w0, w1, w2, w3};
            double r[4], a[4], bq[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
              r[k] = ld(lds, (unsigned)(w[k] >> 16) & 0xFFFFu); a[k] = ld(lds, (unsigned)(w[k] >> 32) & 0xFFFFu); bq[k] = ld(lds, (unsigned)(w[k] >> 48));
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
              acc += r[k] * a[k] * bq[k];
              if (w[k] & 0x8000u) { sd(lds, L.roww + 8 * ((unsigned)w[k] & 0x7FFFu), acc); acc = 0.0; }
            }
          };
#pragma unroll
          for (int sl = 0; sl < NSM; sl += 4)
            if (sl < NS) four(tw[sl], tw[sl + 1], tw[sl + 2], tw[sl + 3]);
          // patterns whose longest sums exceed the prefetched slots (a column of 16 entries is a diagonal position of 18
          // terms): the words of the further slots straight from L2, four at a time
          if (NS > NSM) {
            const unsigned long long *sp = S.stream + (size_t)K * NS * QT + tid();
            for (int sl = NSM; sl < NS; sl += 4) four(sp[(size_t)sl * QT], sp[(size_t)(sl + 1) * QT], sp[(size_t)(sl + 2) * QT], sp[(size_t)(sl + 3) * QT]);
          }
        }
        if constexpr (K + 1 < NCH) fetch_words(K + 1);
        __syncthreads();
        {
          ME;
          [&]<int... RS>(std::integer_sequence<int, RS...>) {
            ([&] {
              if constexpr (K * CH2 + RS < NH) {
                const bool live = me.col && (FULL || me.hb * NH + K * CH2 + RS < n);
                const unsigned pos = L.roww + ((me.hb * CH2 + RS) * n + me.j) * 8;
                U[K * CH2 + RS] = live ? ld(lds, pos) : 0.0;
                if (live) sd(lds, pos, 0.0);
              }
            }(), ...);
          }(std::make_integer_sequence<int, CH2>{});
        }
        __syncthreads();
      };
      [&]<int... KS>(std::integer_sequence<int, KS...>) { (window(std::integral_constant<int, KS>{}), ...); }(std::make_integer_sequence<int, NCH>{});
      stage_words();
      QPROF(2)
      // ---- the inverse by symmetric sweeps, in the registers ---------------------------------------------------------
      // sc * U[r] = element (hb NH + r, j) of the swept array; result -M^-1.  Buffers of a step: scaled pivot row, then the
      // unscaled one (entries at the positions cb NHP + c of the padded halves), d and the pivot behind them.
      double sc = 1.0;
      bool pd = true;
      // ---- first phase, two-ended (round 6): while the pivots taken from the top of row half 0 (0, 1, ...) reach no row or
      // column beyond the quadrant (0, 0), and the pivots taken from the top of row half 1 (NH, NH + 1, ... -- the host numbers
      // the variables of that half in REVERSE, S.perm, so that these are the LAST variables of a banded pattern) none outside
      // (1, 1), the two sweeps touch disjoint registers of ONE wavefront each (pivots of a sweep commute; the swept array of
      // a banded M -- the MPC family: two stages, half bandwidth 19 -- stays zero outside the reach of the pivots taken so
      // far).  Wavefront 0 takes the pivots of half 0, wavefront 3 those of half 1, at the same time and through the same
      // code, each with its own pivot-row buffer and WITHOUT a workgroup barrier (the LDS operations of one wavefront
      // complete in order); the other two wait at the hand-over.  With the half bandwidth compiled in (BWC) a step also skips
      // the registers its pivot cannot reach.  S.p1_top / p1_bot: counted on the host by symbolic elimination of the
      // pattern of M (0: a pattern without such pivots, the phase is skipped).
      const int T1 = uni(S.p1_top), B1 = uni(S.p1_bot);
      if (T1 + B1 > 0) {
        constexpr int BWC = CN ? 19 : 0;  // the half bandwidth compiled in (the host leaves the phase out when the pattern's is larger)
        ME;
        const unsigned slot8 = (me.cb * NHP + me.cl) * 8, brow8 = (me.hb * NHP + me.lane16) * 8;
        const bool inpad = NHP == 64 || me.cl < NHP;
        // (the same shape as the four-wavefront step below -- publish under `hb == PH`, read back, update, put -- minus the
        // barrier: other arrangements of the same step left the allocator with copies of the sixteen candidate pivot
        // registers in every trip; the host caps the counts at the pivots of the first NB - 1 sixteen-pivot blocks)
        if constexpr (CN > 0) {
          // the compiled-in family: the steps unrolled, the pivot a compile-time register and lane.  No register-selection
          // trees (two per step, ~600 of its ~1200 cycles), no LDS: the pivot comes through v_readlane, the pivot row through
          // ds_bpermute (the crossbar alone), and a step stops at the last register its pivot can reach.  One code path for
          // both wavefronts (wavefront 0: pivots of half 0, wavefront 3: of half 1).
          const int cnt = me.wv == 0 ? T1 : (me.wv == 3 ? B1 : 0);
          const int a4 = me.lane16 * 4;
          constexpr int CAP = (NB > 1 ? NB - 1 : 1) * 16 < NH ? (NB > 1 ? NB - 1 : 1) * 16 : NH;
          auto step = [&](auto k_tag) {
            constexpr int K = decltype(k_tag)::value;
            constexpr int R1 = K + BWC + 1 < NH ? K + BWC + 1 : NH, NBK = (R1 + 15) / 16;
            if (K < cnt) {
              const double up = U[K], pv = sc * up;
              double B[NB];
#pragma unroll
              for (int k = 0; k < NB; k++) B[k] = k < NBK ? row16_of(pv, a4 + 64 * k) : 0.0;
              const double pvp = lane_of<K>(pv);
              double d = __builtin_amdgcn_rcp(pvp);
              d = __builtin_fma(__builtin_fma(-pvp, d, 1.0), d, d);
              d = __builtin_fma(__builtin_fma(-pvp, d, 1.0), d, d);
              if (!(pvp > 0.0)) pd = false;
              const bool mine = me.cl == K;
              const double g = mine ? 0.0 : -d * up;
              rank1_range<0, 0, R1, NB, NH>(U, B, g);
              U[K] = mine ? -1.0 : -g;
              sc = mine ? d : sc;
            }
          };
          [&]<int... IS>(std::integer_sequence<int, IS...>) { (step(std::integral_constant<int, IS>{}), ...); }(std::make_integer_sequence<int, CAP>{});
        } else {
        auto solo16 = [&](auto ph_tag, auto pb_tag) {
          constexpr int PH = decltype(ph_tag)::value, PB = decltype(pb_tag)::value, P1 = (PB + 1) * 16 < NH ? (PB + 1) * 16 : NH;
          constexpr int R1 = BWC > 0 && P1 + BWC < NH ? P1 + BWC : NH;
          const int cnt = me.wv == (PH ? 3 : 0) ? (PH ? B1 : T1) : 0;
          const unsigned pbw = L.vec + PH * 2 * L.pbstride * 8;
#pragma unroll 1
          for (int pr = PB * 16; pr < P1 && pr < cnt; pr++) {
            if (me.hb == PH) {
              const double up = reg_get<PB * 16, P1, NH>(U, pr);
              const double pv = sc * up;
              if (inpad) { sd(lds, pbw + slot8, pv); sd(lds, pbw + L.pbstride * 8 + slot8, up); }
              if (me.cl == pr) {
                double d = __builtin_amdgcn_rcp(pv);
                d = __builtin_fma(__builtin_fma(-pv, d, 1.0), d, d);
                d = __builtin_fma(__builtin_fma(-pv, d, 1.0), d, d);
                st2at(lds, pbw + L.nh2 * 8, d, pv);
              }
            }
            OQ_FENCE();
            double B[NB];
#pragma unroll
            for (int k = 0; k < NB; k++) B[k] = (k * 16 < R1) ? ld(lds, pbw + brow8 + 128 * k) : 0.0;
            const double up = inpad ? ld(lds, pbw + L.pbstride * 8 + slot8) : 0.0;
            const d2_t dp = ld2at(lds, pbw + L.nh2 * 8);
            OQ_FENCE();
            const double d = dp.x;
            if (!(dp.y > 0.0)) pd = false;
            const bool mine = me.cl == pr;
            const double g = mine ? 0.0 : -d * up;
            rank1_range<0, 0, R1, NB, NH>(U, B, g);
            if (me.hb == PH) reg_put<PB * 16, P1, NH>(U, pr, mine ? -1.0 : -g);
            sc = mine ? d : sc;
          }
        };
        [&]<int... IS>(std::integer_sequence<int, IS...>) {
          (solo16(std::integral_constant<int, IS / (NB > 1 ? NB - 1 : 1)>{}, std::integral_constant<int, IS % (NB > 1 ? NB - 1 : 1)>{}), ...);
        }(std::make_integer_sequence<int, 2 * (NB > 1 ? NB - 1 : 1)>{});
        }
        const unsigned pbw = L.vec + (me.wv == 3 ? 2 * L.pbstride * 8 : 0);
        // the column factors and the definiteness flags of the two sweeps, to the wavefronts that share their columns
        if (me.wv == 0 || me.wv == 3) {
          if (inpad) sd(lds, pbw + L.pbstride * 8 + slot8, sc);
          if (me.cl == 0) sd(lds, pbw + L.nh2 * 8, pd ? 1.0 : 0.0);
        }
        __syncthreads();
        if (me.wv == 1 || me.wv == 2) {
          const unsigned pbo = L.vec + (me.wv == 1 ? 2 * L.pbstride * 8 : 0);
          if (inpad) sc = ld(lds, pbo + L.pbstride * 8 + slot8);
        }
        pd = ld(lds, L.vec + L.nh2 * 8) != 0.0 && ld(lds, L.vec + 2 * L.pbstride * 8 + L.nh2 * 8) != 0.0;
        __syncthreads();
      }
      auto sweep16 = [&](auto ph_tag, auto pb_tag) {
        constexpr int PH = decltype(ph_tag)::value, PB = decltype(pb_tag)::value;
        constexpr int P1 = (PB + 1) * 16 < NH ? (PB + 1) * 16 : NH;
        ME;
        const unsigned slot8 = (me.cb * NHP + me.cl) * 8;       // column j in a padded buffer (column halves padded like row halves)
        const unsigned brow8 = (me.hb * NHP + me.lane16) * 8;   // the lane's element of the row half's broadcast registers
        const bool inpad = NHP == 64 || me.cl < NHP;
#pragma unroll 1
        for (int pr = (PH ? B1 : T1) > PB * 16 ? (PH ? B1 : T1) : PB * 16; pr < P1 && PH * NH + pr < n; pr++) {  // what the first phase left
          const int p = PH * NH + pr;
          const unsigned pb = L.vec + (p & 1) * 2 * L.pbstride * 8;
          if (me.hb == PH) {  // the wavefronts that hold row p publish it
            const double up = reg_get<PB * 16, P1, NH>(U, pr);  // lanes without a column hold zeros
            const double pv = sc * up;
            if (inpad) { sd(lds, pb + slot8, pv); sd(lds, pb + L.pbstride * 8 + slot8, up); }
            if (me.jp == p) {
              double d = __builtin_amdgcn_rcp(pv);
              d = __builtin_fma(__builtin_fma(-pv, d, 1.0), d, d);
              d = __builtin_fma(__builtin_fma(-pv, d, 1.0), d, d);
              st2at(lds, pb + L.nh2 * 8, d, pv);
            }
          }
          __syncthreads();
          double B[NB];
#pragma unroll
          for (int k = 0; k < NB; k++) B[k] = ld(lds, pb + brow8 + 128 * k);
          const double up = inpad ? ld(lds, pb + L.pbstride * 8 + slot8) : 0.0;
          const d2_t dp = ld2at(lds, pb + L.nh2 * 8);
          const double d = dp.x;
          if (!(dp.y > 0.0)) pd = false;
          const double g = (me.jp == p) ? 0.0 : -d * up;
          rank1_all<0, NB, NH>(U, B, g);
          if (me.hb == PH) reg_put<PB * 16, P1, NH>(U, pr, (me.jp == p) ? -1.0 : -g);
          sc = (me.jp == p) ? d : sc;
        }
      };
      // the compiled-in family: the four-wavefront steps of the last two sixteen-pivot blocks of a row half unrolled as well
      // (the pivot a compile-time register: no selection trees; what the first phase leaves of the MPC pattern -- ten pivots
      // per half -- lies there); UB: the first unrolled block
      constexpr int UB = CN > 0 && NB > 2 ? NB - 2 : NB;
      auto joint = [&](auto ph_tag) {
        constexpr int PH = decltype(ph_tag)::value;
        if constexpr (UB < NB) {
          ME;
          const unsigned slot8 = (me.cb * NHP + me.cl) * 8, brow8 = (me.hb * NHP + me.lane16) * 8;
          const bool inpad = NHP == 64 || me.cl < NHP;
          const int lo = PH ? B1 : T1;
          auto step = [&](auto k_tag) {
            constexpr int K = UB * 16 + decltype(k_tag)::value, p = PH * NH + K;
            constexpr unsigned pbo = (p & 1) * 2;
            if (K >= lo && p < n) {
              const unsigned pb = L.vec + pbo * L.pbstride * 8;
              if (me.hb == PH) {
                const double up = U[K], pv = sc * up;
                if (inpad) { sd(lds, pb + slot8, pv); sd(lds, pb + L.pbstride * 8 + slot8, up); }
                if (me.jp == p) {
                  double d = __builtin_amdgcn_rcp(pv);
                  d = __builtin_fma(__builtin_fma(-pv, d, 1.0), d, d);
                  d = __builtin_fma(__builtin_fma(-pv, d, 1.0), d, d);
                  st2at(lds, pb + L.nh2 * 8, d, pv);
                }
              }
              __syncthreads();
              double B[NB];
#pragma unroll
              for (int k = 0; k < NB; k++) B[k] = ld(lds, pb + brow8 + 128 * k);
              const double up = inpad ? ld(lds, pb + L.pbstride * 8 + slot8) : 0.0;
              const d2_t dp = ld2at(lds, pb + L.nh2 * 8);
              const double d = dp.x;
              if (!(dp.y > 0.0)) pd = false;
              const double g = (me.jp == p) ? 0.0 : -d * up;
              rank1_all<0, NB, NH>(U, B, g);
              if (me.hb == PH) U[K] = (me.jp == p) ? -1.0 : -g;
              sc = (me.jp == p) ? d : sc;
            }
          };
          [&]<int... IS>(std::integer_sequence<int, IS...>) { (step(std::integral_constant<int, IS>{}), ...); }(std::make_integer_sequence<int, NH - UB * 16>{});
        }
      };
      [&]<int... IS>(std::integer_sequence<int, IS...>) { (sweep16(std::integral_constant<int, 0>{}, std::integral_constant<int, IS>{}), ...); }(std::make_integer_sequence<int, UB>{});
      joint(std::integral_constant<int, 0>{});
      [&]<int... IS>(std::integer_sequence<int, IS...>) { (sweep16(std::integral_constant<int, 1>{}, std::integral_constant<int, IS>{}), ...); }(std::make_integer_sequence<int, UB>{});
      joint(std::integral_constant<int, 1>{});
      nsc = -sc;  // the registers stay as the sweeps left them: column j of M^-1 is nsc U, applied to the sum of the product
      __syncthreads();  // the pivot buffers are the partial b / x~ and the copy of x again
      for (int k = tid(); k < 2 * L.nh2 + 4 * n; k += QT) sd(lds, L.vec + k * 8, 0.0);
      __syncthreads();
      if (!uni((int)pd)) { status = OSQP_NON_CVX; iter--; break; }
      need_factor = false;
      QPROF(3)
    }
    // ---- K5 (a): b_j = sigma x_j - q_j + (A'(rho z - y))_j, in two parts (the entries of either parity) -------------
    {
      ME;
      double xo = 0.0, qo = 0.0;
      if (me.owner) { xo = ld(lds, L.cx + me.j * 8); qo = ld(lds, L.cq + me.j * 8); }
      double a = col_dot(me, F_ZT);
      if (me.owner) a += sigma * xo - qo;
      if (me.col) sd(lds, L.bp + (me.hb * L.nh2 + me.cb * NHP + me.cl) * 8, a);
    }
    __syncthreads();
    QPROF(4)
    // ---- K3/K4 as one dense product: the part of x~_j over the rows of this half ----------------------------------------
    {
      ME;
      const unsigned b8 = L.bp + (me.hb * NHP + me.lane16) * 8;
      double B[NB];
#pragma unroll
      for (int k = 0; k < NB; k++) B[k] = ld(lds, b8 + 128 * k) + ld(lds, b8 + L.nh2 * 8 + 128 * k);
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      dot_all<0, NB, NH, 4>(U, B, acc);
      if (me.col) sd(lds, L.xp + (me.j * 2 + me.hb) * 8, nsc * ((acc[0] + acc[1]) + (acc[2] + acc[3])));
    }
    __syncthreads();
    QPROF(5)
    // ---- K5 (b): x, delta_x by the column's owner; z~ = A x~ row by row, each row finished by its lane ------------------
    const double alpha = opaque_s(st.alpha);
    const bool last = iter == max_iter;
    const bool due = check && (iter % check == 0);
    const bool rho_due = rho_interval && (iter % rho_interval == 0);
    const bool evaluate = due || rho_due || last;
    {
      ME;
      const unsigned r = my_rec(me);
      const double zt = row_dot(me);
      if (r != 0xFFFFFFFFu) {
        const d2_t zy = ld2at(lds, L.rec + r + F_Z), lu = ld2at(lds, L.rec + r + F_L), rr = ld2at(lds, L.rec + r + F_RHO);
        const double zh = alpha * zt + (1.0 - alpha) * zy.x;
        const double zn = fmin(fmax(zh + rr.y * zy.y, lu.x), lu.y);
        const double d = rr.x * (zh - zn);
        const double yn = zy.y + d;
        st2at(lds, L.rec + r + F_Z, zn, yn);
        sd(lds, L.rec + r + F_ZT, rr.x * zn - yn);
        if (evaluate) sd(lds, L.dy + (r / RECB) * 8, d);
      }
      if (me.owner) {  // last: its operands would otherwise sit in registers through the row products
        const d2_t xa = ld2at(lds, L.xp + me.j * 16);
        const double xo = ld(lds, L.cx + me.j * 8);
        const double xn = alpha * (xa.x + xa.y) + (1.0 - alpha) * xo;
        sd(lds, L.cx + me.j * 8, xn);
        if (evaluate) sd(lds, L.cdx + me.j * 8, xn - xo);
      }
    }
    __syncthreads();
    QPROF(6)
    if (!evaluate) continue;

    // ---- K8: residual evaluation; termination tests ----------------------------------------------------
    lchar *nrm = lds + L.nrm;
    auto NRM = [&](int k) -> double { return *(const ldouble *)(nrm + 8 * k); };
    // full column sums of A' v (v a field of the row records): the odd half through tmp; one barrier
    auto col_full = [&](const Me &me, int field) -> double {
      double a = col_dot(me, field);
      if (me.col && me.hb == 1) sd(lds, L.tmp + me.j * 8, a);
      __syncthreads();
      if (me.owner) a += ld(lds, L.tmp + me.j * 8);
      return a;
    };
    {
      // every group of norms goes through its reduction and into nrm[] before the next is formed: the evaluation runs next
      // to the inverse (2 NH registers), its own working set has to stay small
      ME;
      const int t = me.t, j = me.j;
      const unsigned myrec = my_rec(me);
      const bool hasrow = myrec != 0xFFFFFFFFu, owner = me.owner;
      if (owner) { const double xv = ld(lds, L.cx + j * 8); sd(lds, L.xs + j * 8, xv); st2at(lds, L.xp + j * 16, xv, fresh_zero()); }
      __syncthreads();
      {
        double v[6] = {0, 0, 0, 0, 0, 0};
        const double ax = row_dot(me);
        if (hasrow) {
          const double zi = ld(lds, L.rec + myrec + F_Z), e = 1.0 / ld(lds, L.rec + myrec + F_E), rs = ax - zi;
          v[0] = fabs(rs); v[1] = fabs(e * rs); v[2] = fabs(zi); v[3] = fabs(ax); v[4] = fabs(e * zi); v[5] = fabs(e * ax);
        }
        quad_reduce<6, 0>(v, lds, L.red, flip, wvs);
        if (t == 0) {
#pragma unroll
          for (int k = 0; k < 6; k++) *(ldouble *)(nrm + 8 * k) = v[k];
          *(ldouble *)(nrm + 8 * N_PRI) = m == 0 ? 0.0 : (uns ? v[1] : v[0]);
        }
      }
      OQ_FENCE();
      const double at = col_full(me, F_Y);
      OQ_FENCE();
      const double px = p_row_dot(me, L.xs);
      const double qj = owner ? ld(lds, L.cq + j * 8) : 0.0, dinv = owner ? 1.0 / ld(lds, L.cD + j * 8) : 0.0;
      {
        double w[4] = {0, 0, 0, 0};
        if (owner) { const double r = (qj + px) + at; w[0] = fabs(r); w[1] = fabs(dinv * r); w[2] = fabs(qj); w[3] = fabs(at); }
        quad_reduce<4, 0>(w, lds, L.red, flip, wvs);
        if (t == 0) {
#pragma unroll
          for (int k = 0; k < 4; k++) *(ldouble *)(nrm + 8 * (6 + k)) = w[k];
          *(ldouble *)(nrm + 8 * N_DUA) = uns ? cinv * w[1] : w[0];
        }
      }
      OQ_FENCE();
      {
        double w[4] = {0, 0, 0, 0};
        if (owner) { w[0] = fabs(px); w[1] = fabs(dinv * qj); w[2] = fabs(dinv * at); w[3] = fabs(dinv * px); }
        quad_reduce<4, 0>(w, lds, L.red, flip, wvs);
        if (t == 0) {
#pragma unroll
          for (int k = 0; k < 4; k++) *(ldouble *)(nrm + 8 * (10 + k)) = w[k];
        }
      }
      OQ_FENCE();
      {
        const double xj = owner ? ld(lds, L.cx + j * 8) : 0.0;
        double sm[2] = {xj * px, qj * xj};
        quad_reduce<2, 1>(sm, lds, L.red, flip, wvs);
        if (t == 0) *(ldouble *)(nrm + 8 * N_OBJ) = cinv * (0.5 * sm[0] + sm[1]);
      }
      __syncthreads();
    }
    pri_res = uni(NRM(N_PRI)); dua_res = uni(NRM(N_DUA)); obj = uni(NRM(N_OBJ));
    int code = 0;
    const int passes = (due || last) ? (last ? 2 : 1) : 0;
    for (int pass = 0; pass < passes && code == 0; pass++) {
      ME;
      const int t = me.t, j = me.j;
      const bool owner = me.owner;
      const bool approx = pass == 1;
      double ea = st.eps_abs, er = st.eps_rel, epi = st.eps_prim_inf, edi = st.eps_dual_inf;
      if (!(pri_res <= OSQP_INFTY) || !(dua_res <= OSQP_INFTY)) { code = OSQP_NON_CVX; break; }
      if (approx) { ea *= 10; er *= 10; epi *= 10; edi *= 10; }
      bool pc = false, dc = false, pinf = false, dinf = false;
      if (m == 0) pc = true;
      else {
        const double eps_p = ea + er * (uns ? nmax(NRM(4), NRM(5)) : nmax(NRM(2), NRM(3)));
        if (uni((int)(pri_res < eps_p))) pc = true;
        else {  // primal infeasibility on delta_y
          double v1[1] = {0.0}, s1[1] = {0.0};
          for (int i = t; i < m; i += QT) {
            const unsigned r = (unsigned)i * RECB;
            const double lo = ld(lds, L.rec + r + F_L), hi = ld(lds, L.rec + r + F_U), e = ld(lds, L.rec + r + F_E);
            double d = ld(lds, L.dy + i * 8);
            if (hi > B_INF) { if (lo < -B_INF) d = 0.0; else d = fmin(d, 0.0); }
            else if (lo < -B_INF) d = fmax(d, 0.0);
            sd(lds, L.dy + i * 8, d);
            v1[0] = nmax(v1[0], fabs(uns ? e * d : d));
            s1[0] += hi * fmax(d, 0.0) + lo * fmin(d, 0.0);
          }
          quad_reduce<1, 0>(v1, lds, L.red, flip, wvs);
          quad_reduce<1, 1>(s1, lds, L.red, flip, wvs);
          const double nv = v1[0], lhs = s1[0];
          if (uni((int)(nv > epi && lhs < -epi * nv))) {
            // A' delta_y: the column walk reads its operand from a row record -- lend the ZT field of every record to delta_y
            for (int i = t; i < m; i += QT) {
              const unsigned r = (unsigned)i * RECB;
              sd(lds, L.ax + i * 8, ld(lds, L.rec + r + F_ZT));
              sd(lds, L.rec + r + F_ZT, ld(lds, L.dy + i * 8));
            }
            __syncthreads();
            const double tn = col_full(me, F_ZT);
            __syncthreads();
            for (int i = t; i < m; i += QT) sd(lds, L.rec + (unsigned)i * RECB + F_ZT, ld(lds, L.ax + i * 8));
            double w1[1] = {owner ? fabs(uns ? tn / ld(lds, L.cD + j * 8) : tn) : 0.0};
            quad_reduce<1, 0>(w1, lds, L.red, flip, wvs);
            pinf = w1[0] < epi * nv;
            __syncthreads();
          }
        }
      }
      const double eps_d = ea + er * (uns ? cinv * nmax(NRM(11), nmax(NRM(12), NRM(13))) : nmax(NRM(8), nmax(NRM(9), NRM(10))));
      if (uni((int)(dua_res < eps_d))) dc = true;
      else {  // dual infeasibility on delta_x
        const double dxj = owner ? ld(lds, L.cdx + j * 8) : 0.0, Dj = owner ? ld(lds, L.cD + j * 8) : 1.0;
        double v1[1] = {fabs(uns ? Dj * dxj : dxj)}, s1[1] = {owner ? ld(lds, L.cq + j * 8) * dxj : 0.0};
        quad_reduce<1, 0>(v1, lds, L.red, flip, wvs);
        quad_reduce<1, 1>(s1, lds, L.red, flip, wvs);
        const double nv = v1[0], qdx = s1[0];
        const double cs = uns ? c : 1.0;
        if (uni((int)(nv > edi && qdx < -cs * edi * nv))) {
          __syncthreads();
          if (owner) { sd(lds, L.xs + j * 8, dxj); st2at(lds, L.xp + j * 16, dxj, fresh_zero()); }
          __syncthreads();
          const double pdx = p_row_dot(me, L.xs);
          double w1[1] = {owner ? fabs(uns ? pdx / Dj : pdx) : 0.0};
          quad_reduce<1, 0>(w1, lds, L.red, flip, wvs);
          if (uni((int)(w1[0] < cs * edi * nv))) {
            double bad[1] = {0.0};
            const double adx = row_dot(me);
            const unsigned myrec = my_rec(me);
            if (myrec != 0xFFFFFFFFu) {
              const double tt = uns ? adx / ld(lds, L.rec + myrec + F_E) : adx;
              const double lo = ld(lds, L.rec + myrec + F_L), hi = ld(lds, L.rec + myrec + F_U);
              if ((hi < B_INF && tt > edi * nv) || (lo > -B_INF && tt < -edi * nv) || tt != tt) bad[0] = 1.0;
            }
            quad_reduce<1, 0>(bad, lds, L.red, flip, wvs);
            dinf = bad[0] == 0.0;
          }
          __syncthreads();
        }
      }
      if (pc && dc) code = approx ? OSQP_SOLVED_INACCURATE : OSQP_SOLVED;
      else if (pinf) code = approx ? OSQP_PRIMAL_INFEASIBLE_INACCURATE : OSQP_PRIMAL_INFEASIBLE;
      else if (dinf) code = approx ? OSQP_DUAL_INFEASIBLE_INACCURATE : OSQP_DUAL_INFEASIBLE;
    }
    if (last && passes && code == 0) code = OSQP_MAX_ITER_REACHED;
    code = uni(code);
    QPROF(7)
    if (code != 0) { status = code; break; }
    // ---- adaptive rho ----
    if (rho_due) {
      const double pr = m == 0 ? 0.0 : NRM(0) / (nmax(NRM(2), NRM(3)) + 1e-10);
      const double du = NRM(6) / (nmax(nmax(NRM(8), NRM(9)), NRM(10)) + 1e-10);
      const double est = uni(fmin(fmax(rho * sqrt(pr / (du + 1e-10)), B_RHO_MIN), B_RHO_MAX));
      if (est > rho * st.adaptive_rho_tolerance || est < rho / st.adaptive_rho_tolerance) {
        rho = est; rho_updates++;
        __syncthreads();
        set_rho(false);
        need_factor = true;
      }
    }
    QPROF(8)
  }
  if (iter > max_iter) iter = max_iter;
  QPROF_PRINT
  // ---- store -----------------------------------------------------
  {
    ME;
    const bool has_sol = status == OSQP_SOLVED || status == OSQP_SOLVED_INACCURATE || status == OSQP_MAX_ITER_REACHED;
    if (me.owner) x_out[(size_t)inst * x_stride + S.perm[me.j]] = has_sol ? ld(lds, L.cD + me.j * 8) * ld(lds, L.cx + me.j * 8) : NAN;
    for (int i = me.t; i < m; i += QT) {
      const unsigned r = (unsigned)i * RECB;
      y_out[(size_t)inst * y_stride + i] = has_sol ? cinv * ld(lds, L.rec + r + F_E) * ld(lds, L.rec + r + F_Y) : NAN;
    }
    if (me.t == 0) {
      double *o = info_out + (size_t)inst * info_stride;
      o[0] = (double)iter; o[1] = (double)status; o[2] = pri_res; o[3] = dua_res;
      if (info_cols > 4) { o[4] = status == OSQP_NON_CVX ? NAN : obj; o[5] = (double)rho_updates; }
    }
  }
#undef ME
}

// Two occupancies: quadrants of at most 50 columns keep three QPs per compute unit (168 vector registers a lane), larger
// ones (NH = 64: 128 registers of inverse alone) two.
template <int NH, int KC, int KE, int CH, int CN, int CM, int CA, int CF>
__global__ __launch_bounds__(QT) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_batch_quad(
    Sched S, OSQPSettings st, int count, const double *__restrict__ Px_all, const double *__restrict__ Ax_all,
    const double *__restrict__ q_all, const double *__restrict__ l_all, const double *__restrict__ u_all,
    double *__restrict__ x_out, double *__restrict__ y_out, double *__restrict__ info_out, int x_stride, int y_stride,
    int info_stride, int info_cols) {
  quad_body<NH, KC, KE, CH, CN, CM, CA, CF>(S, st, count, Px_all, Ax_all, q_all, l_all, u_all, x_out, y_out, info_out, x_stride, y_stride,
                                            info_stride, info_cols);
}
template <int NH, int KC, int KE, int CH, int CN, int CM, int CA, int CF>
__global__ __launch_bounds__(QT) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_batch_quad2(
    Sched S, OSQPSettings st, int count, const double *__restrict__ Px_all, const double *__restrict__ Ax_all,
    const double *__restrict__ q_all, const double *__restrict__ l_all, const double *__restrict__ u_all,
    double *__restrict__ x_out, double *__restrict__ y_out, double *__restrict__ info_out, int x_stride, int y_stride,
    int info_stride, int info_cols) {
  quad_body<NH, KC, KE, CH, CN, CM, CA, CF>(S, st, count, Px_all, Ax_all, q_all, l_all, u_all, x_out, y_out, info_out, x_stride, y_stride,
                                            info_stride, info_cols);
}

}  // namespace quad
