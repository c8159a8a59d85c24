// panel.hip -- LDS-staged SpMV for matrices whose x vector does not fit the caches
// (BASELINE.json config 3: n = m = 1e6, 1000 non-zeros per row, x = 8 MB).
//
// The plain CSR kernel (k_spmv) streams the matrix at full width but gathers x with one
// random 8-byte access per non-zero; at n = 1e6 every gather is an L2 / Infinity-Fabric line
// fetch and the kernel runs at ~1.1 TB/s of algorithmic bandwidth (profiles/r01_a_*).
// Here the columns are cut into panels of W = 2^shift columns (default 16384 = 128 KB of x); a
// 1024-thread workgroup stages a panel of x in LDS, then streams the rows of its tile inside that
// panel and gathers from LDS.  Consecutive panels form groups of Gp (1 for mid-size matrices, 4 when
// there are enough non-zeros to keep > 1000 workgroups busy); the rows of every group are cut into
// tiles at equal non-zero counts (not equal row counts: dense corners leave no tail, and the cut
// is two-dimensional -- a matrix whose density varies along its columns stays balanced).  A
// workgroup walks the Gp panels of its tile one after the other with the row sums staying in LDS,
// and writes them once per group to an [NG x rows] buffer that a second kernel adds up in group
// order (fixed order => reproducible) together with the SpMV epilogue.
//
// Inside a tile the entries are laid out as sliced ELL:
//   * the rows of the tile are ordered by their length inside the panel, longest first;
//   * 64 consecutive rows of that order form a slice, stored column-major and padded to the
//     longest row of the slice (sorted => a few per cent of padding): element k of lane l sits at
//     slice_base + 64 k + l, so one load instruction of a wavefront moves 512 contiguous bytes of
//     values or 128 contiguous bytes of 16-bit column indices;
//   * lane = row: no shuffles, no reduction; the lane adds its row in ascending column order.
//     Rows with no entry in the panel are not stored at all (their cell of the partial-sum
//     buffer is zeroed once at build time).
// The 16 wavefronts of a workgroup share the LDS copy of the x panel and take slices round-robin;
// the tile's row sums are staged in LDS and stored as one contiguous block.
// (A group-per-row CSR walk over the same panels moves fewer bytes but reached only 3.3 TB/s
// of real traffic with its 8- and 2-byte loads on ~16-entry segments; profiles/r01_e_panel_sweep.md.)
//
// Wide mode (matrices with fewer than ~1.5 entries per row and 16384-column panel: n >> 1e6 at fixed nnz): the same
// tiles and slices over panels of 2^18 columns, whose 2 MB of x are not staged in LDS but left to the L2 of the XCD --
// the workgroups of a panel run side by side (tiles are launched panel-major), so their gathers hit the same lines;
// 32-bit local column ids.  Slower per byte than the LDS scheme, several times faster than gathering from a 32 MB x.
//
// The whole layout is planned on the device (tile cuts, per-tile ordering, slice offsets): the
// host only reads back three totals.
//
// Algorithmic bytes keep the CSR accounting of SURVEY.md 8d (12 nnz + ...); the panel copy
// actually moves 10 B per stored entry + 16 B per (row, panel) of partial sums.
#include <algorithm>

#include "kernels.hpp"
#include <functional>
#include "devutil.hpp"

namespace oq {

namespace {

constexpr int kThreads = 1024;
#ifndef OQ_SELL_BATCH
#define OQ_SELL_BATCH 16
#endif
constexpr int kBatch = OQ_SELL_BATCH;  // entries of a slice per lane whose loads are issued together
constexpr int kWaves = kThreads / 64;
constexpr int kTileRowsMax = 3968;  // row sums of a tile are staged in LDS (31 KB next to the 128 KB x panel)
constexpr int kSortN = 4096;        // power of two >= kTileRowsMax: per-tile ordering of the rows in LDS

// tunables (defaults from the sweep in profiles/r01_e_panel_sweep.md; overridable for experiments)
int env_int(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
int panel_shift() { return env_int("OSQP_AMD_PANEL_SHIFT", 14); }           // W = 16384 columns = 128 KB of fp64 in LDS
int panel_tile_nnz() { return env_int("OSQP_AMD_PANEL_TILE_NNZ", 65536); }  // non-zeros per workgroup tile

// first position in [s, e) with col >= target (cols ascending inside a row)
__device__ __forceinline__ int64_t lower_bound_col(const int *__restrict__ col, int64_t s, int64_t e, int target) {
  while (s < e) { int64_t mid = (s + e) >> 1; if (col[mid] < target) s = mid + 1; else e = mid; }
  return s;
}

// ---------------------------------------------------------------------------------------------
// planning
// ---------------------------------------------------------------------------------------------
// cnt[b * rows + i] = entries of row i inside panel b   (one wavefront per row, lane = panel)
// cellsrc[b * rows + i] = CSR position of the first of them (what the first fill of the slices reads from)
__global__ __launch_bounds__(kBlock) void k_panel_count(int rows, int B, int shift, const int64_t *__restrict__ rp,
                                                        const int *__restrict__ col, int64_t *__restrict__ cnt,
                                                        uint32_t *__restrict__ cellsrc) {
  const int lane = threadIdx.x & 63;
  const int64_t row = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  if (row >= rows) return;
  const int64_t s = rp[row], e = rp[row + 1];
  for (int b = lane; b < B; b += 64) {
    int64_t lo = lower_bound_col(col, s, e, b << shift);
    int64_t hi = (b + 1 == B) ? e : lower_bound_col(col, s, e, (b + 1) << shift);
    cnt[(size_t)b * rows + row] = hi - lo;
    cellsrc[(size_t)b * rows + row] = (uint32_t)lo;
  }
}

// gcnt[g * rows + i] = entries of row i inside the panels of group g
__global__ __launch_bounds__(kBlock) void k_group_count(int rows, int B, int Gp, int64_t gcells, const int64_t *__restrict__ cnt,
                                                        int64_t *__restrict__ gcnt) {
  const int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (c >= gcells) return;
  const int64_t g = c / rows;
  const int r = (int)(c - g * rows);
  int64_t a = 0;
  for (int b = (int)g * Gp; b < B && b < ((int)g + 1) * Gp; b++) a += cnt[(size_t)b * rows + r];
  gcnt[c] = a;
}
// unit u = t * Gp + j is tile t inside the j-th panel of its group (an empty row range when the group has fewer panels)
__global__ __launch_bounds__(kBlock) void k_expand_units(int ntiles, int B, int Gp, const int *__restrict__ tile_g,
                                                         const int *__restrict__ tile_r0, const int *__restrict__ tile_r1,
                                                         int *__restrict__ ub, int *__restrict__ u0, int *__restrict__ u1) {
  const int64_t u = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (u >= (int64_t)ntiles * Gp) return;
  const int t = (int)(u / Gp), j = (int)(u - (int64_t)t * Gp);
  const int b = tile_g[t] * Gp + j;
  const bool real = b < B;
  ub[u] = real ? b : 0;
  u0[u] = tile_r0[t];
  u1[u] = real ? tile_r1[t] : tile_r0[t];
}

// Tile cuts.  Every (group, row) cell costs max(entries, cmin) with cmin = ceil(budget / kTileRowsMax); coff is the
// exclusive scan of the costs (group-major).  Inside group g the cells whose cost offset falls into the same
// window of `budget` share a tile: ~budget non-zeros where the rows are long enough, never more than
// kTileRowsMax rows where they are short.  start[c] = 1 when cell c = (b, r) opens a tile.
__global__ __launch_bounds__(kBlock) void k_tile_cost(int64_t cells, int64_t cmin, int64_t *__restrict__ cnt) {
  const int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (c < cells && cnt[c] < cmin) cnt[c] = cmin;
}
__global__ __launch_bounds__(kBlock) void k_tile_starts(int rows, int64_t cells, const int64_t *__restrict__ coff, int64_t budget,
                                                        int64_t *__restrict__ start) {
  const int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (c >= cells) return;
  const int64_t b = c / rows;
  const int r = (int)(c - b * rows);
  const int64_t *po = coff + b * rows;
  const int64_t base = po[0];
  start[c] = (r == 0 || (po[r - 1] - base) / budget != (po[r] - base) / budget) ? 1 : 0;
}
// tile id of an opening cell = exclusive scan of start; fill (panel, first row) of every tile
__global__ __launch_bounds__(kBlock) void k_tile_fill(int rows, int64_t cells, const int64_t *__restrict__ start,
                                                      const int64_t *__restrict__ tid, int *__restrict__ tile_b, int *__restrict__ tile_r0) {
  const int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (c >= cells || !start[c]) return;
  const int64_t b = c / rows;
  tile_b[tid[c]] = (int)b;
  tile_r0[tid[c]] = (int)(c - b * rows);
}
__global__ __launch_bounds__(kBlock) void k_tile_ends(int rows, int ntiles, const int *__restrict__ tile_b, const int *__restrict__ tile_r0,
                                                      int *__restrict__ tile_r1) {
  const int t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= ntiles) return;
  tile_r1[t] = (t + 1 < ntiles && tile_b[t + 1] == tile_b[t]) ? tile_r0[t + 1] : rows;
}

// The rows of a tile ordered by length inside the panel, longest first, ties by row id: bitonic sort of
// (maxlen - length) << 12 | local row in LDS (one 1024-thread workgroup; lengths <= W <= 2^15, rows < 2^12).
// On return key[i] & 4095 is the i-th local row and *nz_rows the number of rows with at least one entry.
__device__ void order_tile_rows(uint32_t *key, const int64_t *__restrict__ po, int r0, int nrows, int *nz_rows) {
  if (threadIdx.x == 0) *nz_rows = 0;
  __syncthreads();
  int mine = 0;
  for (int i = threadIdx.x; i < kSortN; i += kThreads) {
    uint32_t k = 0xFFFFFFFFu;
    if (i < nrows) {
      const uint32_t len = (uint32_t)(po[r0 + i + 1] - po[r0 + i]);
      k = ((0x7FFFFu - len) << 12) | (uint32_t)i;
      mine += len > 0;
    }
    key[i] = k;
  }
  if (mine) atomicAdd(nz_rows, mine);
  __syncthreads();
  for (int k = 2; k <= kSortN; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < kSortN; i += kThreads) {
        const int p = i ^ j;
        if (p > i) {
          const uint32_t a = key[i], b = key[p];
          if ((a > b) == ((i & k) == 0)) { key[i] = b; key[p] = a; }
        }
      }
      __syncthreads();
    }
}
__device__ __forceinline__ uint32_t key_len(uint32_t key) { return 0x7FFFFu - (key >> 12); }

// pass 1: number of slices and padded size of every tile
__global__ __launch_bounds__(kThreads) void k_tile_measure(int rows, const int *__restrict__ tile_b, const int *__restrict__ tile_r0,
                                                           const int *__restrict__ tile_r1, const int64_t *__restrict__ off,
                                                           int64_t *__restrict__ nslices, int64_t *__restrict__ padded) {
  __shared__ uint32_t key[kSortN];
  __shared__ int nz;
  __shared__ unsigned long long total;
  const int t = blockIdx.x;
  const int r0 = tile_r0[t], nrows = tile_r1[t] - r0;
  if (threadIdx.x == 0) total = 0ULL;
  order_tile_rows(key, off + (size_t)tile_b[t] * rows, r0, nrows, &nz);
  const int ns = (nz + 63) >> 6;
  unsigned long long mine = 0;
  for (int sl = threadIdx.x; sl < ns; sl += kThreads) mine += 64ULL * key_len(key[sl * 64]);
  if (mine) atomicAdd(&total, mine);
  __syncthreads();
  if (threadIdx.x == 0) { nslices[t] = ns; padded[t] = (int64_t)total; }
}
// pass 2: slice tables and the (panel, row) -> slot map
__global__ __launch_bounds__(kThreads) void k_tile_layout(int rows, const int *__restrict__ tile_b, const int *__restrict__ tile_r0,
                                                          const int *__restrict__ tile_r1, const int64_t *__restrict__ off,
                                                          const int64_t *__restrict__ slice0, const int64_t *__restrict__ padded0,
                                                          int *__restrict__ tile_s0, int *__restrict__ tile_ns,
                                                          uint32_t *__restrict__ slice_base, int *__restrict__ slice_len,
                                                          int *__restrict__ slice_rows, uint32_t *__restrict__ cellbase) {
  __shared__ uint32_t key[kSortN];
  __shared__ uint32_t sbase[kSortN / 64 + 1];
  __shared__ int nz;
  const int t = blockIdx.x, b = tile_b[t];
  const int r0 = tile_r0[t], nrows = tile_r1[t] - r0;
  order_tile_rows(key, off + (size_t)b * rows, r0, nrows, &nz);
  const int ns = (nz + 63) >> 6;
  if (threadIdx.x == 0) {  // at most 62 slices: a serial prefix sum is fine
    uint32_t acc = (uint32_t)padded0[t];
    for (int sl = 0; sl < ns; sl++) { sbase[sl] = acc; acc += 64u * key_len(key[sl * 64]); }
    tile_s0[t] = (int)slice0[t];
    tile_ns[t] = ns;
  }
  __syncthreads();
  const int64_t s0 = slice0[t];
  for (int i = threadIdx.x; i < ns * 64; i += kThreads) {
    const int sl = i >> 6, lane = i & 63;
    int row = -1;
    if (i < nz) {
      row = r0 + (int)(key[i] & 4095u);
      cellbase[(size_t)b * rows + row] = sbase[sl] + (uint32_t)lane;
    }
    slice_rows[(size_t)(s0 + sl) * 64 + lane] = row;
    if (lane == 0) { slice_base[s0 + sl] = sbase[sl]; slice_len[s0 + sl] = (int)key_len(key[sl * 64]); }
  }
}

// copy every entry of the CSR matrix to its sliced-ELL slot; with_cols = 0 refreshes the values only; slot (may be null):
// slot[k] = where entry k went (the map compaction needs, recorded in passing).  One wavefront per row, 64 consecutive
// entries at a time.  An entry's place inside its (row, panel) cell is its distance from the first entry of the row in
// that panel: the columns of a row ascend, so that first entry is where the panel id last changed -- found with one
// ballot over the wavefront (and a carry from the chunk before), not with a bisection of the row per entry (round 2:
// 10 dependent loads per entry, 34 ms per 1e9 entries).
template <typename ColT>
__global__ __launch_bounds__(kBlock) void k_sell_scatter(int rows, int shift, const int64_t *__restrict__ rp, const int *__restrict__ col,
                                                         const double *__restrict__ val, const uint32_t *__restrict__ cellbase,
                                                         ColT *__restrict__ scol, double *__restrict__ sval, int with_cols,
                                                         uint32_t *__restrict__ slot) {
  const int lane = threadIdx.x & 63;
  const int64_t row = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  if (row >= rows) return;
  const int64_t s = rp[row], e = rp[row + 1];
  const int mask = (1 << shift) - 1;
  int carry_b = -1;          // panel of the last entry of the chunk before
  int64_t carry_seg = s;     // where that panel's run started
  for (int64_t k0 = s; k0 < e; k0 += 64) {
    const int64_t k = k0 + lane;
    const bool in = k < e;
    const int c = in ? col[k] : 0x7fffffff;
    const int b = in ? (c >> shift) : 0x7fffffff;
    int bprev = __shfl_up(b, 1, 64);
    if (lane == 0) bprev = carry_b;
    const unsigned long long heads = __ballot(b != bprev);               // lanes where a panel starts
    const unsigned long long upto = heads & (~0ull >> (63 - lane));       // ... at or before this lane
    const int64_t seg = upto ? k0 + (63 - __builtin_clzll(upto)) : carry_seg;
    if (in) {
      const size_t dst = (size_t)cellbase[(size_t)b * rows + row] + (size_t)(k - seg) * 64;
      sval[dst] = val[k];
      if (with_cols) scol[dst] = (ColT)(c & mask);
      if (slot) slot[k] = (uint32_t)dst;
    }
    // carry: the state at lane 63 (a full chunk, or the row ends here)
    carry_b = __shfl(b, 63, 64);
    carry_seg = __shfl(seg, 63, 64);
  }
}

// The FIRST fill of the slices (panel_build), through LDS: one wavefront per slice.  The entries of the slice's 64 (row, panel)
// cells are read in CSR order -- each cell is a contiguous run of ~16 entries, so the loads are whole lines -- sixteen
// positions of every row at a time, put down transposed in LDS and written out position by position as 512-byte pieces
// (values, local column ids, padding included); the slot of every entry goes out in read order.  k_sell_scatter, which
// writes the 8-byte values of a row 512 bytes apart and leaves the merging to the L2, stays for the refresh of values; a
// plain gather (lane = row, no LDS) was 17 ms SLOWER than the scatter: 64 lanes on 64 different lines per instruction.
constexpr int kFillK = 16;            // positions per pass
constexpr int kFillStride = 65;       // LDS row stride (64 rows + 1: the transposed writes spread over the banks)
constexpr int kFillWaves = 4;
template <typename ColT>
__global__ __launch_bounds__(kFillWaves * 64) void k_sell_fill_lds(int rows, int shift, const int *__restrict__ ub, const int *__restrict__ unit_s0,
                                                                 const int *__restrict__ unit_ns, const uint32_t *__restrict__ slice_base,
                                                                 const int *__restrict__ slice_len, const int *__restrict__ slice_rows,
                                                                 const uint32_t *__restrict__ cellsrc, const int64_t *__restrict__ off,
                                                                 const int *__restrict__ col, const double *__restrict__ val,
                                                                 ColT *__restrict__ scol, double *__restrict__ sval, uint32_t *__restrict__ slot,
                                                                 int what) {  // bit 0: column ids (+ slots), bit 1: values
  const bool do_cols = what & 1, do_vals = what & 2;
  __shared__ double tv[kFillWaves][kFillK * kFillStride];
  __shared__ uint32_t tc[kFillWaves][kFillK * kFillStride];
  __shared__ int start[kFillWaves][65];
  const int u = blockIdx.x, b = ub[u];
  const int s0 = unit_s0[u], ns = unit_ns[u];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int mask = (1 << shift) - 1;
  double *mv = tv[wave];
  uint32_t *mc = tc[wave];
  int *st = start[wave];
  for (int sl = s0 + wave; sl < s0 + ns; sl += kFillWaves) {
    const uint32_t base = slice_base[sl];
    const int L = slice_len[sl];
    const int row = slice_rows[(size_t)sl * 64 + lane];
    uint32_t src = 0;
    int len = 0;
    if (row >= 0) {
      const size_t c = (size_t)b * rows + row;
      src = cellsrc[c];
      len = (int)(off[c + 1] - off[c]);
    }
    for (int k0 = 0; k0 < L; k0 += kFillK) {
      int mine = len - k0;
      mine = mine < 0 ? 0 : (mine > kFillK ? kFillK : mine);
      int incl = mine;  // inclusive scan over the lanes
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
      const int total = __shfl(incl, 63, 64);
      st[lane] = incl - mine;
      if (lane == 63) st[64] = total;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      for (int f = lane; f < ((total + 63) & ~63); f += 64) {
        const bool in = f < total;
        int r = 0;
        if (in) {  // the row whose run holds f: the last r with st[r] <= f
          int lo = 0, hi = 64;
          while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (st[mid] <= f) lo = mid; else hi = mid; }
          r = lo;
          // rows with an empty share have st[r] == st[r + 1]: step to the last of the equal ones (it is the one with entries)
        }
        const uint32_t rsrc = (uint32_t)__shfl((int)src, r, 64);
        if (in) {
          const int k = f - st[r];
          const size_t g = (size_t)rsrc + (size_t)(k0 + k);
          if (do_vals) mv[k * kFillStride + r] = val[g];
          if (do_cols) {
            mc[k * kFillStride + r] = (uint32_t)(col[g] & mask);
            if (slot) slot[g] = base + (uint32_t)((k0 + k) * 64 + r);
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int kend = L - k0 < kFillK ? L - k0 : kFillK;
      for (int k = 0; k < kend; k++) {
        const size_t at = (size_t)base + (size_t)(k0 + k) * 64 + lane;
        const bool in = k < mine;
        if (do_vals) sval[at] = in ? mv[k * kFillStride + lane] : 0.0;
        if (do_cols) scol[at] = in ? (ColT)mc[k * kFillStride + lane] : (ColT)0;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// ---------------------------------------------------------------------------------------------
// the product
// ---------------------------------------------------------------------------------------------
// kSquare: y = (M .* M) x -- the Jacobi diagonal of A' diag(rho) A is this product of A' with rho (compact mode)
template <typename ColT, bool kStageX, bool kSquare>
__device__ __forceinline__ void sell_tile(int t, int rows, int cols, int shift, int B, int Gp, const int *__restrict__ tile_g,
                                          const int *__restrict__ tile_r0, const int *__restrict__ tile_r1,
                                          const int *__restrict__ unit_s0, const int *__restrict__ unit_ns,
                                          const uint32_t *__restrict__ slice_base, const int *__restrict__ slice_len,
                                          const int *__restrict__ slice_rows, const ColT *__restrict__ scol,
                                          const double *__restrict__ sval, const double *__restrict__ x, double *__restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int g = tile_g[t];
  const int W = 1 << shift;
  const int r0 = tile_r0[t], nrows = tile_r1[t] - r0;
  double *xs_lds = lds;
  double *ys = kStageX ? lds + W : lds;  // row sums of the tile: accumulated over the panels of the group, stored as one block
  for (int i = threadIdx.x; i < nrows; i += kThreads) ys[i] = 0.0;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  bool first = true;
  for (int j = 0; j < Gp; j++) {
    const int b = g * Gp + j;
    if (b >= B) break;
    const int s0 = unit_s0[(size_t)t * Gp + j], ns = unit_ns[(size_t)t * Gp + j];
    if (ns == 0) continue;           // the same decision in every thread
    const int c0 = b << shift;
    const double *xs = x + c0;       // wide mode: the panel of x stays where it is (L2)
    if (kStageX) {
      if (!first) __syncthreads();   // the previous panel is no longer read
      const int wlen = cols - c0 < W ? cols - c0 : W;
      double2 *xl = reinterpret_cast<double2 *>(xs_lds);
      if (wlen == 16 * kThreads) {
        // a full panel of the default width: 16 doubles per thread as 8 independent 16-byte loads, all in flight before the
        // first LDS write (the plain loop below is 16 dependent round trips to L2 -- as long as streaming half the tile).
        // (Fetching the NEXT panel into registers while this one streams was measured too: 1.90 -> 2.51 ms at nnz = 1e9.)
        const double2 *xg = reinterpret_cast<const double2 *>(x + c0);
        double2 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) v[k] = xg[threadIdx.x + k * kThreads];
#pragma unroll
        for (int k = 0; k < 8; k++) xl[threadIdx.x + k * kThreads] = v[k];
      } else {
        for (int i = threadIdx.x; i < wlen; i += kThreads) xs_lds[i] = x[c0 + i];
      }
      __syncthreads();
      xs = xs_lds;
    } else {
      __syncthreads();               // first pass: the zeroes of ys are in place; later: the row sums of the previous panel are
    }
    first = false;
    // A slice is short (a row has ~16 entries per panel at 1000 per row and 61 panels), so what a wavefront pays per slice
    // is round trips, not bytes: the descriptor of the NEXT slice is fetched while this one is worked on, and a slice's
    // entries go out as batches of up to kBatch values + column ids per lane, all issued before the first is consumed --
    // the tail of a slice is a guarded batch (the guards are wave-uniform), not one load at a time.
    int sl = s0 + wave;
    size_t nbase = 0;
    int nL = 0, nrow = -1;
    if (sl < s0 + ns) { nbase = (size_t)slice_base[sl]; nL = slice_len[sl]; nrow = slice_rows[(size_t)sl * 64 + lane]; }
    for (; sl < s0 + ns; sl += kWaves) {
      const size_t base = nbase + lane;
      const int L = nL, row = nrow;
      if (sl + kWaves < s0 + ns) {
        nbase = (size_t)slice_base[sl + kWaves]; nL = slice_len[sl + kWaves]; nrow = slice_rows[(size_t)(sl + kWaves) * 64 + lane];
      }
      const double *v = sval + base;
      const ColT *c = scol + base;
      double a0 = 0.0;
      for (int k = 0; k < L; k += kBatch) {
        double cv[kBatch];
        ColT cc[kBatch];
        if (k + kBatch <= L) {
#pragma unroll
          for (int u = 0; u < kBatch; u++) { cv[u] = v[(size_t)(k + u) * 64]; cc[u] = c[(size_t)(k + u) * 64]; }
        } else {
#pragma unroll
          for (int u = 0; u < kBatch; u++) {
            const bool in = k + u < L;
            cv[u] = in ? v[(size_t)(k + u) * 64] : 0.0;
            cc[u] = in ? c[(size_t)(k + u) * 64] : (ColT)0;
          }
        }
#pragma unroll
        for (int u = 0; u < kBatch; u++) a0 += (kSquare ? cv[u] * cv[u] : cv[u]) * xs[cc[u]];
      }
      if (row >= 0) ys[row - r0] += a0;  // a row appears once per panel; panels are separated by barriers
    }
  }
  __syncthreads();
  double *out = partial + (size_t)g * rows + r0;
  for (int i = threadIdx.x; i < nrows; i += kThreads) out[i] = ys[i];
}
template <typename ColT, bool kStageX, bool kSquare = false>
__global__ __launch_bounds__(kThreads) void k_spmv_sell(int rows, int cols, int shift, int B, int Gp, const int *__restrict__ tile_g,
                                                        const int *__restrict__ tile_r0, const int *__restrict__ tile_r1,
                                                        const int *__restrict__ unit_s0, const int *__restrict__ unit_ns,
                                                        const uint32_t *__restrict__ slice_base, const int *__restrict__ slice_len,
                                                        const int *__restrict__ slice_rows, const ColT *__restrict__ scol,
                                                        const double *__restrict__ sval, const double *__restrict__ x,
                                                        double *__restrict__ partial, const int *__restrict__ skip) {
  if (skip && *skip) return;
  sell_tile<ColT, kStageX, kSquare>(blockIdx.x, rows, cols, shift, B, Gp, tile_g, tile_r0, tile_r1, unit_s0, unit_ns, slice_base, slice_len,
                                    slice_rows, scol, sval, x, partial);
}
// Two matrices against the same vector in one launch (A p and P p of a CG iteration): at mid size neither fills the
// device on its own -- 1e7 non-zeros are 153 tiles on 256 CUs -- and the two products do not depend on each other.  Same
// tiles, same arithmetic: the results are those of the two single launches, bit for bit.
struct SellSide {
  int rows, cols, B, Gp, ntiles;
  const int *tile_g, *tile_r0, *tile_r1, *unit_s0, *unit_ns, *slice_len, *slice_rows;
  const uint32_t *slice_base;
  const uint16_t *scol;
  const double *sval;
  double *partial;
};
__global__ __launch_bounds__(kThreads) void k_spmv_sell_pair(SellSide a, SellSide b, int shift, const double *__restrict__ x,
                                                             const int *__restrict__ skip) {
  if (skip && *skip) return;
  const bool first = (int)blockIdx.x < a.ntiles;
  const SellSide &m = first ? a : b;
  sell_tile<uint16_t, true, false>(first ? (int)blockIdx.x : (int)blockIdx.x - a.ntiles, m.rows, m.cols, shift, m.B, m.Gp, m.tile_g, m.tile_r0,
                                   m.tile_r1, m.unit_s0, m.unit_ns, m.slice_base, m.slice_len, m.slice_rows, m.scol, m.sval, x, m.partial);
}

// y[i] = (rscale ? rscale[i] : 1) * sum_g partial[g][i] + beta * y[i] + gamma * v[i]   (group order is fixed)
// Grid of at most kReduceBlocks blocks, grid-stride over the rows -- the thread-to-row assignment of the two-stage
// reductions (k_dot_partial), so that the optional dot product of the result with another vector lands in the same
// partials, in the same order, as a separate reduce_dot would form (SpmvExtra).
__device__ __forceinline__ void panel_reduce_rows(int vb, int vg, int rows, int B, const double *__restrict__ partial, double *__restrict__ y,
                                                  const double *__restrict__ rscale, double beta, double gamma,
                                                  const double *__restrict__ v, const SpmvExtra &ex) {
  double dot = 0.0, mx = 0.0, num = 0.0, den = 0.0;
  for (int i = vb * kBlock + threadIdx.x; i < rows; i += vg * kBlock) {
    double acc = 0.0;
    int b = 0;
    for (; b + 8 <= B; b += 8) {  // eight loads in flight, added in group order
      double t[8];
#pragma unroll
      for (int u = 0; u < 8; u++) t[u] = partial[(size_t)(b + u) * rows + i];
#pragma unroll
      for (int u = 0; u < 8; u++) acc += t[u];
    }
    if (b + 4 <= B) {
      double t[4];
#pragma unroll
      for (int u = 0; u < 4; u++) t[u] = partial[(size_t)(b + u) * rows + i];
#pragma unroll
      for (int u = 0; u < 4; u++) acc += t[u];
      b += 4;
    }
    for (; b < B; b++) acc += partial[(size_t)b * rows + i];
    if (ex.y2) ex.y2[i] = ex.s2[i] * acc;
    if (rscale) acc *= rscale[i];
    if (beta != 0.0) acc += beta * y[i];
    if (v) acc += gamma * v[i];
    y[i] = acc;
    if (ex.dotv) dot += ex.dotv[i] * acc;
    if (ex.absmax_slot) mx = nanmax(mx, fabs(acc));
    if (ex.e2_partials) {
      const double d = ex.e2_x1[i] - ex.e2_x0[i], m1 = ex.e2_m1[i];
      num += d * (acc - m1);
      den += d * (m1 - ex.e2_m0[i]);
    }
  }
  if (ex.e2_partials) {
    num = block_sum(num);
    den = block_sum(den);
    if (threadIdx.x == 0) { ex.e2_partials[vb] = num; ex.e2_partials[kReduceBlocks + vb] = den; }
    if (vb == 0)
      for (int t = vg + threadIdx.x; t < kReduceBlocks; t += kBlock) { ex.e2_partials[t] = 0.0; ex.e2_partials[kReduceBlocks + t] = 0.0; }
  }
  if (ex.dot_partials) {
    dot = block_sum(dot);
    if (threadIdx.x == 0) ex.dot_partials[vb] = dot;
    if (vb == 0)  // blocks that do not exist hold zeroes, as in a kReduceBlocks-wide first stage
      for (int t = vg + threadIdx.x; t < kReduceBlocks; t += kBlock) ex.dot_partials[t] = 0.0;
  }
  if (ex.absmax_slot) {
    mx = block_max(mx);
    if (threadIdx.x == 0) atomic_max_nonneg(ex.absmax_slot, mx);
  }
}
__global__ __launch_bounds__(kBlock) void k_panel_reduce(int rows, int B, const double *__restrict__ partial, double *__restrict__ y,
                                                         const double *__restrict__ rscale, double beta, double gamma,
                                                         const double *__restrict__ v, SpmvExtra ex, const int *__restrict__ skip) {
  if (skip && *skip) return;
  panel_reduce_rows(blockIdx.x, gridDim.x, rows, B, partial, y, rscale, beta, gamma, v, ex);
}
// the reduces of a pair launch: blocks [0, ga) are the grid of the first, the rest that of the second
struct ReduceSide {
  int rows, B, grid;
  const double *partial;
  double *y;
  double gamma;
  const double *v;
  SpmvExtra ex;
};
__global__ __launch_bounds__(kBlock) void k_panel_reduce_pair(ReduceSide a, ReduceSide b, const int *__restrict__ skip) {
  if (skip && *skip) return;
  const bool first = (int)blockIdx.x < a.grid;
  const ReduceSide &m = first ? a : b;
  panel_reduce_rows(first ? (int)blockIdx.x : (int)blockIdx.x - a.grid, m.grid, m.rows, m.B, m.partial, m.y, nullptr, 0.0, m.gamma, m.v, m.ex);
}

// ---------------------------------------------------------------------------------------------------------
// Compact mode (panel_compact): the CSR column / value arrays of a matrix have been released and the sliced-ELL copy
// is the only one.  What the CSR arrays were still used for after setup -- Ruiz passes (row maxima, row / column
// scaling), the Jacobi diagonal, value updates by nnz index -- walks the slices instead: same tiles, same units as the
// product, lane = row, global column = panel base + local id.
// ---------------------------------------------------------------------------------------------------------
// op 0: val <- ((val * a) * b) * scalar with the three orders of k_scale_rows_cols; op 1: partial[g][row] = max |val|;
// op 2: diag[row] = val where column == row + row0
template <typename ColT, int kOp>
__global__ __launch_bounds__(kThreads) void k_sell_visit(int rows, int shift, int B, int Gp, const int *__restrict__ tile_g,
                                                         const int *__restrict__ tile_r0, const int *__restrict__ tile_r1,
                                                         const int *__restrict__ unit_s0, const int *__restrict__ unit_ns,
                                                         const uint32_t *__restrict__ slice_base, const int *__restrict__ slice_len,
                                                         const int *__restrict__ slice_rows, const ColT *__restrict__ scol,
                                                         double *__restrict__ sval, const double *__restrict__ r, const double *__restrict__ c,
                                                         int order, double scalar, int row0, double *__restrict__ out) {
  __shared__ double ys[kOp == 1 ? kTileRowsMax : 1];
  const int t = blockIdx.x, g = tile_g[t];
  const int r0 = tile_r0[t], nrows = tile_r1[t] - r0;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (kOp == 1) for (int i = threadIdx.x; i < nrows; i += kThreads) ys[i] = 0.0;
  for (int j = 0; j < Gp; j++) {
    const int b = g * Gp + j;
    if (b >= B) break;
    const int s0 = unit_s0[(size_t)t * Gp + j], ns = unit_ns[(size_t)t * Gp + j];
    if (kOp == 1) __syncthreads();  // a row appears once per panel; panels are separated by barriers
    const int c0 = b << shift;
    for (int sl = s0 + wave; sl < s0 + ns; sl += kWaves) {
      const size_t base = (size_t)slice_base[sl] + lane;
      const int L = slice_len[sl];
      const int row = slice_rows[(size_t)sl * 64 + lane];
      if (row < 0) continue;
      double mx = 0.0;
      for (int k = 0; k < L; k++) {
        const size_t at = base + (size_t)k * 64;
        double v = sval[at];
        if (kOp == 0) {
          if (r) {
            const int col = c0 + (int)scol[at];
            double a, bb;
            if (order == 1) { const int gi = row + row0; const int lo = col < gi ? col : gi, hi = col < gi ? gi : col; a = c[lo]; bb = c[hi]; }
            else if (order == 2) { a = c[col]; bb = r[row]; }
            else { a = r[row]; bb = c[col]; }
            v = (v * a) * bb;
          }
          if (scalar != 1.0) v *= scalar;
          if (v != 0.0 || sval[at] != 0.0) sval[at] = v;  // padding slots (value 0 at local column 0) stay exact zeros
        } else if (kOp == 1) {
          mx = fmax(mx, fabs(v));
        } else {
          if (c0 + (int)scol[at] == row + row0 && v != 0.0) out[row] = v;
        }
      }
      if (kOp == 1) ys[row - r0] = fmax(ys[row - r0], mx);
    }
  }
  if (kOp == 1) {
    __syncthreads();
    double *o = out + (size_t)g * rows + r0;
    for (int i = threadIdx.x; i < nrows; i += kThreads) o[i] = ys[i];
  }
}
// One pass of a Ruiz iteration over the sliced-ELL copy: val <- (((val * pre) * a) * b) with the three orders of
// k_scale_rows_cols, written back in place, AND partial[g][row] = max |new val| -- the row norms the next iteration
// needs, so that an iteration touches every matrix once (10 B read + 8 B written per entry) instead of once for the
// norms, once for the scaling and, for P, twice more for the cost scaling.  The product kernel's structure: the column
// scaling vector goes through LDS panel by panel (the gathers never leave the CU), lane = row, batches of kBatch
// entries in flight.  pre: a scalar applied FIRST (the cost scaling of the previous iteration, deferred to here --
// ((v c) D_lo) D_hi is exactly what scaling by c in place and by D afterwards gives).
template <typename ColT>
__global__ __launch_bounds__(kThreads) void k_sell_scale_norm(int rows, int cols, int shift, int B, int Gp, const int *__restrict__ tile_g,
                                                              const int *__restrict__ tile_r0, const int *__restrict__ tile_r1,
                                                              const int *__restrict__ unit_s0, const int *__restrict__ unit_ns,
                                                              const uint32_t *__restrict__ slice_base, const int *__restrict__ slice_len,
                                                              const int *__restrict__ slice_rows, const ColT *__restrict__ scol,
                                                              double *__restrict__ sval, const double *__restrict__ r,
                                                              const double *__restrict__ c, int order, double pre, int row0,
                                                              double *__restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int t = blockIdx.x, g = tile_g[t];
  const int W = 1 << shift;
  const int r0 = tile_r0[t], nrows = tile_r1[t] - r0;
  double *cs = lds, *ys = lds + W;
  for (int i = threadIdx.x; i < nrows; i += kThreads) ys[i] = 0.0;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double *rv = order == 1 ? c + row0 : r;  // the row's own factor: c[global row] for the symmetric order
  bool first = true;
  for (int j = 0; j < Gp; j++) {
    const int b = g * Gp + j;
    if (b >= B) break;
    const int s0 = unit_s0[(size_t)t * Gp + j], ns = unit_ns[(size_t)t * Gp + j];
    if (ns == 0) continue;
    const int c0 = b << shift;
    if (!first) __syncthreads();
    const int wlen = cols - c0 < W ? cols - c0 : W;
    if (wlen == 16 * kThreads) {
      const double2 *xg = reinterpret_cast<const double2 *>(c + c0);
      double2 *xl = reinterpret_cast<double2 *>(cs);
      double2 v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = xg[threadIdx.x + k * kThreads];
#pragma unroll
      for (int k = 0; k < 8; k++) xl[threadIdx.x + k * kThreads] = v[k];
    } else {
      for (int i = threadIdx.x; i < wlen; i += kThreads) cs[i] = c[c0 + i];
    }
    __syncthreads();
    first = false;
    int sl = s0 + wave;
    size_t nbase = 0;
    int nL = 0, nrow = -1;
    double nrs = 1.0;
    if (sl < s0 + ns) {
      nbase = (size_t)slice_base[sl]; nL = slice_len[sl]; nrow = slice_rows[(size_t)sl * 64 + lane];
      nrs = nrow >= 0 ? rv[nrow] : 1.0;
    }
    for (; sl < s0 + ns; sl += kWaves) {
      const size_t base = nbase + lane;
      const int L = nL, row = nrow;
      const double rs = nrs;
      if (sl + kWaves < s0 + ns) {
        nbase = (size_t)slice_base[sl + kWaves]; nL = slice_len[sl + kWaves]; nrow = slice_rows[(size_t)(sl + kWaves) * 64 + lane];
        nrs = nrow >= 0 ? rv[nrow] : 1.0;
      }
      double *v = sval + base;
      const ColT *cc_ = scol + base;
      const int grow = row + row0 - c0;  // the row's global id relative to the panel (symmetric order: col < grow <=> column below the diagonal)
      double mx = 0.0;
      for (int k = 0; k < L; k += kBatch) {
        double cv[kBatch];
        ColT cc[kBatch];
        const bool full = k + kBatch <= L;
        if (full) {
#pragma unroll
          for (int u = 0; u < kBatch; u++) { cv[u] = v[(size_t)(k + u) * 64]; cc[u] = cc_[(size_t)(k + u) * 64]; }
        } else {
#pragma unroll
          for (int u = 0; u < kBatch; u++) {
            const bool in = k + u < L;
            cv[u] = in ? v[(size_t)(k + u) * 64] : 0.0;
            cc[u] = in ? cc_[(size_t)(k + u) * 64] : (ColT)0;
          }
        }
#pragma unroll
        for (int u = 0; u < kBatch; u++) {
          double x = cv[u];
          if (pre != 1.0) x *= pre;
          const double cf = cs[cc[u]];
          if (order == 1) { const bool below = (int)cc[u] < grow; x = (x * (below ? cf : rs)) * (below ? rs : cf); }
          else if (order == 2) x = (x * cf) * rs;
          else x = (x * rs) * cf;
          cv[u] = x;
          mx = fmax(mx, fabs(x));
        }
        if (row >= 0) {
          if (full) {
#pragma unroll
            for (int u = 0; u < kBatch; u++) v[(size_t)(k + u) * 64] = cv[u];
          } else {
#pragma unroll
            for (int u = 0; u < kBatch; u++) if (k + u < L) v[(size_t)(k + u) * 64] = cv[u];
          }
        }
      }
      if (row >= 0) ys[row - r0] = fmax(ys[row - r0], mx);  // a row appears once per panel; panels are separated by barriers
    }
  }
  __syncthreads();
  double *out = partial + (size_t)g * rows + r0;
  for (int i = threadIdx.x; i < nrows; i += kThreads) out[i] = ys[i];
}
__global__ __launch_bounds__(kBlock) void k_panel_reduce_max(int rows, int NG, const double *__restrict__ partial, double *__restrict__ out,
                                                             int accumulate) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= rows) return;
  double m = 0.0;
  for (int g = 0; g < NG; g++) m = fmax(m, partial[(size_t)g * rows + i]);
  out[i] = accumulate ? fmax(out[i], m) : m;
}
// slot of every CSR position (the address computation of k_sell_scatter, recorded instead of used)
__global__ __launch_bounds__(kBlock) void k_sell_slot_of_pos(int rows, int shift, const int64_t *__restrict__ rp, const int *__restrict__ col,
                                                             const uint32_t *__restrict__ cellbase, uint32_t *__restrict__ slot) {
  const int lane = threadIdx.x & 63;
  const int64_t row = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  if (row >= rows) return;
  const int64_t s = rp[row], e = rp[row + 1];
  for (int64_t k = s + lane; k < e; k += 64) {
    const int b = col[k] >> shift;
    const int64_t seg = lower_bound_col(col, s, k + 1, b << shift);
    slot[k] = (uint32_t)((size_t)cellbase[(size_t)b * rows + row] + (size_t)(k - seg) * 64);
  }
}

size_t spmv_lds_bytes(int shift) { return (sizeof(double) << shift) + sizeof(double) * kTileRowsMax; }

int64_t read_i64(const int64_t *dev, hipStream_t s) {
  int64_t v = 0;
  HIP_CHECK(hipMemcpyAsync(&v, dev, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  return v;
}

}  // namespace

// panels per workgroup tile: 4 when that still leaves > 1000 workgroups (measured at nnz = 1e9: 4 and 8 are equal,
// 16 is slower), otherwise every panel on its own
static int panel_group_size(const DevCsr &M, int B) {
  const int want = env_int("OSQP_AMD_PANEL_GROUP", 0);
  if (want > 0) return std::min(want, B);
  if ((double)M.nnz / (4.0 * (double)panel_tile_nnz()) >= 1024.0) return std::min(4, B);
  return 1;
}

// 0: plain CSR kernel; 1: panels of 2^shift columns staged in LDS; 2: wide panels of 2^18 columns left to L2
// width of the wide panels: 2^18 columns (2 MB of x, half an L2), up to 2^20 when the rows are so thin that a narrower
// panel holds fewer than ~3.5 entries of a row (measured at n = 4e6 / 150 per row: 2^16..2^18 equal, 2^19 +25 %, 2^20 +55 %;
// at n = 8e6 / 60 per row 2^19 is 5.2 ms against 8.9 ms of the CSR kernel).  0: no width is worth it.
static int wide_shift(const DevCsr &M) {
  if (const int forced = env_int("OSQP_AMD_WIDE_SHIFT", 0)) return forced;
  for (int sh = 18; sh <= 20; sh++) {
    const int Bw = (M.cols + (1 << sh) - 1) >> sh;
    if (Bw < 4) return 0;
    if ((double)M.nnz / ((double)M.rows * Bw) >= 3.5) return sh;
  }
  return 0;
}
static int panel_mode(const DevCsr &M) {
  const int shift = panel_shift();
  if (const char *e = getenv("OSQP_AMD_PANEL")) {
    if (atoi(e) == 0) return 0;
    if (atoi(e) == 2) return M.cols > (1 << shift) ? 1 : 0;
    if (atoi(e) == 3) return M.cols > (1 << shift) ? 2 : 0;  // wide mode forced (tests; width 2^18 unless OSQP_AMD_WIDE_SHIFT)
  }
  // Worth it when the matrix is large enough to be bandwidth-bound, spans at least two panels and its row
  // segments per panel are long enough to pay for the partial sums.  Measured (tools/sweep_spmv.py): at
  // n = 1e6 / 1000 per row 10.8 -> 2.0 ms per SpMV, at n = 1e5 / 100 per row 0.052 -> 0.027 ms.
  if (M.nnz < 2000000 || M.nnz >= 4000000000LL) return 0;  // 32-bit slot offsets
  const int B = (M.cols + (1 << shift) - 1) >> shift;
  if (B >= 2) {
    // the partial sums cost 16 B per (row, group of panels): at least 4 entries behind each; and the slices need rows of
    // more than an entry or two per panel to be worth their padding
    const int Gp = panel_group_size(M, B), NG = (B + Gp - 1) / Gp;
    if ((double)M.nnz / ((double)M.rows * NG) >= 4.0 && (double)M.nnz / ((double)M.rows * B) >= 1.5) return 1;
  }
  // rows too thin for 16384-column panels: wide panels when x is far beyond L2 and the rows fill those
  return wide_shift(M) ? 2 : 0;
}

bool panel_wanted(const DevCsr &M) { return panel_mode(M) != 0; }

void panel_fill(DevCsr &M, bool with_cols, hipStream_t s, uint32_t *slot) {
  DevPanel &P = M.panel;
  if (P.wide)
    OQ_LAUNCH(k_sell_scatter<uint32_t>, dim3(blocks_for((int64_t)M.rows * 64)), dim3(kBlock), 0, s, M.rows, P.shift, M.rowptr.get(),
              M.col.get(), M.val.get(), P.cellbase.get(), P.scol32.get(), P.sval.get(), with_cols ? 1 : 0, slot);
  else
    OQ_LAUNCH(k_sell_scatter<uint16_t>, dim3(blocks_for((int64_t)M.rows * 64)), dim3(kBlock), 0, s, M.rows, P.shift, M.rowptr.get(),
              M.col.get(), M.val.get(), P.cellbase.get(), P.scol.get(), P.sval.get(), with_cols ? 1 : 0, slot);
}

void panel_build(DevCsr &M, hipStream_t s, uint32_t *slot, bool will_compact, const std::function<void()> &after_cols) {
  DevPanel &P = M.panel;
  P.wide = panel_mode(M) == 2;
  P.shift = P.wide ? (wide_shift(M) ? wide_shift(M) : 18) : panel_shift(); P.W = 1 << P.shift;
  if (!P.wide && P.shift > 15) throw PanelRefused(6, "panel width above 2^15 columns is not supported (16-bit local column ids, LDS size)");
  P.B = (M.cols + P.W - 1) >> P.shift;
  const int64_t cells = (int64_t)P.B * M.rows;
  // per-(panel, row) counts and their offsets
  DevBuf<int64_t> cnt((size_t)cells + 1), off((size_t)cells + 1);
  DevBuf<uint32_t> cellsrc((size_t)cells);
  OQ_LAUNCH(k_panel_count, dim3(blocks_for((int64_t)M.rows * 64)), dim3(kBlock), 0, s, M.rows, P.B, P.shift, M.rowptr.get(),
            M.col.get(), cnt.get(), cellsrc.get());
  exclusive_scan(cnt.get(), off.get(), cells, s);
  P.Gp = panel_group_size(M, P.B);
  P.NG = (P.B + P.Gp - 1) / P.Gp;
  const int64_t gcells = (int64_t)P.NG * M.rows;
  // tiles of ~equal non-zero count inside each group (gcnt is reused: counts, costs, then tile-start flags)
  DevBuf<int64_t> gcnt((size_t)gcells + 1), tid((size_t)gcells + 1);
  OQ_LAUNCH(k_group_count, dim3(blocks_for(gcells)), dim3(kBlock), 0, s, M.rows, P.B, P.Gp, gcells, cnt.get(), gcnt.get());
  cnt.release();
  // Mid-size matrices (fewer tiles than twice the compute units): a 65536-entry budget leaves a matrix of 1e7 entries with 153
  // tiles on 256 CUs, and the paired A p / P p launch with one full round and a fifth of one.  A budget that gives the single
  // matrix ~0.85 tiles per CU (rand-1e5: 46000 -> 217 tiles, the pair 437 = two rounds, the second 70 % full) measured +4 % on
  // the rand-1e5 step; below ~42000 the single product falls into a second round and loses more than the pair gains
  // (profiles/r03_rand1e5_tile_sweep.txt).
  int64_t tile_nnz = panel_tile_nnz();
  if (!getenv("OSQP_AMD_PANEL_TILE_NNZ") && P.Gp == 1) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if ((double)M.nnz / (double)tile_nnz < (double)cus)
      tile_nnz = std::max<int64_t>(32768, std::min<int64_t>(tile_nnz, (int64_t)((double)M.nnz / (0.85 * (double)cus))));
  }
  const int64_t budget = std::max<int64_t>(tile_nnz * P.Gp, kTileRowsMax);
  const int64_t cmin = (budget + kTileRowsMax - 1) / kTileRowsMax;
  OQ_LAUNCH(k_tile_cost, dim3(blocks_for(gcells)), dim3(kBlock), 0, s, gcells, cmin, gcnt.get());
  exclusive_scan(gcnt.get(), tid.get(), gcells, s);
  OQ_LAUNCH(k_tile_starts, dim3(blocks_for(gcells)), dim3(kBlock), 0, s, M.rows, gcells, tid.get(), budget, gcnt.get());
  exclusive_scan(gcnt.get(), tid.get(), gcells, s);
  const int64_t ntiles = read_i64(tid.get() + gcells, s);
  if (ntiles <= 0 || ntiles * P.Gp >= 2147483647LL) throw PanelRefused(6, "panel layout: bad tile count");
  P.ntiles = (int)ntiles;
  P.tile_g.alloc((size_t)ntiles); P.tile_r0.alloc((size_t)ntiles); P.tile_r1.alloc((size_t)ntiles);
  OQ_LAUNCH(k_tile_fill, dim3(blocks_for(gcells)), dim3(kBlock), 0, s, M.rows, gcells, gcnt.get(), tid.get(), P.tile_g.get(), P.tile_r0.get());
  OQ_LAUNCH(k_tile_ends, dim3(blocks_for(ntiles)), dim3(kBlock), 0, s, M.rows, P.ntiles, P.tile_g.get(), P.tile_r0.get(), P.tile_r1.get());
  const int64_t nunits = ntiles * P.Gp;
  DevBuf<int> ub((size_t)nunits), u0((size_t)nunits), u1((size_t)nunits);
  OQ_LAUNCH(k_expand_units, dim3(blocks_for(nunits)), dim3(kBlock), 0, s, P.ntiles, P.B, P.Gp, P.tile_g.get(), P.tile_r0.get(),
            P.tile_r1.get(), ub.get(), u0.get(), u1.get());
  HIP_CHECK(hipStreamSynchronize(s));
  gcnt.release(); tid.release();
  // slices: measure every tile, scan, lay out
  DevBuf<int64_t> nsl((size_t)nunits + 1), pad((size_t)nunits + 1), slice0((size_t)nunits + 1), padded0((size_t)nunits + 1);
  OQ_LAUNCH(k_tile_measure, dim3((unsigned)nunits), dim3(kThreads), 0, s, M.rows, ub.get(), u0.get(), u1.get(), off.get(), nsl.get(), pad.get());
  exclusive_scan(nsl.get(), slice0.get(), nunits, s);
  exclusive_scan(pad.get(), padded0.get(), nunits, s);
  const int64_t nslices = read_i64(slice0.get() + nunits, s), padded = read_i64(padded0.get() + nunits, s);
  if (padded >= 4294967295LL) throw PanelRefused(6, "sliced-ELL copy exceeds 2^32 entries");
  // A few very long rows among short ones (a factor model, a budget row) leave slices of 64 lanes with a handful of rows:
  // the copy is mostly padding and the product several times slower than the CSR kernel (portfolio, n = 20 k: 374 us
  // instead of 16 per product).  Unless the layout was asked for by name, such a matrix stays on the CSR kernel.
  if (!getenv("OSQP_AMD_PANEL") && (double)padded > 1.5 * (double)M.nnz) throw PanelRefused(6, "sliced-ELL copy would be mostly padding");
  P.padded = (size_t)padded;
  P.unit_s0.alloc((size_t)nunits); P.unit_ns.alloc((size_t)nunits);
  P.slice_base.alloc((size_t)nslices); P.slice_len.alloc((size_t)nslices); P.slice_rows.alloc((size_t)nslices * 64);
  P.cellbase.alloc((size_t)cells); P.cellbase.zero(s);
  OQ_LAUNCH(k_tile_layout, dim3((unsigned)nunits), dim3(kThreads), 0, s, M.rows, ub.get(), u0.get(), u1.get(), off.get(), slice0.get(),
            padded0.get(), P.unit_s0.get(), P.unit_ns.get(), P.slice_base.get(), P.slice_len.get(), P.slice_rows.get(), P.cellbase.get());
  P.partial.alloc((size_t)gcells);
  P.partial.zero(s);  // every (group, row) cell is rewritten by each product: the zeroes only matter before the first one
  static const bool lds_fill = !(getenv("OSQP_AMD_SELL_FILL") && atoi(getenv("OSQP_AMD_SELL_FILL")) == 0);
  auto fill = [&](int what) {
    if (P.wide)
      OQ_LAUNCH(k_sell_fill_lds<uint32_t>, dim3((unsigned)nunits), dim3(kFillWaves * 64), 0, s, M.rows, P.shift, ub.get(), P.unit_s0.get(),
                P.unit_ns.get(), P.slice_base.get(), P.slice_len.get(), P.slice_rows.get(), cellsrc.get(), off.get(), M.col.get(), M.val.get(),
                P.scol32.get(), P.sval.get(), slot, what);
    else
      OQ_LAUNCH(k_sell_fill_lds<uint16_t>, dim3((unsigned)nunits), dim3(kFillWaves * 64), 0, s, M.rows, P.shift, ub.get(), P.unit_s0.get(),
                P.unit_ns.get(), P.slice_base.get(), P.slice_len.get(), P.slice_rows.get(), cellsrc.get(), off.get(), M.col.get(), M.val.get(),
                P.scol.get(), P.sval.get(), slot, what);
  };
  if (!lds_fill) {
    P.sval.alloc((size_t)padded);
    P.sval.zero(s);  // padding slots: value 0 times x[panel column 0]
    if (P.wide) { P.scol32.alloc((size_t)padded); P.scol32.zero(s); }
    else { P.scol.alloc((size_t)padded); P.scol.zero(s); }
    panel_fill(M, true, s, slot);
  } else if (will_compact && !P.wide) {
    // The matrix gives up its CSR arrays right after this: column ids (and slots) first, then the CSR column array goes BEFORE the
    // slice values are allocated -- 4 B per entry less at the high-water mark of a large setup (rand-1e6: 59.75 -> 54.5 GB) for
    // a second walk over the slice structure.  The fill writes every position of every slice, padding included: no memset.
    P.scol.alloc((size_t)padded);
    fill(1);
    HIP_CHECK(hipStreamSynchronize(s));
    M.col.release();
    if (after_cols) after_cols();  // the slots are all known now: the caller folds them into its maps and lets the temporary go
    P.sval.alloc((size_t)padded);
    fill(2);
  } else {
    P.sval.alloc((size_t)padded);
    if (P.wide) P.scol32.alloc((size_t)padded); else P.scol.alloc((size_t)padded);
    fill(3);
  }
  HIP_CHECK(hipStreamSynchronize(s));
  if (!P.wide)
    HIP_CHECK(hipFuncSetAttribute((const void *)k_spmv_sell<uint16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)spmv_lds_bytes(P.shift)));
  P.active = true;
}

static int reduce_grid(int rows) { return std::min(blocks_for(rows), kReduceBlocks); }

void spmv_panel(const DevCsr &M, const double *x, double *y, const double *rscale, double beta, double gamma, const double *v,
                hipStream_t s, const SpmvExtra *extra) {
  const DevPanel &P = M.panel;
  const SpmvExtra ex = extra ? *extra : SpmvExtra();
  if (P.wide)
    OQ_LAUNCH((k_spmv_sell<uint32_t, false>), dim3(P.ntiles), dim3(kThreads), sizeof(double) * kTileRowsMax, s, M.rows, M.cols, P.shift, P.B,
              P.Gp, P.tile_g.get(), P.tile_r0.get(), P.tile_r1.get(), P.unit_s0.get(), P.unit_ns.get(), P.slice_base.get(),
              P.slice_len.get(), P.slice_rows.get(), P.scol32.get(), P.sval.get(), x, P.partial.get(), g_skip);
  else
    OQ_LAUNCH((k_spmv_sell<uint16_t, true>), dim3(P.ntiles), dim3(kThreads), spmv_lds_bytes(P.shift), s, M.rows, M.cols, P.shift, P.B,
              P.Gp, P.tile_g.get(), P.tile_r0.get(), P.tile_r1.get(), P.unit_s0.get(), P.unit_ns.get(), P.slice_base.get(),
              P.slice_len.get(), P.slice_rows.get(), P.scol.get(), P.sval.get(), x, P.partial.get(), g_skip);
  OQ_LAUNCH(k_panel_reduce, dim3(reduce_grid(M.rows)), dim3(kBlock), 0, s, M.rows, P.NG, P.partial.get(), y, rscale, beta, gamma, v, ex, g_skip);
}

// ya = Ma x (+ extras), yb = Mb x + gamma_b vb: one product launch and one reduce launch for the two
bool spmv_pair_ok(const DevCsr &Ma, const DevCsr &Mb) {
  const DevPanel &A = Ma.panel, &B = Mb.panel;
  static const bool enabled = !(getenv("OSQP_AMD_SPMV_PAIR") && atoi(getenv("OSQP_AMD_SPMV_PAIR")) == 0);
  // worth it while one matrix alone leaves compute units idle for a good part of its launch
  return enabled && A.active && B.active && !A.wide && !B.wide && A.shift == B.shift && Ma.cols == Mb.cols && A.ntiles + B.ntiles <= 2048;
}
void spmv_pair(const DevCsr &Ma, const DevCsr &Mb, const double *x, double *ya, const SpmvExtra *extra_a, double *yb, double gamma_b,
               const double *vb, hipStream_t s) {
  auto side = [](const DevCsr &M) {
    const DevPanel &P = M.panel;
    SellSide t;
    t.rows = M.rows; t.cols = M.cols; t.B = P.B; t.Gp = P.Gp; t.ntiles = P.ntiles;
    t.tile_g = P.tile_g.get(); t.tile_r0 = P.tile_r0.get(); t.tile_r1 = P.tile_r1.get(); t.unit_s0 = P.unit_s0.get(); t.unit_ns = P.unit_ns.get();
    t.slice_len = P.slice_len.get(); t.slice_rows = P.slice_rows.get(); t.slice_base = P.slice_base.get(); t.scol = P.scol.get();
    t.sval = P.sval.get(); t.partial = P.partial.get();
    return t;
  };
  const DevPanel &A = Ma.panel, &B = Mb.panel;
  static bool attr_set = false;
  if (!attr_set) {
    HIP_CHECK(hipFuncSetAttribute((const void *)k_spmv_sell_pair, hipFuncAttributeMaxDynamicSharedMemorySize, (int)spmv_lds_bytes(A.shift)));
    attr_set = true;
  }
  OQ_LAUNCH(k_spmv_sell_pair, dim3(A.ntiles + B.ntiles), dim3(kThreads), spmv_lds_bytes(A.shift), s, side(Ma), side(Mb), A.shift, x, g_skip);
  ReduceSide ra{Ma.rows, A.NG, reduce_grid(Ma.rows), A.partial.get(), ya, 0.0, nullptr, extra_a ? *extra_a : SpmvExtra()};
  ReduceSide rb{Mb.rows, B.NG, reduce_grid(Mb.rows), B.partial.get(), yb, gamma_b, vb, SpmvExtra()};
  OQ_LAUNCH(k_panel_reduce_pair, dim3(ra.grid + rb.grid), dim3(kBlock), 0, s, ra, rb, g_skip);
}

// ---- compact mode: host side -------------------------------------------------------------------------------------------
#define OQ_SELL_VISIT(OP, ...)                                                                                                              \
  do {                                                                                                                                      \
    if (P.wide) OQ_LAUNCH((k_sell_visit<uint32_t, OP>), dim3(P.ntiles), dim3(kThreads), 0, s, M.rows, P.shift, P.B, P.Gp, P.tile_g.get(),    \
                          P.tile_r0.get(), P.tile_r1.get(), P.unit_s0.get(), P.unit_ns.get(), P.slice_base.get(), P.slice_len.get(),        \
                          P.slice_rows.get(), P.scol32.get(), P.sval.get(), __VA_ARGS__);                                                    \
    else OQ_LAUNCH((k_sell_visit<uint16_t, OP>), dim3(P.ntiles), dim3(kThreads), 0, s, M.rows, P.shift, P.B, P.Gp, P.tile_g.get(),           \
                   P.tile_r0.get(), P.tile_r1.get(), P.unit_s0.get(), P.unit_ns.get(), P.slice_base.get(), P.slice_len.get(),               \
                   P.slice_rows.get(), P.scol.get(), P.sval.get(), __VA_ARGS__);                                                             \
  } while (0)

void panel_row_absmax(const DevCsr &M, double *out, bool accumulate, hipStream_t s) {
  const DevPanel &P = M.panel;
  OQ_SELL_VISIT(1, (const double *)nullptr, (const double *)nullptr, 0, 1.0, 0, P.partial.get());
  OQ_LAUNCH(k_panel_reduce_max, dim3(blocks_for(M.rows)), dim3(kBlock), 0, s, M.rows, P.NG, P.partial.get(), out, (int)accumulate);
}
void panel_scale(DevCsr &M, const double *r, const double *c, int order, double scalar, hipStream_t s, int row0) {
  const DevPanel &P = M.panel;
  OQ_SELL_VISIT(0, r, c, order, scalar, row0, (double *)nullptr);
}
// the fused Ruiz pass (k_sell_scale_norm): scale in place, norm[i] = max |row i| of the result
void panel_scale_norm(DevCsr &M, const double *r, const double *c, int order, double pre, int row0, double *norm, hipStream_t s) {
  const DevPanel &P = M.panel;
  if (P.wide) throw Error(6, "internal: the fused scaling pass needs LDS-staged panels");
  HIP_CHECK(hipFuncSetAttribute((const void *)k_sell_scale_norm<uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)spmv_lds_bytes(P.shift)));
  OQ_LAUNCH((k_sell_scale_norm<uint16_t>), dim3(P.ntiles), dim3(kThreads), spmv_lds_bytes(P.shift), s, M.rows, M.cols, P.shift, P.B, P.Gp,
            P.tile_g.get(), P.tile_r0.get(), P.tile_r1.get(), P.unit_s0.get(), P.unit_ns.get(), P.slice_base.get(), P.slice_len.get(),
            P.slice_rows.get(), P.scol.get(), P.sval.get(), r, c, order, pre, row0, P.partial.get());
  OQ_LAUNCH(k_panel_reduce_max, dim3(blocks_for(M.rows)), dim3(kBlock), 0, s, M.rows, P.NG, P.partial.get(), norm, 0);
}
void panel_diag(const DevCsr &M, double *diag, int row0, hipStream_t s) {
  const DevPanel &P = M.panel;
  HIP_CHECK(hipMemsetAsync(diag, 0, sizeof(double) * (size_t)M.rows, s));
  OQ_SELL_VISIT(2, (const double *)nullptr, (const double *)nullptr, 0, 1.0, row0, diag);
}
#undef OQ_SELL_VISIT
// y = (M .* M) x + gamma v  (LDS-staged panels only; the wide mode keeps its CSR arrays)
void spmv_panel_squared(const DevCsr &M, const double *x, double *y, double gamma, const double *v, hipStream_t s) {
  const DevPanel &P = M.panel;
  HIP_CHECK(hipFuncSetAttribute((const void *)k_spmv_sell<uint16_t, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)spmv_lds_bytes(P.shift)));
  OQ_LAUNCH((k_spmv_sell<uint16_t, true, true>), dim3(P.ntiles), dim3(kThreads), spmv_lds_bytes(P.shift), s, M.rows, M.cols, P.shift, P.B,
            P.Gp, P.tile_g.get(), P.tile_r0.get(), P.tile_r1.get(), P.unit_s0.get(), P.unit_ns.get(), P.slice_base.get(),
            P.slice_len.get(), P.slice_rows.get(), P.scol.get(), P.sval.get(), x, P.partial.get(), (const int *)nullptr);
  OQ_LAUNCH(k_panel_reduce, dim3(reduce_grid(M.rows)), dim3(kBlock), 0, s, M.rows, P.NG, P.partial.get(), y, (const double *)nullptr, 0.0, gamma, v,
            SpmvExtra(), (const int *)nullptr);
}
// slot[k] = position of CSR entry k inside sval (needs the CSR arrays: call before panel_compact)
void panel_slot_of_pos(const DevCsr &M, uint32_t *slot, hipStream_t s) {
  const DevPanel &P = M.panel;
  OQ_LAUNCH(k_sell_slot_of_pos, dim3(blocks_for((int64_t)M.rows * 64)), dim3(kBlock), 0, s, M.rows, P.shift, M.rowptr.get(), M.col.get(),
            P.cellbase.get(), slot);
}
bool panel_can_compact(const DevCsr &M) { return M.panel.active && !M.panel.wide; }
void panel_compact(DevCsr &M) {
  M.col.release();
  M.val.release();
  M.panel.cellbase.release();  // only the value refresh from the CSR arrays needed it
  M.compact = true;
}

}  // namespace oq
