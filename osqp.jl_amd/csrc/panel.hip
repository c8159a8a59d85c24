// panel.hip -- LDS-staged SpMV for matrices whose x vector does not fit the caches
// (BASELINE.json config 3: n = m = 1e6, 1000 non-zeros per row, x = 8 MB).
//
// The plain CSR kernel (k_spmv) streams the matrix at full width but gathers x with one
// random 8-byte access per non-zero; at n = 1e6 every gather is an L2 / Infinity-Fabric line
// fetch and the kernel runs at ~1.1 TB/s of algorithmic bandwidth (profiles/r01_a_*).
// Here the columns are cut into panels of W = 2^shift columns (default 16384 = 128 KB of x); a
// 1024-thread workgroup stages its panel of x in LDS once, then streams a tile of rows of that
// panel (values fp64 + 16-bit local column indices, G lanes per row segment) and gathers from
// LDS.  Tiles are cut at equal non-zero counts (not equal row counts) so that dense corners
// do not leave a tail.  Per-panel row sums go to a [B x rows] buffer that a second kernel adds
// up in panel order (fixed order => reproducible) together with the SpMV epilogue.
//
// Algorithmic bytes keep the CSR accounting of SURVEY.md 8d (12 nnz + ...); the panel copy
// actually moves 10 B per non-zero + 4 B per (row, panel) + 16 B per (row, panel) of partials.
#include <algorithm>

#include "kernels.hpp"

namespace oq {

namespace {

constexpr int kPanelThreads = 1024;
// tunables (defaults from the sweep in profiles/r01_e_panel_sweep.md; overridable for experiments)
int env_int(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
int panel_shift() { return env_int("OSQP_AMD_PANEL_SHIFT", 14); }        // W = 16384 columns = 128 KB of fp64 in LDS
int panel_tile_nnz() { return env_int("OSQP_AMD_PANEL_TILE_NNZ", 65536); }  // non-zeros per workgroup tile
int panel_group() { return env_int("OSQP_AMD_PANEL_G", 8); }             // lanes per row segment

// first position in [s, e) with col >= target (cols ascending inside a row)
__device__ __forceinline__ int64_t lower_bound_col(const int *__restrict__ col, int64_t s, int64_t e, int target) {
  while (s < e) { int64_t mid = (s + e) >> 1; if (col[mid] < target) s = mid + 1; else e = mid; }
  return s;
}

// cnt[b * rows + i] = entries of row i inside panel b   (one wavefront per row, lane = panel)
__global__ __launch_bounds__(kBlock) void k_panel_count(int rows, int B, int shift, const int64_t *__restrict__ rp,
                                                        const int *__restrict__ col, int64_t *__restrict__ cnt) {
  const int lane = threadIdx.x & 63;
  const int64_t row = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  if (row >= rows) return;
  const int64_t s = rp[row], e = rp[row + 1];
  for (int b = lane; b < B; b += 64) {
    int64_t lo = lower_bound_col(col, s, e, b << shift);
    int64_t hi = (b + 1 == B) ? e : lower_bound_col(col, s, e, (b + 1) << shift);
    cnt[(size_t)b * rows + row] = hi - lo;
  }
}
__global__ __launch_bounds__(kBlock) void k_to_u32(int64_t n, const int64_t *__restrict__ in, uint32_t *__restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) out[i] = (uint32_t)in[i];
}
// copy every entry to its (panel, row) slot; with_cols = 0 refreshes the values only
__global__ __launch_bounds__(kBlock) void k_panel_scatter(int rows, int shift, const int64_t *__restrict__ rp, const int *__restrict__ col,
                                                          const double *__restrict__ val, const uint32_t *__restrict__ pptr,
                                                          uint16_t *__restrict__ pcol, double *__restrict__ pval, int with_cols) {
  const int lane = threadIdx.x & 63;
  const int64_t row = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  if (row >= rows) return;
  const int64_t s = rp[row], e = rp[row + 1];
  const int mask = (1 << shift) - 1;
  for (int64_t k = s + lane; k < e; k += 64) {
    const int c = col[k];
    const int b = c >> shift;
    const int64_t seg = lower_bound_col(col, s, k + 1, b << shift);  // first entry of this row in panel b
    const size_t dst = (size_t)pptr[(size_t)b * rows + row] + (size_t)(k - seg);
    pval[dst] = val[k];
    if (with_cols) pcol[dst] = (uint16_t)(c & mask);
  }
}

// One workgroup = one tile (rows [r0, r1) of panel b): stage x[b*W .. b*W+W) in LDS, stream the tile,
// gather from LDS.  Each G-lane group walks 4 rows at a time and 2 chunks of G entries per row per pass
// (8 independent value/index loads in flight per lane); the row bounds of the next batch are fetched
// while the current one is processed.
template <int G>
__global__ __launch_bounds__(kPanelThreads) void k_spmv_panel(int rows, int cols, int shift, const int *__restrict__ tile_b,
                                                              const int *__restrict__ tile_r0, const int *__restrict__ tile_r1,
                                                              const uint32_t *__restrict__ pptr, const uint16_t *__restrict__ pcol,
                                                              const double *__restrict__ pval, const double *__restrict__ x,
                                                              double *__restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) double xs[];
  const int b = tile_b[blockIdx.x], r0 = tile_r0[blockIdx.x], r1 = tile_r1[blockIdx.x];
  const int W = 1 << shift;
  const int c0 = b << shift;
  const int wlen = cols - c0 < W ? cols - c0 : W;
  for (int i = threadIdx.x; i < wlen; i += kPanelThreads) xs[i] = x[c0 + i];
  __syncthreads();
  constexpr int NG = kPanelThreads / G;
  const int lane = threadIdx.x & (G - 1), grp = threadIdx.x / G;
  const uint32_t *pp = pptr + (size_t)b * rows;
  double *out = partial + (size_t)b * rows;
  uint32_t s[4], e[4];
  int row = r0 + grp;
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int r = row + u * NG;
    const bool ok = r < r1;
    s[u] = ok ? pp[r] : 0u;
    e[u] = ok ? pp[r + 1] : 0u;
  }
  while (row < r1) {
    const int nrow = row + 4 * NG;
    uint32_t ns[4], ne[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int r = nrow + u * NG;
      const bool ok = r < r1;
      ns[u] = ok ? pp[r] : 0u;
      ne[u] = ok ? pp[r + 1] : 0u;
    }
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    while (true) {
      bool any = false;
#pragma unroll
      for (int u = 0; u < 4; u++) any |= s[u] < e[u];
      if (!any) break;
      uint16_t ca[4], cb[4];
      double va[4], vb[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t k0 = s[u] + lane, k1 = k0 + G;
        const bool ok0 = k0 < e[u], ok1 = k1 < e[u];
        ca[u] = ok0 ? pcol[k0] : (uint16_t)0;
        va[u] = ok0 ? pval[k0] : 0.0;
        cb[u] = ok1 ? pcol[k1] : (uint16_t)0;
        vb[u] = ok1 ? pval[k1] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        acc[u] += va[u] * xs[ca[u]];
        acc[u] += vb[u] * xs[cb[u]];
        s[u] = s[u] + 2 * G < e[u] ? s[u] + 2 * G : e[u];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      double a = acc[u];
#pragma unroll
      for (int o = G >> 1; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
      const int r = row + u * NG;
      if (lane == 0 && r < r1) out[r] = a;
      s[u] = ns[u]; e[u] = ne[u];
    }
    row = nrow;
  }
}


// Bandwidth probe (measurement only, osqp_amd_time_kernel which = 6): same tiles, same LDS staging of x, but the
// tile's non-zero range is streamed lane-contiguously (16 B of values + 4 B of indices per lane per load) with no
// row structure -- the ceiling a CSR-stream variant of the kernel could reach.
__global__ __launch_bounds__(kPanelThreads) void k_panel_stream_probe(int rows, int cols, int shift, const int *__restrict__ tile_b,
                                                                      const int *__restrict__ tile_r0, const int *__restrict__ tile_r1,
                                                                      const uint32_t *__restrict__ pptr, const uint16_t *__restrict__ pcol,
                                                                      const double *__restrict__ pval, const double *__restrict__ x,
                                                                      double *__restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) double xs[];
  const int b = tile_b[blockIdx.x], r0 = tile_r0[blockIdx.x], r1 = tile_r1[blockIdx.x];
  const int W = 1 << shift;
  const int c0 = b << shift;
  const int wlen = cols - c0 < W ? cols - c0 : W;
  for (int i = threadIdx.x; i < wlen; i += kPanelThreads) xs[i] = x[c0 + i];
  __syncthreads();
  const uint32_t *pp = pptr + (size_t)b * rows;
  const uint32_t k0 = pp[r0] & ~1u, k1 = pp[r1];
  double acc = 0.0;
  for (uint32_t k = k0 + 2 * threadIdx.x; k + 1 < k1; k += 2 * kPanelThreads) {
    const double2 v = *reinterpret_cast<const double2 *>(pval + k);
    const ushort2 c = *reinterpret_cast<const ushort2 *>(pcol + k);
    acc += v.x * xs[c.x] + v.y * xs[c.y];
  }
  if (r0 + (int)threadIdx.x < r1) partial[(size_t)b * rows + r0 + threadIdx.x] = acc;
}

// y[i] = (rscale ? rscale[i] : 1) * sum_b partial[b][i] + beta * y[i] + gamma * v[i]   (panel order is fixed)
__global__ __launch_bounds__(kBlock) void k_panel_reduce(int rows, int B, const double *__restrict__ partial, double *__restrict__ y,
                                                         const double *__restrict__ rscale, double beta, double gamma,
                                                         const double *__restrict__ v) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= rows) return;
  double acc = 0.0;
  for (int b = 0; b < B; b++) acc += partial[(size_t)b * rows + i];
  if (rscale) acc *= rscale[i];
  if (beta != 0.0) acc += beta * y[i];
  if (v) acc += gamma * v[i];
  y[i] = acc;
}

}  // namespace

bool panel_wanted(const DevCsr &M) {
  const int shift = panel_shift();
  if (const char *e = getenv("OSQP_AMD_PANEL")) { if (atoi(e) == 0) return false; if (atoi(e) == 2) return M.cols > (1 << shift); }
  // Worth it when the matrix is large enough to be bandwidth-bound, spans at least two panels and its row
  // segments per panel are long enough to pay for the partial sums.  Measured (tools/sweep_spmv.py): at
  // n = 1e6 / 1000 per row 10.8 -> 2.4 ms per SpMV, at n = 1e5 / 100 per row 0.052 -> 0.028 ms.
  const int B = (M.cols + (1 << shift) - 1) >> shift;
  if (B < 2 || M.nnz < 2000000) return false;
  if (M.nnz >= 4000000000LL) return false;  // 32-bit panel offsets
  return (double)M.nnz / ((double)M.rows * B) >= 4.0;
}

void panel_fill(DevCsr &M, bool with_cols, hipStream_t s) {
  DevPanel &P = M.panel;
  if (P.sell) { panel_sell_fill(M, with_cols, s); return; }
  OQ_LAUNCH(k_panel_scatter, dim3(blocks_for((int64_t)M.rows * 64)), dim3(kBlock), 0, s, M.rows, P.shift, M.rowptr.get(), M.col.get(),
            M.val.get(), P.pptr.get(), P.pcol.get(), P.pval.get(), with_cols ? 1 : 0);
}

void panel_build(DevCsr &M, hipStream_t s) {
  DevPanel &P = M.panel;
  P.shift = panel_shift(); P.W = 1 << P.shift;
  P.B = (M.cols + P.W - 1) >> P.shift;
  const size_t cells = (size_t)P.B * M.rows;
  {
    DevBuf<int64_t> cnt(cells + 1), ptr(cells + 1);
    OQ_LAUNCH(k_panel_count, dim3(blocks_for((int64_t)M.rows * 64)), dim3(kBlock), 0, s, M.rows, P.B, P.shift, M.rowptr.get(),
              M.col.get(), cnt.get());
    exclusive_scan(cnt.get(), ptr.get(), (int64_t)cells, s);
    P.pptr.alloc(cells + 1);
    OQ_LAUNCH(k_to_u32, dim3(blocks_for((int64_t)cells + 1)), dim3(kBlock), 0, s, (int64_t)cells + 1, ptr.get(), P.pptr.get());
    HIP_CHECK(hipStreamSynchronize(s));
  }
  // tiles of ~equal non-zero count inside each panel (host: one pass over the offsets)
  {
    std::vector<uint32_t> hp(cells + 1);
    P.pptr.download(hp.data(), cells + 1, s);
    HIP_CHECK(hipStreamSynchronize(s));
    const uint32_t budget = (uint32_t)panel_tile_nnz();
    const int max_rows = 3968;  // = kTileRowsMax of panel_sell.hip (row sums of a tile are staged in LDS)
    std::vector<int> tb, t0, t1;
    for (int b = 0; b < P.B; b++) {
      const uint32_t *pp = hp.data() + (size_t)b * M.rows;
      int r = 0;
      while (r < M.rows) {
        const uint32_t *lim = std::upper_bound(pp + r, pp + M.rows + 1, pp[r] + budget);
        int r_end = (int)(lim - pp) - 1;           // last row boundary with offset <= start + budget
        if (r_end <= r) r_end = r + 1;             // a single row longer than the budget
        if (r_end > M.rows) r_end = M.rows;
        if (r_end - r > max_rows) r_end = r + max_rows;
        tb.push_back(b); t0.push_back(r); t1.push_back(r_end);
        r = r_end;
      }
    }
    // 1 (default): sliced-ELL tiles (panel_sell.hip); 0: panel-CSR tiles with one lane group per row segment
    if (env_int("OSQP_AMD_PANEL_KERNEL", 1) == 1) panel_sell_prepare(M, hp, tb, t0, t1, s);
    P.ntiles = (int)tb.size();
    P.tile_b.alloc(tb.size()); P.tile_r0.alloc(tb.size()); P.tile_r1.alloc(tb.size());
    P.tile_b.upload(tb.data(), tb.size(), s); P.tile_r0.upload(t0.data(), t0.size(), s); P.tile_r1.upload(t1.data(), t1.size(), s);
    HIP_CHECK(hipStreamSynchronize(s));
  }
  if (!P.sell) { P.pcol.alloc((size_t)M.nnz); P.pval.alloc((size_t)M.nnz); }
  P.partial.alloc(cells);
  P.partial.zero(s);  // cells of rows without entries in a panel are never written again
  panel_fill(M, true, s);
  const int lds = (int)(sizeof(double) << P.shift);
  HIP_CHECK(hipFuncSetAttribute((const void *)k_spmv_panel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  HIP_CHECK(hipFuncSetAttribute((const void *)k_spmv_panel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  HIP_CHECK(hipFuncSetAttribute((const void *)k_spmv_panel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  P.active = true;
}

void spmv_panel(const DevCsr &M, const double *x, double *y, const double *rscale, double beta, double gamma, const double *v,
                hipStream_t s) {
  const DevPanel &P = M.panel;
  static const int G = panel_group();
  const size_t lds = sizeof(double) << P.shift;
#define OQ_PANEL(GG)                                                                                                          \
  OQ_LAUNCH(k_spmv_panel<GG>, dim3(P.ntiles), dim3(kPanelThreads), lds, s, M.rows, M.cols, P.shift, P.tile_b.get(), P.tile_r0.get(), \
            P.tile_r1.get(), P.pptr.get(), P.pcol.get(), P.pval.get(), x, P.partial.get())
  if (P.sell) spmv_panel_sell(M, x, s);
  else if (G == 4) OQ_PANEL(4); else if (G == 16) OQ_PANEL(16); else OQ_PANEL(8);
#undef OQ_PANEL
  OQ_LAUNCH(k_panel_reduce, dim3(blocks_for(M.rows)), dim3(kBlock), 0, s, M.rows, P.B, P.partial.get(), y, rscale, beta, gamma, v);
}

void spmv_panel_probe(const DevCsr &M, const double *x, hipStream_t s) {
  const DevPanel &P = M.panel;
  const size_t lds = sizeof(double) << P.shift;
  static bool once = false;
  if (!once) { HIP_CHECK(hipFuncSetAttribute((const void *)k_panel_stream_probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); once = true; }
  OQ_LAUNCH(k_panel_stream_probe, dim3(P.ntiles), dim3(kPanelThreads), lds, s, M.rows, M.cols, P.shift, P.tile_b.get(), P.tile_r0.get(),
            P.tile_r1.get(), P.pptr.get(), P.pcol.get(), P.pval.get(), x, P.partial.get());
}

}  // namespace oq
