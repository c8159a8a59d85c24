// panel.hip -- LDS-staged SpMV for matrices whose x vector does not fit the caches
// (BASELINE.json config 3: n = m = 1e6, 1000 non-zeros per row, x = 8 MB).
//
// The plain CSR kernel (k_spmv) streams the matrix at full width but gathers x with one
// random 8-byte access per non-zero; at n = 1e6 every gather is an L2 / Infinity-Fabric line
// fetch and the kernel runs at ~1.1 TB/s of algorithmic bandwidth (profiles/r01_a_*).
// Here the columns are cut into panels of W = 16384 columns (128 KB of x); a 1024-thread
// workgroup stages its panel of x in LDS once, then streams a tile of R rows of that panel
// (values fp64 + 16-bit local column indices, 8 lanes per row segment) and gathers from LDS.
// Per-panel row sums go to a [B x rows] buffer that a second kernel adds up in panel order
// (fixed order => reproducible) together with the SpMV epilogue.
//
// Algorithmic bytes keep the CSR accounting of SURVEY.md 8d (12 nnz + ...); the panel copy
// actually moves 10 B per non-zero + 4 B per (row, panel) + 16 B per (row, panel) of partials.
#include "kernels.hpp"

namespace oq {

namespace {

constexpr int kPanelShift = 14;            // W = 16384 columns = 128 KB of fp64 in LDS
constexpr int kPanelThreads = 1024;
constexpr int kPanelG = 8;                 // lanes per row segment
constexpr int kPanelRows = 4096;           // rows per workgroup tile

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

// one workgroup = (panel b, row tile t): stage x[b*W .. b*W+W) in LDS, stream the tile, gather from LDS.
// Each 8-lane group walks 4 rows at a time so that >= 4 independent value/index loads are in flight per lane.
__global__ __launch_bounds__(kPanelThreads) void k_spmv_panel(int rows, int cols, int shift, int T, int R,
                                                              const uint32_t *__restrict__ pptr, const uint16_t *__restrict__ pcol,
                                                              const double *__restrict__ pval, const double *__restrict__ x,
                                                              double *__restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) double xs[];
  const int b = blockIdx.x / T, t = blockIdx.x - b * T;
  const int W = 1 << shift;
  const int c0 = b << shift;
  const int wlen = cols - c0 < W ? cols - c0 : W;
  for (int i = threadIdx.x; i < wlen; i += kPanelThreads) xs[i] = x[c0 + i];
  __syncthreads();
  constexpr int G = kPanelG, NG = kPanelThreads / G;
  const int lane = threadIdx.x & (G - 1), grp = threadIdx.x / G;
  const int r0 = t * R, r1 = r0 + R < rows ? r0 + R : rows;
  const uint32_t *pp = pptr + (size_t)b * rows;
  double *out = partial + (size_t)b * rows;
  for (int row = r0 + grp; row < r1; row += 4 * NG) {
    uint32_t s[4], e[4];
    double acc[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int r = row + u * NG;
      const bool ok = r < r1;
      s[u] = ok ? pp[r] : 0u;
      e[u] = ok ? pp[r + 1] : 0u;
      acc[u] = 0.0;
    }
    while (true) {
      bool any = false;
#pragma unroll
      for (int u = 0; u < 4; u++) any |= s[u] < e[u];
      if (!any) break;
      uint16_t cc[4];
      double vv[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t k = s[u] + lane;
        const bool ok = k < e[u];
        cc[u] = ok ? pcol[k] : (uint16_t)0;
        vv[u] = ok ? pval[k] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        acc[u] += vv[u] * xs[cc[u]];
        s[u] = s[u] + G < e[u] ? s[u] + G : e[u];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      double a = acc[u];
#pragma unroll
      for (int o = G >> 1; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
      const int r = row + u * NG;
      if (lane == 0 && r < r1) out[r] = a;
    }
  }
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
  if (const char *e = getenv("OSQP_AMD_PANEL")) { if (atoi(e) == 0) return false; if (atoi(e) == 2) return M.cols > (1 << kPanelShift); }
  // x must be too large for the per-XCD L2 (4 MB) and the row segments per panel long enough to pay for the partial sums
  const int B = (M.cols + (1 << kPanelShift) - 1) >> kPanelShift;
  if ((size_t)M.cols * 8 < (size_t)3 << 20) return false;
  if (M.nnz >= 4000000000LL) return false;  // 32-bit panel offsets
  return (double)M.nnz / ((double)M.rows * B) >= 4.0;
}

void panel_fill(DevCsr &M, bool with_cols, hipStream_t s) {
  DevPanel &P = M.panel;
  OQ_LAUNCH(k_panel_scatter, dim3(blocks_for((int64_t)M.rows * 64)), dim3(kBlock), 0, s, M.rows, P.shift, M.rowptr.get(), M.col.get(),
            M.val.get(), P.pptr.get(), P.pcol.get(), P.pval.get(), with_cols ? 1 : 0);
}

void panel_build(DevCsr &M, hipStream_t s) {
  DevPanel &P = M.panel;
  P.shift = kPanelShift; P.W = 1 << kPanelShift;
  P.B = (M.cols + P.W - 1) >> kPanelShift;
  P.R = kPanelRows; P.T = (M.rows + P.R - 1) / P.R;
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
  P.pcol.alloc((size_t)M.nnz);
  P.pval.alloc((size_t)M.nnz);
  P.partial.alloc(cells);
  panel_fill(M, true, s);
  static bool attr_set = false;
  if (!attr_set) {
    HIP_CHECK(hipFuncSetAttribute((const void *)k_spmv_panel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) << kPanelShift)));
    attr_set = true;
  }
  P.active = true;
}

void spmv_panel(const DevCsr &M, const double *x, double *y, const double *rscale, double beta, double gamma, const double *v,
                hipStream_t s) {
  const DevPanel &P = M.panel;
  OQ_LAUNCH(k_spmv_panel, dim3(P.B * P.T), dim3(kPanelThreads), sizeof(double) << P.shift, s, M.rows, M.cols, P.shift, P.T, P.R,
            P.pptr.get(), P.pcol.get(), P.pval.get(), x, P.partial.get());
  OQ_LAUNCH(k_panel_reduce, dim3(blocks_for(M.rows)), dim3(kBlock), 0, s, M.rows, P.B, P.partial.get(), y, rscale, beta, gamma, v);
}

}  // namespace oq
