// panel_sell.hip -- sliced-ELL layout inside the tiles of the LDS-staged panel SpMV (see panel.hip).
//
// The group-per-row kernel (k_spmv_panel) moves the minimal number of bytes but with 8-byte and
// 2-byte loads on ~16-entry row segments it stays at ~3.3 TB/s of real HBM traffic; a
// lane-contiguous probe over the same bytes reaches 5.4 TB/s (profiles/r01_e_panel_sweep.md).
// Here every tile (rows of one column panel, ~64 K non-zeros) is re-laid as sliced ELL:
//   * the rows of the tile are ordered by their length inside the panel (counting sort, host);
//   * 64 consecutive rows of that order form a slice, stored column-major and padded to the
//     longest row of the slice (sorted => a few per cent of padding): element k of lane l sits at
//     slice_base + 64 k + l, so one load instruction of a wavefront moves 512 contiguous bytes of
//     values or 128 contiguous bytes of 16-bit column indices;
//   * lane = row: no shuffles, no reduction; the lane adds its row in ascending column order and
//     writes the per-panel row sum.  Rows with no entry in the panel are not stored at all (their
//     cell of the partial-sum buffer is zeroed once at build time).
// The 16 wavefronts of a workgroup share the LDS copy of the x panel and take slices round-robin.
#include <algorithm>

#include "kernels.hpp"

namespace oq {

namespace {

constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;
constexpr int kTileRowsMax = 3968;  // row sums of a tile are staged in LDS (31 KB next to the 128 KB x panel)

__device__ __forceinline__ int64_t lb_col(const int *__restrict__ col, int64_t s, int64_t e, int target) {
  while (s < e) { int64_t mid = (s + e) >> 1; if (col[mid] < target) s = mid + 1; else e = mid; }
  return s;
}

// copy every entry of the CSR matrix to its sliced-ELL slot; with_cols = 0 refreshes the values only
__global__ __launch_bounds__(kBlock) void k_sell_scatter(int rows, int shift, const int64_t *__restrict__ rp, const int *__restrict__ col,
                                                         const double *__restrict__ val, const uint32_t *__restrict__ cellbase,
                                                         uint16_t *__restrict__ scol, double *__restrict__ sval, int with_cols) {
  const int lane = threadIdx.x & 63;
  const int64_t row = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  if (row >= rows) return;
  const int64_t s = rp[row], e = rp[row + 1];
  const int mask = (1 << shift) - 1;
  for (int64_t k = s + lane; k < e; k += 64) {
    const int c = col[k];
    const int b = c >> shift;
    const int64_t seg = lb_col(col, s, k + 1, b << shift);  // first entry of this row in panel b
    const size_t dst = (size_t)cellbase[(size_t)b * rows + row] + (size_t)(k - seg) * 64;
    sval[dst] = val[k];
    if (with_cols) scol[dst] = (uint16_t)(c & mask);
  }
}

__global__ __launch_bounds__(kThreads) void k_spmv_sell(int rows, int cols, int shift, const int *__restrict__ tile_b,
                                                        const int *__restrict__ tile_r0, const int *__restrict__ tile_r1,
                                                        const int *__restrict__ tile_s0, const int *__restrict__ tile_ns,
                                                        const uint32_t *__restrict__ slice_base, const int *__restrict__ slice_len,
                                                        const int *__restrict__ slice_rows, const uint16_t *__restrict__ scol,
                                                        const double *__restrict__ sval, const double *__restrict__ x,
                                                        double *__restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) double xs[];
  const int b = tile_b[blockIdx.x], s0 = tile_s0[blockIdx.x], ns = tile_ns[blockIdx.x];
  const int W = 1 << shift;
  const int c0 = b << shift;
  const int wlen = cols - c0 < W ? cols - c0 : W;
  const int r0 = tile_r0[blockIdx.x], nrows = tile_r1[blockIdx.x] - r0;
  double *ys = xs + W;  // row sums of the tile: written scattered here, stored to HBM as one contiguous block
  for (int i = threadIdx.x; i < wlen; i += kThreads) xs[i] = x[c0 + i];
  for (int i = threadIdx.x; i < nrows; i += kThreads) ys[i] = 0.0;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double *out = partial + (size_t)b * rows + r0;
  for (int sl = s0 + wave; sl < s0 + ns; sl += kWaves) {
    const size_t base = (size_t)slice_base[sl] + lane;
    const int L = slice_len[sl];
    const int row = slice_rows[(size_t)sl * 64 + lane];
    const double *v = sval + base;
    const uint16_t *c = scol + base;
    double a0 = 0.0, a1 = 0.0;
    int k = 0;
    for (; k + 4 <= L; k += 4) {
      const double v0 = v[(size_t)k * 64], v1 = v[(size_t)(k + 1) * 64], v2 = v[(size_t)(k + 2) * 64], v3 = v[(size_t)(k + 3) * 64];
      const uint16_t c0_ = c[(size_t)k * 64], c1_ = c[(size_t)(k + 1) * 64], c2_ = c[(size_t)(k + 2) * 64], c3_ = c[(size_t)(k + 3) * 64];
      a0 += v0 * xs[c0_]; a0 += v1 * xs[c1_]; a0 += v2 * xs[c2_]; a0 += v3 * xs[c3_];
    }
    for (; k < L; k++) a1 += v[(size_t)k * 64] * xs[c[(size_t)k * 64]];
    if (row >= 0) ys[row - r0] = a0 + a1;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nrows; i += kThreads) out[i] = ys[i];
}

}  // namespace

// Build the sliced-ELL copy from the panel offsets (hp = host copy of pptr) and the tiles.
void panel_sell_prepare(DevCsr &M, const std::vector<uint32_t> &hp, const std::vector<int> &tb, const std::vector<int> &t0,
                        const std::vector<int> &t1, hipStream_t s) {
  DevPanel &P = M.panel;
  const size_t cells = (size_t)P.B * M.rows;
  std::vector<uint32_t> cellbase(cells, 0u), sbase;
  std::vector<int> tsub0(tb.size()), tnsub(tb.size()), slen, srows;
  std::vector<int> order, bucket;
  size_t padded = 0;
  for (size_t t = 0; t < tb.size(); t++) {
    const uint32_t *pp = hp.data() + (size_t)tb[t] * M.rows;
    const int r0 = t0[t], r1 = t1[t];
    // counting sort of the tile's rows by length, longest first, stable
    int maxlen = 0;
    for (int r = r0; r < r1; r++) maxlen = std::max(maxlen, (int)(pp[r + 1] - pp[r]));
    bucket.assign((size_t)maxlen + 2, 0);
    for (int r = r0; r < r1; r++) bucket[maxlen - (int)(pp[r + 1] - pp[r]) + 1]++;
    for (int i = 0; i <= maxlen; i++) bucket[i + 1] += bucket[i];
    order.resize(r1 - r0);
    for (int r = r0; r < r1; r++) order[bucket[maxlen - (int)(pp[r + 1] - pp[r])]++] = r;
    tsub0[t] = (int)slen.size();
    int count = 0;
    for (size_t i = 0; i < order.size(); i += 64) {
      const int L = (int)(pp[order[i] + 1] - pp[order[i]]);
      if (L == 0) break;  // the rest of the tile has no entry in this panel
      if (padded + (size_t)L * 64 >= 4294967295ULL) throw Error(6, "sliced-ELL copy exceeds 2^32 entries");
      sbase.push_back((uint32_t)padded);
      slen.push_back(L);
      for (int l = 0; l < 64; l++) {
        int r = -1;
        if (i + l < order.size() && pp[order[i + l] + 1] - pp[order[i + l]] > 0) r = order[i + l];
        srows.push_back(r);
        if (r >= 0) cellbase[(size_t)tb[t] * M.rows + r] = (uint32_t)(padded + l);
      }
      padded += (size_t)L * 64;
      count++;
    }
    tnsub[t] = count;
  }
  P.padded = padded;
  auto up = [&](DevBuf<int> &d, const std::vector<int> &h) { d.alloc(h.size()); d.upload(h.data(), h.size(), s); };
  up(P.tile_sub0, tsub0); up(P.tile_nsub, tnsub); up(P.sub_row, slen); up(P.slice_rows, srows);
  P.sub_k.alloc(sbase.size()); P.sub_k.upload(sbase.data(), sbase.size(), s);
  P.cellbase.alloc(cells); P.cellbase.upload(cellbase.data(), cells, s);
  P.sval.alloc(padded); P.scol.alloc(padded);
  P.sval.zero(s); P.scol.zero(s);
  HIP_CHECK(hipStreamSynchronize(s));
  for (size_t t = 0; t < tb.size(); t++)
    if (t1[t] - t0[t] > kTileRowsMax) throw Error(6, "internal: tile taller than the LDS row-sum buffer");
  HIP_CHECK(hipFuncSetAttribute((const void *)k_spmv_sell, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)((sizeof(double) << P.shift) + sizeof(double) * kTileRowsMax)));
  P.sell = true;
}

void panel_sell_fill(DevCsr &M, bool with_cols, hipStream_t s) {
  DevPanel &P = M.panel;
  OQ_LAUNCH(k_sell_scatter, dim3(blocks_for((int64_t)M.rows * 64)), dim3(kBlock), 0, s, M.rows, P.shift, M.rowptr.get(), M.col.get(),
            M.val.get(), P.cellbase.get(), P.scol.get(), P.sval.get(), with_cols ? 1 : 0);
}

void spmv_panel_sell(const DevCsr &M, const double *x, hipStream_t s) {
  const DevPanel &P = M.panel;
  OQ_LAUNCH(k_spmv_sell, dim3(P.ntiles), dim3(kThreads), (sizeof(double) << P.shift) + sizeof(double) * kTileRowsMax, s, M.rows,
            M.cols, P.shift, P.tile_b.get(), P.tile_r0.get(), P.tile_r1.get(), P.tile_sub0.get(), P.tile_nsub.get(), P.sub_k.get(), P.sub_row.get(), P.slice_rows.get(), P.scol.get(), P.sval.get(), x,
            P.partial.get());
}

}  // namespace oq
