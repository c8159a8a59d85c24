// panel_stream.hip -- streaming variant of the LDS-staged panel SpMV (see panel.hip for the format).
//
// k_spmv_panel (one G-lane group per row segment) is bound by the shape of its loads: 8-byte and
// 2-byte accesses to short row segments keep it at ~3.3 TB/s of real HBM traffic although the
// traffic itself is minimal (FETCH_SIZE = 11.2 GB per launch for 10^9 non-zeros, PMC pass in
// profiles/).  A lane-contiguous probe over the same bytes reaches 5.4 TB/s.  This kernel gets
// that access shape and still produces row sums, without workgroup barriers in the main loop:
//   * a tile (rows of one panel, ~64 K non-zeros) is cut on the host into row-aligned mini-chunks
//     of at most 128 non-zeros and < 64 rows; the 16 wavefronts of the workgroup take the
//     mini-chunks of the tile round-robin and run independently of each other;
//   * phase A: the wavefront streams its mini-chunk lane-contiguously (16 B of values + 4 B of
//     16-bit column indices per lane), gathers x from the workgroup's LDS copy of the panel and
//     leaves the products in its private 1 KB LDS buffer, together with the row offsets;
//   * phase B: 8 lanes per row add the row's products from LDS and write the per-panel row sum;
//   * the global loads of the next kDepth mini-chunks are already in flight (register prefetch).
// A row with more than 128 non-zeros inside one panel forms a mini-chunk of its own and is
// reduced by its wavefront in several passes.
#include <algorithm>

#include "kernels.hpp"

namespace oq {

namespace {

constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;
constexpr int kMini = 128;       // non-zeros per mini-chunk = 2 per lane
constexpr int kMiniRows = 63;    // rows per mini-chunk (row offsets are fetched by lanes 0..nrows)
constexpr int kDepth = 3;        // mini-chunks prefetched in registers

struct Regs { double v0, v1; unsigned short c0, c1; uint32_t off, kb, ke; int ra, nrows; };

__device__ __forceinline__ void fetch(Regs &r, int j, int nmc, const int *__restrict__ mrow, const uint32_t *__restrict__ mk,
                                      const uint32_t *__restrict__ pp, const uint16_t *__restrict__ pcol,
                                      const double *__restrict__ pval, int lane) {
  r.v0 = 0.0; r.v1 = 0.0; r.c0 = 0; r.c1 = 0; r.off = 0; r.kb = 0; r.ke = 0; r.ra = 0; r.nrows = 0;
  if (j >= nmc) return;
  r.kb = mk[j]; r.ke = mk[j + 1];
  r.ra = mrow[j]; r.nrows = mrow[j + 1] - r.ra;
  if (r.ke - r.kb <= (uint32_t)kMini) {
    const uint32_t k = r.kb + 2 * lane;
    if (k + 1 < r.ke) {
      r.v0 = pval[k]; r.v1 = pval[k + 1];
      r.c0 = pcol[k]; r.c1 = pcol[k + 1];
    } else if (k < r.ke) {
      r.v0 = pval[k]; r.c0 = pcol[k];
    }
  }
  if (lane <= r.nrows) r.off = pp[r.ra + lane];
}

__global__ __launch_bounds__(kThreads) void k_spmv_stream(int rows, int cols, int shift, const int *__restrict__ tile_b,
                                                          const int *__restrict__ tile_sub0, const int *__restrict__ tile_nsub,
                                                          const int *__restrict__ sub_row, const uint32_t *__restrict__ sub_k,
                                                          const uint32_t *__restrict__ pptr, const uint16_t *__restrict__ pcol,
                                                          const double *__restrict__ pval, const double *__restrict__ x,
                                                          double *__restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int W = 1 << shift;
  double *xs = lds;                                            // W doubles, shared by the workgroup
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double *prod = xs + W + wave * kMini;                        // kMini doubles per wavefront
  uint32_t *roff = (uint32_t *)(xs + W + kWaves * kMini) + wave * 64;  // 64 row offsets per wavefront

  const int b = tile_b[blockIdx.x], mc0 = tile_sub0[blockIdx.x], nmc = tile_nsub[blockIdx.x];
  const int c0 = b << shift;
  const int wlen = cols - c0 < W ? cols - c0 : W;
  for (int i = threadIdx.x; i < wlen; i += kThreads) xs[i] = x[c0 + i];
  __syncthreads();
  const uint32_t *pp = pptr + (size_t)b * rows;
  double *out = partial + (size_t)b * rows;
  const int *mrow = sub_row + mc0;
  const uint32_t *mk = sub_k + mc0;

  Regs r0, r1, r2;
  fetch(r0, wave, nmc, mrow, mk, pp, pcol, pval, lane);
  fetch(r1, wave + kWaves, nmc, mrow, mk, pp, pcol, pval, lane);
  fetch(r2, wave + 2 * kWaves, nmc, mrow, mk, pp, pcol, pval, lane);
  const int lane8 = lane & 7, grp = lane >> 3;
  for (int j = wave; j < nmc; j += kWaves) {
    const uint32_t kb = r0.kb, ke = r0.ke;
    const int ra = r0.ra, nrows = r0.nrows;
    if (ke - kb <= (uint32_t)kMini) {
      // phase A: products and row offsets of this mini-chunk into the wavefront's LDS buffer
      const uint32_t k = kb + 2 * lane;
      if (k < ke) prod[2 * lane] = r0.v0 * xs[r0.c0];
      if (k + 1 < ke) prod[2 * lane + 1] = r0.v1 * xs[r0.c1];
      if (lane <= nrows) roff[lane] = r0.off - kb;
      r0 = r1; r1 = r2;
      fetch(r2, j + kDepth * kWaves, nmc, mrow, mk, pp, pcol, pval, lane);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // phase B: 8 lanes per row
      for (int rr = grp; rr < nrows; rr += 8) {
        const uint32_t s = roff[rr], e = roff[rr + 1];
        double a = 0.0;
        for (uint32_t t = s + lane8; t < e; t += 8) a += prod[t];
        a += __shfl_xor(a, 1, 64);
        a += __shfl_xor(a, 2, 64);
        a += __shfl_xor(a, 4, 64);
        if (lane8 == 0) out[ra + rr] = a;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    } else {
      // one long row: the wavefront streams it in passes and reduces
      r0 = r1; r1 = r2;
      fetch(r2, j + kDepth * kWaves, nmc, mrow, mk, pp, pcol, pval, lane);
      double a = 0.0;
      for (uint32_t k = kb + lane; k < ke; k += 64) a += pval[k] * xs[pcol[k]];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
      if (lane == 0) out[ra] = a;
    }
  }
}

size_t stream_lds_bytes(int shift) { return (sizeof(double) << shift) + sizeof(double) * kWaves * kMini + 4 * kWaves * 64 + 64; }

}  // namespace

// cut the tiles into row-aligned mini-chunks; hp = host copy of pptr
void panel_stream_prepare(DevCsr &M, const std::vector<uint32_t> &hp, std::vector<int> &tb, std::vector<int> &t0, std::vector<int> &t1,
                          hipStream_t s) {
  DevPanel &P = M.panel;
  std::vector<int> tsub0(tb.size()), tnsub(tb.size()), srow;
  std::vector<uint32_t> sk;
  srow.reserve((size_t)(M.nnz / 96) + tb.size() * 2 + 16);
  sk.reserve(srow.capacity());
  for (size_t t = 0; t < tb.size(); t++) {
    const uint32_t *pp = hp.data() + (size_t)tb[t] * M.rows;
    int r = t0[t];
    tsub0[t] = (int)srow.size();
    int count = 0;
    while (r < t1[t]) {
      int re = r + 1;  // at least one row (a row longer than kMini stands alone)
      while (re < t1[t] && re - r < kMiniRows && pp[re + 1] - pp[r] <= (uint32_t)kMini) re++;
      srow.push_back(r); sk.push_back(pp[r]);
      r = re;
      count++;
    }
    srow.push_back(r); sk.push_back(pp[r]);
    tnsub[t] = count;
  }
  auto up = [&](DevBuf<int> &d, const std::vector<int> &h) { d.alloc(h.size()); d.upload(h.data(), h.size(), s); };
  up(P.tile_sub0, tsub0); up(P.tile_nsub, tnsub); up(P.sub_row, srow);
  P.sub_k.alloc(sk.size()); P.sub_k.upload(sk.data(), sk.size(), s);
  HIP_CHECK(hipStreamSynchronize(s));
  const size_t lds = stream_lds_bytes(P.shift);
  if (lds > 163840) throw Error(6, "panel width too large for the streaming SpMV kernel");
  HIP_CHECK(hipFuncSetAttribute((const void *)k_spmv_stream, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
}

void spmv_panel_stream(const DevCsr &M, const double *x, hipStream_t s) {
  const DevPanel &P = M.panel;
  OQ_LAUNCH(k_spmv_stream, dim3(P.ntiles), dim3(kThreads), stream_lds_bytes(P.shift), s, M.rows, M.cols, P.shift, P.tile_b.get(),
            P.tile_sub0.get(), P.tile_nsub.get(), P.sub_row.get(), P.sub_k.get(), P.pptr.get(), P.pcol.get(), P.pval.get(), x,
            P.partial.get());
}

}  // namespace oq
