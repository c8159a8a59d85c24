// kernels.hip -- hand-written HIP kernels for gfx950 (CDNA4, wave64).
//
// Everything on the ADMM hot path is HBM-bandwidth bound (<= 0.25 flop/byte), so
// the rules that matter are: coalesced 8-byte value / 4-byte index streams, one
// sub-wave group of lanes per CSR row sized to the row length, wavefront
// shuffle reductions, no atomics on fp64 sums (two-stage fixed-order
// reductions keep every solve bit-reproducible), max-norms through integer
// atomicMax on the bit pattern of non-negative doubles (order independent).
#include "kernels.hpp"
#include "devutil.hpp"
#include <algorithm>

namespace oq {

thread_local const int *g_skip = nullptr;
int g_debug_sync = getenv("OSQP_AMD_DEBUG") ? atoi(getenv("OSQP_AMD_DEBUG")) : 0;

// --------------------------------------------------------------------------
// sparse structure
// --------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_expand_colptr(int cols, const int64_t *__restrict__ colptr, int64_t nnz,
                                                          int *__restrict__ out) {
  int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k >= nnz) return;
  int lo = 0, hi = cols;  // find largest j with colptr[j] <= k
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (colptr[mid] <= k) lo = mid; else hi = mid;
  }
  out[k] = lo;
}
void expand_colptr(int cols, const int64_t *colptr, int64_t nnz, int *out, hipStream_t s) {
  if (nnz == 0) return;
  OQ_LAUNCH(k_expand_colptr, dim3(blocks_for(nnz)), dim3(kBlock), 0, s, cols, colptr, nnz, out);
}

constexpr int kScanItems = 4;
constexpr int kScanTile = kBlock * kScanItems;
// per-tile exclusive scan; tile totals to sums[]
__global__ __launch_bounds__(kBlock) void k_scan_tiles(const int64_t *__restrict__ in, int64_t *__restrict__ out,
                                                       int64_t n, int64_t *__restrict__ sums) {
  __shared__ int64_t sm[kBlock];
  int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  int64_t v[kScanItems], tot = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; i++) { v[i] = (base + i < n) ? in[base + i] : 0; tot += v[i]; }
  sm[threadIdx.x] = tot;
  __syncthreads();
  for (int off = 1; off < kBlock; off <<= 1) {
    int64_t t = (threadIdx.x >= off) ? sm[threadIdx.x - off] : 0;
    __syncthreads();
    sm[threadIdx.x] += t;
    __syncthreads();
  }
  int64_t excl = sm[threadIdx.x] - tot;
#pragma unroll
  for (int i = 0; i < kScanItems; i++) { if (base + i < n) out[base + i] = excl; excl += v[i]; }
  if (threadIdx.x == kBlock - 1 && sums) sums[blockIdx.x] = sm[kBlock - 1];
}
__global__ __launch_bounds__(kBlock) void k_scan_add(int64_t *__restrict__ out, int64_t n, const int64_t *__restrict__ offs) {
  int64_t i = (int64_t)blockIdx.x * kScanTile + threadIdx.x;
  int64_t o = offs[blockIdx.x];
  for (int k = 0; k < kScanItems; k++, i += kBlock) if (i < n) out[i] += o;
}
__global__ void k_set_i64(int64_t *p, const int64_t *a, const int64_t *b) { *p = *a + *b; }

static void scan_rec(const int64_t *in, int64_t *out, int64_t n, hipStream_t s) {
  int tiles = blocks_for(n, kScanTile);
  if (tiles == 1) {
    OQ_LAUNCH(k_scan_tiles, dim3(1), dim3(kBlock), 0, s, in, out, n, (int64_t *)nullptr);
    return;
  }
  DevBuf<int64_t> sums(tiles), offs(tiles);
  OQ_LAUNCH(k_scan_tiles, dim3(tiles), dim3(kBlock), 0, s, in, out, n, sums.get());
  scan_rec(sums.get(), offs.get(), tiles, s);
  OQ_LAUNCH(k_scan_add, dim3(tiles), dim3(kBlock), 0, s, out, n, offs.get());
  HIP_CHECK(hipStreamSynchronize(s));  // sums/offs are freed on return
}
void exclusive_scan(const int64_t *counts, int64_t *out, int64_t n, hipStream_t s) {
  // out has n+1 entries; scan the n counts, then out[n] = out[n-1] + counts[n-1]
  if (n == 0) { HIP_CHECK(hipMemsetAsync(out, 0, sizeof(int64_t), s)); return; }
  scan_rec(counts, out, n, s);
  OQ_LAUNCH(k_set_i64, dim3(1), dim3(1), 0, s, out + n, out + n - 1, counts + n - 1);
}

__global__ __launch_bounds__(kBlock) void k_count_rows(int64_t E, const int *__restrict__ erow, int64_t *__restrict__ counts) {
  int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= E) return;
  int r = erow[e];
  if (r >= 0) atomicAdd((unsigned long long *)&counts[r], 1ULL);
}
__global__ __launch_bounds__(kBlock) void k_scatter_coo(int64_t E, const int *__restrict__ erow, const int *__restrict__ ecol,
                                                        int64_t *__restrict__ cursor, int *__restrict__ col, int *__restrict__ src) {
  int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= E) return;
  int r = erow[e];
  if (r < 0) return;
  int64_t pos = (int64_t)atomicAdd((unsigned long long *)&cursor[r], 1ULL);
  col[pos] = ecol[e];
  src[pos] = (int)e;
}
// rows of length <= kSmallRow: one thread per row, insertion sort on (col, src)
constexpr int kSmallRow = 16;
constexpr int kLdsRow = 4096;
__global__ __launch_bounds__(kBlock) void k_sort_rows_small(int rows, const int64_t *__restrict__ rp, int *__restrict__ col,
                                                            int *__restrict__ src) {
  int r = blockIdx.x * kBlock + threadIdx.x;
  if (r >= rows) return;
  int64_t s = rp[r];
  int len = (int)(rp[r + 1] - s);
  if (len > kSmallRow || len < 2) return;
  for (int i = 1; i < len; i++) {
    int c = col[s + i], q = src[s + i];
    int j = i - 1;
    while (j >= 0 && (col[s + j] > c || (col[s + j] == c && src[s + j] > q))) {
      col[s + j + 1] = col[s + j]; src[s + j + 1] = src[s + j]; j--;
    }
    col[s + j + 1] = c; src[s + j + 1] = q;
  }
}
// A row longer than the LDS tile (a dense constraint row such as sum(x) = 1): the normalised bitonic network
// (every compare-exchange ascending; the first step of a merge mirrors inside its block, i ^ (k - 1)) on the
// (col, src) pairs where they lie in global memory.  Partners past the end are skipped -- they stand for +inf
// padding, which an all-ascending network never moves -- so any length works.  One workgroup per row.
__device__ void sort_row_global(int *col, int *src, int64_t len) {
  int64_t N = 1;
  while (N < len) N <<= 1;
  for (int64_t k = 2; k <= N; k <<= 1)
    for (int64_t j = k >> 1; j > 0; j >>= 1) {
      for (int64_t i = threadIdx.x; i < len; i += kBlock) {
        const int64_t p = (j == (k >> 1)) ? (i ^ (k - 1)) : (i ^ j);
        if (p > i && p < len) {
          const int ca = col[i], cb = col[p], sa = src[i], sb = src[p];
          if (ca > cb || (ca == cb && sa > sb)) { col[i] = cb; col[p] = ca; src[i] = sb; src[p] = sa; }
        }
      }
      __syncthreads();
    }
}
// rows of kSmallRow < length <= kLdsRow: one workgroup per row, bitonic sort of 64-bit keys in LDS
__global__ __launch_bounds__(kBlock) void k_sort_rows_lds(int rows, const int64_t *__restrict__ rp, int *__restrict__ col,
                                                          int *__restrict__ src) {
  __shared__ unsigned long long key[kLdsRow];
  int r = blockIdx.x;
  int64_t s = rp[r];
  int64_t len64 = rp[r + 1] - s;
  if (len64 <= kSmallRow) return;
  if (len64 > kLdsRow) { sort_row_global(col + s, src + s, len64); return; }
  int len = (int)len64, N = 1;
  while (N < len) N <<= 1;
  for (int i = threadIdx.x; i < N; i += kBlock)
    key[i] = (i < len) ? (((unsigned long long)(unsigned)col[s + i] << 32) | (unsigned)src[s + i]) : ~0ULL;
  __syncthreads();
  for (int k = 2; k <= N; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < N; i += kBlock) {
        int ixj = i ^ j;
        if (ixj > i) {
          unsigned long long a = key[i], b = key[ixj];
          bool up = ((i & k) == 0);
          if ((a > b) == up) { key[i] = b; key[ixj] = a; }
        }
      }
      __syncthreads();
    }
  for (int i = threadIdx.x; i < len; i += kBlock) { col[s + i] = (int)(key[i] >> 32); src[s + i] = (int)(key[i] & 0xFFFFFFFFu); }
}

// ---------------------------------------------------------------------------------------------------------
// Large inputs: a STABLE least-significant-digit radix sort of (row << 32 | entry index) by row, 6-8 bits per pass, with
// no device-scope atomics (the counting path above spends 49 + 70 + 44 ms per 10^9 entries on row counters, cursors and
// the per-row sort; device-scope integer atomics run at ~20 G/s on this part).  Stable means: inside a row the entries
// keep their source order -- and the two callers that matter hand their entries over column-ascending per row (the
// transpose of a CSR matrix with sorted rows; the lower copies of a sorted upper triangle followed by its mirrored
// entries), so the rows come out sorted and the per-row sort is skipped.  That is checked (k_rows_unsorted); any other
// input gets the per-row sort on top and ends in the same canonical (column, index) order.
//
// One pass = histogram per chunk of 8192 entries (LDS atomics) -> exclusive scan of the [digit][chunk] table -> scatter:
// a wave ranks its quarter of the chunk 64 entries at a time (the peers of a lane = the lanes with the same digit, from
// one ballot per digit bit; running counts per wave in LDS, written by the first peer), the chunk is put in digit order
// in LDS and written out run by run, so that a digit's entries of a chunk leave as one contiguous piece.
// ---------------------------------------------------------------------------------------------------------
constexpr int kSortChunk = 8192;
constexpr int kSortWaveSpan = kSortChunk / 4;   // 4 waves per workgroup
constexpr int kSortSteps = kSortWaveSpan / 64;  // 32

__device__ __forceinline__ unsigned long long sort_word(const int *__restrict__ erow, int64_t e, int rows) {
  const int r = erow[e];
  return ((unsigned long long)(unsigned)(r < 0 ? rows : r) << 32) | (unsigned long long)(unsigned)e;  // skipped entries sort behind the last row
}

template <bool kFirst>
__global__ __launch_bounds__(256) void k_radix_hist(int64_t E, const int *__restrict__ erow, const unsigned long long *__restrict__ in, int rows,
                                                    int shift, unsigned mask, int nb, int64_t nchunks, int64_t *__restrict__ H) {
  __shared__ int h[256];
  const int64_t c = blockIdx.x;
  h[threadIdx.x] = 0;
  __syncthreads();
  const int64_t e0 = c * kSortChunk, e1 = e0 + kSortChunk < E ? e0 + kSortChunk : E;
  for (int64_t e = e0 + threadIdx.x; e < e1; e += 256) {
    const unsigned key = kFirst ? (unsigned)(sort_word(erow, e, rows) >> 32) : (unsigned)(in[e] >> 32);
    atomicAdd(&h[(key >> shift) & mask], 1);
  }
  __syncthreads();
  if ((int)threadIdx.x < nb) H[(int64_t)threadIdx.x * nchunks + c] = h[threadIdx.x];
}

struct SortLds {
  unsigned long long key[kSortChunk];
  int cw[4][256];
  int tot[2][256];
  unsigned goff[256];
  unsigned char dig[kSortChunk];
};

template <bool kFirst>
__global__ __launch_bounds__(256) void k_radix_scatter(int64_t E, const int *__restrict__ erow, const unsigned long long *__restrict__ in, int rows,
                                                       int shift, unsigned mask, int bits, int nb, int64_t nchunks,
                                                       const int64_t *__restrict__ offs, unsigned long long *__restrict__ out) {
  extern __shared__ __align__(16) unsigned char sort_raw[];
  SortLds &L = *reinterpret_cast<SortLds *>(sort_raw);
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const int64_t c = blockIdx.x;
  const int64_t e0 = c * kSortChunk, e1 = e0 + kSortChunk < E ? e0 + kSortChunk : E;
  for (int i = tid; i < 4 * 256; i += 256) (&L.cw[0][0])[i] = 0;
  __syncthreads();
  // (a) every wave ranks its quarter, in source order, without a workgroup barrier
  unsigned long long key[kSortSteps];
  unsigned short rk[kSortSteps];
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
  for (int st = 0; st < kSortSteps; st++) {
    const int64_t e = e0 + (int64_t)w * kSortWaveSpan + st * 64 + lane;
    const bool valid = e < e1;
    unsigned long long k = 0ull;
    if (valid) k = kFirst ? sort_word(erow, e, rows) : in[e];
    const unsigned d = ((unsigned)(k >> 32) >> shift) & mask;
    unsigned long long peers = __ballot(valid);
    for (int b = 0; b < bits; b++) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long bal = __ballot(bit);
      peers &= bit ? bal : ~bal;
    }
    const int before = __popcll(peers & below), cnt = __popcll(peers);
    int old = 0;
    if (valid) old = L.cw[w][d];
    __builtin_amdgcn_wave_barrier();
    if (valid && before == 0) L.cw[w][d] = old + cnt;
    __builtin_amdgcn_wave_barrier();
    key[st] = k;
    rk[st] = (unsigned short)(old + before);
  }
  __syncthreads();
  // (b) where a digit starts inside the chunk, where each wave's share of it starts, where the chunk's piece goes
  int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  if (tid < nb) { c0 = L.cw[0][tid]; c1 = L.cw[1][tid]; c2 = L.cw[2][tid]; c3 = L.cw[3][tid]; }
  L.tot[0][tid] = c0 + c1 + c2 + c3;
  __syncthreads();
  int cur = 0;
  for (int d = 1; d < 256; d <<= 1) {  // inclusive scan over the 256 slots (slots >= nb hold 0)
    const int v = L.tot[cur][tid] + (tid >= d ? L.tot[cur][tid - d] : 0);
    L.tot[1 - cur][tid] = v;
    cur = 1 - cur;
    __syncthreads();
  }
  if (tid < nb) {
    const int start = L.tot[cur][tid] - (c0 + c1 + c2 + c3);
    L.cw[0][tid] = start; L.cw[1][tid] = start + c0; L.cw[2][tid] = start + c0 + c1; L.cw[3][tid] = start + c0 + c1 + c2;
    L.goff[tid] = (unsigned)((unsigned long long)offs[(int64_t)tid * nchunks + c]) - (unsigned)start;  // modulo 2^32: positions are < 2^32
  }
  __syncthreads();
  // (c) the chunk in digit order in LDS
#pragma unroll
  for (int st = 0; st < kSortSteps; st++) {
    const int64_t e = e0 + (int64_t)w * kSortWaveSpan + st * 64 + lane;
    if (e < e1) {
      const unsigned d = ((unsigned)(key[st] >> 32) >> shift) & mask;
      const int pos = L.cw[w][d] + (int)rk[st];
      L.key[pos] = key[st];
      L.dig[pos] = (unsigned char)d;
    }
  }
  __syncthreads();
  // (d) out, run by run
  const int nvalid = (int)(e1 - e0);
  for (int sidx = tid; sidx < nvalid; sidx += 256) {
    const unsigned d = L.dig[sidx];
    out[(size_t)(unsigned)(L.goff[d] + (unsigned)sidx)] = L.key[sidx];
  }
}

// rowptr from the sorted words: rowptr[r] = the first entry whose row is >= r (r = rows: the sentinel row of the dropped
// entries, i.e. the count of kept ones), one bisection per row.  (Up to round 4 entry i wrote every row of the gap
// (row(i-1), row(i)] itself: the forward lists of a supernodal factor have 1.9e6 EMPTY leading rows on control-1e6 -- its
// level 0 -- and one thread wrote them all, 53 ms of a 110 ms device setup.)
__global__ __launch_bounds__(kBlock) void k_rowptr_from_sorted(int64_t E, const unsigned long long *__restrict__ wsorted, int rows,
                                                               int64_t *__restrict__ rowptr) {
  const int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (r > rows) return;
  int64_t lo = 0, hi = E;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)(wsorted[mid] >> 32) < r) lo = mid + 1; else hi = mid;
  }
  rowptr[r] = lo;
}
__global__ __launch_bounds__(kBlock) void k_sorted_extract(int64_t nnz, const unsigned long long *__restrict__ wsorted,
                                                           const int *__restrict__ ecol, int *__restrict__ col, int *__restrict__ src) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= nnz) return;
  const unsigned e = (unsigned)wsorted[i];
  src[i] = (int)e;
  col[i] = ecol[e];
}
__global__ __launch_bounds__(kBlock) void k_rows_unsorted(int64_t nnz, const unsigned long long *__restrict__ wsorted, const int *__restrict__ col,
                                                          int *__restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < 1 || i >= nnz) return;
  if ((wsorted[i] >> 32) == (wsorted[i - 1] >> 32) && col[i - 1] > col[i]) *flag = 1;
}

static bool csr_from_coo_radix(int rows, int cols, int64_t E, const int *erow, const int *ecol, DevCsr &out, DevBuf<int> &src, hipStream_t s) {
  static const int64_t min_entries = getenv("OSQP_AMD_RADIX_MIN") ? atoll(getenv("OSQP_AMD_RADIX_MIN")) : (int64_t)1 << 22;
  if (min_entries < 0 || E < min_entries || E >= 4294967295LL || rows < 1) return false;
  int total_bits = 0;
  while (((unsigned)rows >> total_bits) != 0u) total_bits++;  // the sentinel `rows` itself has to fit
  const int npass = (total_bits + 7) / 8, bits = (total_bits + npass - 1) / npass, nb = 1 << bits;
  const unsigned mask = (unsigned)nb - 1u;
  const int64_t nchunks = (E + kSortChunk - 1) / kSortChunk;
  if (nchunks > 2147483647LL) return false;
  // per call: the attribute belongs to the current device's copy of the kernel
  HIP_CHECK(hipFuncSetAttribute((const void *)k_radix_scatter<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SortLds)));
  HIP_CHECK(hipFuncSetAttribute((const void *)k_radix_scatter<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SortLds)));
  out.rows = rows; out.cols = cols;
  DevBuf<unsigned long long> ping((size_t)E), pong((size_t)E);
  DevBuf<int64_t> H((size_t)nb * nchunks), offs((size_t)nb * nchunks + 1);
  unsigned long long *cur = nullptr, *nxt = ping.get();
  for (int p = 0; p < npass; p++) {
    const int shift = p * bits;
    if (p == 0) OQ_LAUNCH(k_radix_hist<true>, dim3((unsigned)nchunks), dim3(256), 0, s, E, erow, (const unsigned long long *)nullptr, rows, shift, mask, nb, nchunks, H.get());
    else OQ_LAUNCH(k_radix_hist<false>, dim3((unsigned)nchunks), dim3(256), 0, s, E, (const int *)nullptr, (const unsigned long long *)cur, rows, shift, mask, nb, nchunks, H.get());
    exclusive_scan(H.get(), offs.get(), (int64_t)nb * nchunks, s);
    if (p == 0) OQ_LAUNCH(k_radix_scatter<true>, dim3((unsigned)nchunks), dim3(256), sizeof(SortLds), s, E, erow, (const unsigned long long *)nullptr, rows, shift, mask, bits, nb, nchunks, (const int64_t *)offs.get(), nxt);
    else OQ_LAUNCH(k_radix_scatter<false>, dim3((unsigned)nchunks), dim3(256), sizeof(SortLds), s, E, (const int *)nullptr, (const unsigned long long *)cur, rows, shift, mask, bits, nb, nchunks, (const int64_t *)offs.get(), nxt);
    cur = nxt;
    nxt = cur == ping.get() ? pong.get() : ping.get();
  }
  out.rowptr.alloc((size_t)rows + 1);
  OQ_LAUNCH(k_rowptr_from_sorted, dim3(blocks_for((int64_t)rows + 1)), dim3(kBlock), 0, s, E, (const unsigned long long *)cur, rows, out.rowptr.get());
  int64_t nnz = 0;
  HIP_CHECK(hipMemcpyAsync(&nnz, out.rowptr.get() + rows, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  H.release(); offs.release();
  (cur == ping.get() ? pong : ping).release();
  out.nnz = nnz;
  out.col.alloc((size_t)nnz);
  src.alloc((size_t)nnz);
  DevBuf<int> unsorted(1);
  unsorted.zero(s);
  int bad = 0;
  if (nnz > 0) {
    OQ_LAUNCH(k_sorted_extract, dim3(blocks_for(nnz)), dim3(kBlock), 0, s, nnz, (const unsigned long long *)cur, ecol, out.col.get(), src.get());
    OQ_LAUNCH(k_rows_unsorted, dim3(blocks_for(nnz)), dim3(kBlock), 0, s, nnz, (const unsigned long long *)cur, (const int *)out.col.get(), unsorted.get());
    unsorted.download(&bad, 1, s);
    HIP_CHECK(hipStreamSynchronize(s));
  }
  ping.release(); pong.release();
  out.val.alloc((size_t)nnz);
  if (bad) {  // some row did not arrive column-ascending: the per-row sort of the counting path, same canonical order
    OQ_LAUNCH(k_sort_rows_small, dim3(blocks_for(rows)), dim3(kBlock), 0, s, rows, out.rowptr.get(), out.col.get(), src.get());
    OQ_LAUNCH(k_sort_rows_lds, dim3(rows), dim3(kBlock), 0, s, rows, out.rowptr.get(), out.col.get(), src.get());
  }
  out.group = pick_group(rows, nnz);
  return true;
}

void csr_from_coo(int rows, int cols, int64_t E, const int *erow, const int *ecol, DevCsr &out, DevBuf<int> &src,
                  hipStream_t s) {
  if (E >= 2147483647LL) throw Error(6, "more than 2^31-1 entries in one matrix are not supported");
  if (csr_from_coo_radix(rows, cols, E, erow, ecol, out, src, s)) return;
  out.rows = rows; out.cols = cols;
  DevBuf<int64_t> counts((size_t)rows + 1);
  counts.zero(s);
  out.rowptr.alloc((size_t)rows + 1);
  if (E > 0) OQ_LAUNCH(k_count_rows, dim3(blocks_for(E)), dim3(kBlock), 0, s, E, erow, counts.get());
  exclusive_scan(counts.get(), out.rowptr.get(), rows, s);
  int64_t nnz = 0;
  HIP_CHECK(hipMemcpyAsync(&nnz, out.rowptr.get() + rows, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  out.nnz = nnz;
  out.col.alloc((size_t)nnz);
  out.val.alloc((size_t)nnz);
  src.alloc((size_t)nnz);
  if (nnz > 0) {
    HIP_CHECK(hipMemcpyAsync(counts.get(), out.rowptr.get(), sizeof(int64_t) * (size_t)rows, hipMemcpyDeviceToDevice, s));
    OQ_LAUNCH(k_scatter_coo, dim3(blocks_for(E)), dim3(kBlock), 0, s, E, erow, ecol, counts.get(), out.col.get(), src.get());
    OQ_LAUNCH(k_sort_rows_small, dim3(blocks_for(rows)), dim3(kBlock), 0, s, rows, out.rowptr.get(), out.col.get(), src.get());
    OQ_LAUNCH(k_sort_rows_lds, dim3(rows), dim3(kBlock), 0, s, rows, out.rowptr.get(), out.col.get(), src.get());
  }
  out.group = pick_group(rows, nnz);
}

__global__ __launch_bounds__(kBlock) void k_gather_values(int64_t nnz, const int *__restrict__ src, const double *__restrict__ in,
                                                          double *__restrict__ out, int64_t modulo) {
  int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k >= nnz) return;
  int64_t e = src[k];
  if (modulo > 0 && e >= modulo) e -= modulo;
  out[k] = in[e];
}
void gather_values(int64_t nnz, const int *src, const double *in, double *out, int64_t modulo, hipStream_t s) {
  if (nnz == 0) return;
  OQ_LAUNCH(k_gather_values, dim3(blocks_for(nnz)), dim3(kBlock), 0, s, nnz, src, in, out, modulo);
}
__global__ __launch_bounds__(kBlock) void k_invert_map(int64_t nnz, const int *__restrict__ src, int64_t lo, int64_t hi,
                                                       int *__restrict__ k2pos) {
  int64_t pos = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (pos >= nnz) return;
  int64_t e = src[pos];
  if (e >= lo && e < hi) k2pos[e - lo] = (int)pos;
}
void invert_map(int64_t nnz, const int *src, int64_t lo, int64_t hi, int *k2pos, hipStream_t s) {
  if (nnz == 0) return;
  OQ_LAUNCH(k_invert_map, dim3(blocks_for(nnz)), dim3(kBlock), 0, s, nnz, src, lo, hi, k2pos);
}
__global__ __launch_bounds__(kBlock) void k_convert_i64_i32(int64_t n, const int64_t *__restrict__ in, int *__restrict__ out) {
  int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (k < n) out[k] = (int)in[k];
}
void convert_i64_i32(int64_t n, const int64_t *in, int *out, hipStream_t s) {
  if (n == 0) return;
  OQ_LAUNCH(k_convert_i64_i32, dim3(blocks_for(n)), dim3(kBlock), 0, s, n, in, out);
}

int pick_group(int rows, int64_t nnz) {
  if (const char *e = getenv("OSQP_AMD_SPMV_G")) return atoi(e);
  double mean = rows > 0 ? (double)nnz / (double)rows : 0.0;
  if (mean <= 1.5) return 1;
  if (mean <= 3.0) return 2;
  if (mean <= 6.0) return 4;
  if (mean <= 12.0) return 8;
  if (mean <= 24.0) return 16;
  if (mean <= 48.0) return 32;
  return 64;
}

// --------------------------------------------------------------------------
// K6 / K7: CSR SpMV.  G lanes cooperate on one row (G | 64): coalesced streams of
// 8-byte values and 4-byte column indices, 4 independent accumulators per lane to
// keep >= 4 gathers of x in flight, xor-shuffle reduction inside the group.
// --------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(kBlock) void k_spmv(int rows, const int64_t *__restrict__ rp, const int *__restrict__ ci,
                                                 const double *__restrict__ va, const double *__restrict__ x,
                                                 double *__restrict__ y, const double *__restrict__ rscale, double beta,
                                                 double gamma, const double *__restrict__ v, const int *__restrict__ skip) {
  if (skip && *skip) return;
  const int lane = threadIdx.x & (G - 1);
  const int64_t row = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / G;
  if (row >= rows) return;
  const int64_t s = rp[row], e = rp[row + 1];
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int64_t k = s + lane;
  for (; k + 3 * G < e; k += 4 * G) {
    const int c0 = ci[k], c1 = ci[k + G], c2 = ci[k + 2 * G], c3 = ci[k + 3 * G];
    const double v0 = va[k], v1 = va[k + G], v2 = va[k + 2 * G], v3 = va[k + 3 * G];
    a0 += v0 * x[c0]; a1 += v1 * x[c1]; a2 += v2 * x[c2]; a3 += v3 * x[c3];
  }
  for (; k < e; k += G) a0 += va[k] * x[ci[k]];
  double acc = (a0 + a1) + (a2 + a3);
#pragma unroll
  for (int o = G >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) {
    if (rscale) acc *= rscale[row];
    if (beta != 0.0) acc += beta * y[row];
    if (v) acc += gamma * v[row];
    y[row] = acc;
  }
}

void spmv(const DevCsr &M, const double *x, double *y, const double *rscale, double beta, double gamma, const double *v,
          hipStream_t s, const SpmvExtra *extra) {
  if (M.rows == 0) return;
  if (M.panel.active) { spmv_panel(M, x, y, rscale, beta, gamma, v, s, extra); return; }
  if (extra) throw Error(6, "internal: epilogue extras need the panel kernels");
  const int G = M.group;
  dim3 grid(blocks_for((int64_t)M.rows * G)), block(kBlock);
#define OQ_SPMV(GG) \
  OQ_LAUNCH(k_spmv<GG>, grid, block, 0, s, M.rows, M.rowptr.get(), M.col.get(), M.val.get(), x, y, rscale, beta, gamma, v, g_skip)
  switch (G) {
  case 1: OQ_SPMV(1); break;
  case 2: OQ_SPMV(2); break;
  case 4: OQ_SPMV(4); break;
  case 8: OQ_SPMV(8); break;
  case 16: OQ_SPMV(16); break;
  case 32: OQ_SPMV(32); break;
  default: OQ_SPMV(64); break;
  }
#undef OQ_SPMV
}

// --------------------------------------------------------------------------
// K0: Ruiz equilibration pieces
// --------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(kBlock) void k_row_absmax(int rows, const int64_t *__restrict__ rp, const double *__restrict__ va,
                                                       double *__restrict__ out, int accumulate) {
  const int lane = threadIdx.x & (G - 1);
  const int64_t row = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / G;
  if (row >= rows) return;
  const int64_t s = rp[row], e = rp[row + 1];
  double m = 0.0;
  for (int64_t k = s + lane; k < e; k += G) m = fmax(m, fabs(va[k]));
#pragma unroll
  for (int o = G >> 1; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
  if (lane == 0) out[row] = accumulate ? fmax(out[row], m) : m;
}
void csr_row_absmax(const DevCsr &M, double *out, bool accumulate, hipStream_t s) {
  if (M.rows == 0) return;
  if (M.compact) { panel_row_absmax(M, out, accumulate, s); return; }
  const int G = M.group >= 16 ? 64 : (M.group >= 4 ? 8 : 1);
  dim3 grid(blocks_for((int64_t)M.rows * G)), block(kBlock);
  if (G == 64) OQ_LAUNCH(k_row_absmax<64>, grid, block, 0, s, M.rows, M.rowptr.get(), M.val.get(), out, (int)accumulate);
  else if (G == 8) OQ_LAUNCH(k_row_absmax<8>, grid, block, 0, s, M.rows, M.rowptr.get(), M.val.get(), out, (int)accumulate);
  else OQ_LAUNCH(k_row_absmax<1>, grid, block, 0, s, M.rows, M.rowptr.get(), M.val.get(), out, (int)accumulate);
}

// val[k] = ((val[k] * a) * b) * scalar.  order 0: a = r[row], b = c[col];  order 2: a = c[col], b = r[row]
// (so that A and its transposed copy A' round identically);  order 1: a = r[min(row,col)], b = r[max(row,col)]
// (the order in which the CPU statement multiplies an upper-triangular entry, so P stays bit-symmetric;
// both read from c, which is indexed by global row and column ids)
template <int G>
__global__ __launch_bounds__(kBlock) void k_scale_rows_cols(int rows, const int64_t *__restrict__ rp, const int *__restrict__ ci,
                                                            double *__restrict__ va, const double *__restrict__ r,
                                                            const double *__restrict__ c, int symmetric_order, double scalar,
                                                            int row0, double pre, double *__restrict__ norm) {
  const int lane = threadIdx.x & (G - 1);
  const int64_t row = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / G;
  if (row >= rows) return;
  const int64_t s = rp[row], e = rp[row + 1];
  double mx = 0.0;
  for (int64_t k = s + lane; k < e; k += G) {
    double x = va[k];
    if (pre != 1.0) x *= pre;
    if (r) {
      const int col = ci[k];
      double a, b;
      if (symmetric_order == 1) {  // c is indexed by global row / column (row0 = first row of a row block)
        const int gi = (int)row + row0;
        int lo = col < gi ? col : gi, hi = col < gi ? gi : col;
        a = c[lo]; b = c[hi];
      }
      else if (symmetric_order == 2) { a = c[col]; b = r[row]; }
      else { a = r[row]; b = c[col]; }
      x = (x * a) * b;
    }
    if (scalar != 1.0) x *= scalar;
    va[k] = x;
    mx = fmax(mx, fabs(x));
  }
  if (norm) {  // max |row| of the result: what the next Ruiz iteration starts from
#pragma unroll
    for (int o = G >> 1; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) norm[row] = mx;
  }
}
// val <- ((((val * pre) * a) * b) * scalar); norm (may be null): norm[i] = max |row i| of the result
void csr_scale_rows_cols(DevCsr &M, const double *r, const double *c, int symmetric_order, double scalar, hipStream_t s,
                         int row0, double pre, double *norm) {
  if (M.rows == 0) return;
  if (M.nnz == 0) { if (norm) HIP_CHECK(hipMemsetAsync(norm, 0, sizeof(double) * (size_t)M.rows, s)); return; }
  if (M.compact) {
    if ((pre != 1.0 || norm) && scalar == 1.0 && r && norm) { panel_scale_norm(M, r, c, symmetric_order, pre, row0, norm, s); return; }
    if (pre != 1.0) panel_scale(M, nullptr, nullptr, 0, pre, s, row0);
    panel_scale(M, r, c, symmetric_order, scalar, s, row0);
    if (norm) panel_row_absmax(M, norm, false, s);
    return;
  }
  const int G = M.group >= 16 ? 64 : (M.group >= 4 ? 8 : 1);
  dim3 grid(blocks_for((int64_t)M.rows * G)), block(kBlock);
  if (G == 64) OQ_LAUNCH(k_scale_rows_cols<64>, grid, block, 0, s, M.rows, M.rowptr.get(), M.col.get(), M.val.get(), r, c, symmetric_order, scalar, row0, pre, norm);
  else if (G == 8) OQ_LAUNCH(k_scale_rows_cols<8>, grid, block, 0, s, M.rows, M.rowptr.get(), M.col.get(), M.val.get(), r, c, symmetric_order, scalar, row0, pre, norm);
  else OQ_LAUNCH(k_scale_rows_cols<1>, grid, block, 0, s, M.rows, M.rowptr.get(), M.col.get(), M.val.get(), r, c, symmetric_order, scalar, row0, pre, norm);
}

#define MIN_SCALING 1e-4
#define MAX_SCALING 1e4
__global__ __launch_bounds__(kBlock) void k_vec_op(int op, double *__restrict__ out, const double *__restrict__ a,
                                                   const double *__restrict__ b, double sc, double sc2, int n, const int *__restrict__ skip) {
  if (skip && *skip) return;
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  switch (op) {
  case 0: { double d = out[i]; d = d < MIN_SCALING ? 1.0 : d; d = d > MAX_SCALING ? MAX_SCALING : d; out[i] = 1.0 / sqrt(d); break; }
  case 1: { double d = out[i]; d = d < MIN_SCALING ? 1.0 : d; d = d > MAX_SCALING ? MAX_SCALING : d; out[i] = d; break; }
  case 2: out[i] = a[i] * b[i]; break;
  case 3: out[i] = 1.0 / a[i]; break;
  case 4: out[i] *= sc; break;
  case 5: out[i] = sc; break;
  case 6: out[i] = a[i]; break;
  case 7: out[i] = (out[i] * a[i]) * sc; break;
  case 8: out[i] += sc * a[i]; break;
  case 9: out[i] = fmin(fmax(out[i], sc), sc2); break;
  case 10: { double d = sc * a[i]; if (b) d = fmax(d, b[i]); d = d < MIN_SCALING ? 1.0 : d; d = d > MAX_SCALING ? MAX_SCALING : d; out[i] = 1.0 / sqrt(d); break; }
  }
}
static void vec_op(int op, double *out, const double *a, const double *b, double sc, double sc2, int n, hipStream_t s) {
  if (n <= 0) return;
  OQ_LAUNCH(k_vec_op, dim3(blocks_for(n)), dim3(kBlock), 0, s, op, out, a, b, sc, sc2, n, g_skip);
}
void vec_limit_rsqrt(double *d, int n, hipStream_t s) { vec_op(0, d, nullptr, nullptr, 0, 0, n, s); }
void vec_ruiz_factor(double *out, double sc, const double *a, const double *b, int n, hipStream_t s) { vec_op(10, out, a, b, sc, 0, n, s); }
void vec_limit(double *d, int n, hipStream_t s) { vec_op(1, d, nullptr, nullptr, 0, 0, n, s); }
void vec_ew_prod(double *out, const double *a, const double *b, int n, hipStream_t s) { vec_op(2, out, a, b, 0, 0, n, s); }
void vec_ew_recip(double *out, const double *a, int n, hipStream_t s) { vec_op(3, out, a, nullptr, 0, 0, n, s); }
void vec_scale(double *x, double a, int n, hipStream_t s) { vec_op(4, x, nullptr, nullptr, a, 0, n, s); }
void vec_set(double *x, double a, int n, hipStream_t s) { vec_op(5, x, nullptr, nullptr, a, 0, n, s); }
void vec_copy(double *dst, const double *src, int n, hipStream_t s) { vec_op(6, dst, src, nullptr, 0, 0, n, s); }
void vec_scale_by_vec_scalar(double *x, const double *d, double a, int n, hipStream_t s) { vec_op(7, x, d, nullptr, a, 0, n, s); }
void vec_axpy(double *y, double a, const double *x, int n, hipStream_t s) { vec_op(8, y, x, nullptr, a, 0, n, s); }
void vec_clamp(double *x, double lo, double hi, int n, hipStream_t s) { vec_op(9, x, nullptr, nullptr, lo, hi, n, s); }

// --------------------------------------------------------------------------
// reductions
// --------------------------------------------------------------------------
void zero_slots(double *slots, hipStream_t s) { HIP_CHECK(hipMemsetAsync(slots, 0, sizeof(double) * S_COUNT, s)); }

__global__ __launch_bounds__(kBlock) void k_absmax(const double *__restrict__ x, const double *__restrict__ scale, int n,
                                                   double *__restrict__ slot, const int *__restrict__ skip) {
  if (skip && *skip) return;
  double m = 0.0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
    m = nanmax(m, fabs(scale ? scale[i] * x[i] : x[i]));
  m = block_max(m);
  if (threadIdx.x == 0) atomic_max_nonneg(slot, m);
}
void reduce_absmax(const double *x, const double *scale, int n, double *slot, hipStream_t s) {
  if (n <= 0) return;
  int grid = blocks_for(n); if (grid > kReduceBlocks) grid = kReduceBlocks;
  OQ_LAUNCH(k_absmax, dim3(grid), dim3(kBlock), 0, s, x, scale, n, slot, g_skip);
}
__global__ __launch_bounds__(kBlock) void k_dot_partial(const double *__restrict__ a, const double *__restrict__ b, int n,
                                                        double *__restrict__ partials, const int *__restrict__ skip) {
  if (skip && *skip) return;
  double v = 0.0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) v += b ? a[i] * b[i] : a[i];
  v = block_sum(v);
  if (threadIdx.x == 0) partials[blockIdx.x] = v;
}
__global__ __launch_bounds__(kBlock) void k_sum_partials(const double *__restrict__ partials, double *__restrict__ slot, const int *__restrict__ skip) {
  if (skip && *skip) return;
  double v = sum_partials(partials);
  if (threadIdx.x == 0) *slot = v;
}
void reduce_dot(const double *a, const double *b, int n, double *partials, double *slot, hipStream_t s) {
  OQ_LAUNCH(k_dot_partial, dim3(kReduceBlocks), dim3(kBlock), 0, s, a, b, n, partials, g_skip);
  OQ_LAUNCH(k_sum_partials, dim3(1), dim3(kBlock), 0, s, partials, slot, g_skip);
}
void reduce_sum(const double *x, int n, double *partials, double *slot, hipStream_t s) { reduce_dot(x, nullptr, n, partials, slot, s); }

// --------------------------------------------------------------------------
// K1: constraint classification + rho vector
// --------------------------------------------------------------------------
#define RHO_MIN 1e-6
#define RHO_EQ_OVER_RHO_INEQ 1e3
#define RHO_TOL 1e-4
#define INF_SCALED (OSQP_INFTY * MIN_SCALING)
__global__ __launch_bounds__(kBlock) void k_rho_vec(int m, const double *__restrict__ l, const double *__restrict__ u,
                                                    int *__restrict__ ctype, double *__restrict__ rho, double *__restrict__ rho_inv,
                                                    double rho_s, int mode, int *__restrict__ flag) {
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= m) return;
  int t = ctype[i];
  if (mode != 2) {
    int tn;
    if (l[i] < -INF_SCALED && u[i] > INF_SCALED) tn = -1;
    else if (u[i] - l[i] < RHO_TOL) tn = 1;
    else tn = 0;
    if (mode == 1) { if (tn == t) return; *flag = 1; }
    t = tn;
    ctype[i] = t;
  } else if (t == -1) return;
  double r = t == -1 ? RHO_MIN : (t == 1 ? RHO_EQ_OVER_RHO_INEQ * rho_s : rho_s);
  rho[i] = r;
  rho_inv[i] = 1.0 / r;
}
void rho_vec_update(int m, const double *l, const double *u, int *ctype, double *rho, double *rho_inv, double rho_scalar,
                    int mode, int *flag, hipStream_t s) {
  if (m <= 0) return;
  OQ_LAUNCH(k_rho_vec, dim3(blocks_for(m)), dim3(kBlock), 0, s, m, l, u, ctype, rho, rho_inv, rho_scalar, mode, flag);
}

// --------------------------------------------------------------------------
// K5: ADMM vector updates
// --------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_admm_rhs(int n, int m, double sigma, const double *__restrict__ x_prev,
                                                     const double *__restrict__ q, const double *__restrict__ z_prev,
                                                     const double *__restrict__ rho_inv, const double *__restrict__ y,
                                                     double *__restrict__ xz, const int *__restrict__ skip) {
  if (skip && *skip) return;
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) xz[i] = sigma * x_prev[i] - q[i];
  else if (i < n + m) { int j = i - n; xz[i] = z_prev[j] - rho_inv[j] * y[j]; }
}
void admm_rhs(int n, int m, double sigma, const double *x_prev, const double *q, const double *z_prev, const double *rho_inv,
              const double *y, double *xz, hipStream_t s) {
  OQ_LAUNCH(k_admm_rhs, dim3(blocks_for((int64_t)n + m)), dim3(kBlock), 0, s, n, m, sigma, x_prev, q, z_prev, rho_inv, y, xz, g_skip);
}
// in place: x and z hold the previous iterate on entry and the new one on exit (no x_prev / z_prev copies,
// no pointer swap -- which also keeps every launch argument constant, so the iteration can be graph-captured)
__global__ __launch_bounds__(kBlock) void k_admm_update(int n, int m, double alpha, const double *__restrict__ xt, const double *__restrict__ ztv,
                                                        const double *__restrict__ rho, const double *__restrict__ rho_inv,
                                                        const double *__restrict__ l, const double *__restrict__ u,
                                                        double *__restrict__ x, double *__restrict__ z, double *__restrict__ y,
                                                        double *__restrict__ delta_x, double *__restrict__ delta_y, const int *__restrict__ skip) {
  if (skip && *skip) return;
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) {
    double xp = x[i];
    double xn = alpha * xt[i] + (1.0 - alpha) * xp;
    x[i] = xn;
    delta_x[i] = xn - xp;
  } else if (i < n + m) {
    int j = i - n;
    double zt = ztv[j], zp = z[j], yj = y[j];
    double zh = alpha * zt + (1.0 - alpha) * zp;
    double zn = zh + rho_inv[j] * yj;
    zn = fmin(fmax(zn, l[j]), u[j]);
    z[j] = zn;
    double dy = rho[j] * (zh - zn);
    delta_y[j] = dy;
    y[j] = yj + dy;
  }
}
void admm_update(int n, int m, double alpha, const double *xz, const double *rho, const double *rho_inv, const double *l,
                 const double *u, double *x, double *z, double *y, double *delta_x, double *delta_y, hipStream_t s) {
  OQ_LAUNCH(k_admm_update, dim3(blocks_for((int64_t)n + m)), dim3(kBlock), 0, s, n, m, alpha, xz, xz + n, rho, rho_inv, l, u, x, z, y,
            delta_x, delta_y, g_skip);
}
// the update, and in the same pass what the NEXT iteration's right-hand side starts from (pcg.hip, k_pcg_rhs: the same
// expressions on the values just written): xz_x = sigma x - q, t = rho (z - rho^-1 y), the six scratch slots cleared
__global__ __launch_bounds__(kBlock) void k_admm_update_rhs(int n, int m, double alpha, const double *__restrict__ xt, const double *__restrict__ ztv,
                                                            const double *__restrict__ rho, const double *__restrict__ rho_inv,
                                                            const double *__restrict__ l, const double *__restrict__ u,
                                                            double *__restrict__ x, double *__restrict__ z, double *__restrict__ y,
                                                            double *__restrict__ delta_x, double *__restrict__ delta_y, double sigma,
                                                            const double *__restrict__ q, double *__restrict__ xz_x, double *__restrict__ t,
                                                            double *__restrict__ slots, const int *__restrict__ skip) {
  if (skip && *skip) return;
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < 6) slots[S_T0 + i] = 0.0;
  if (i < n) {
    double xp = x[i];
    double xn = alpha * xt[i] + (1.0 - alpha) * xp;
    x[i] = xn;
    delta_x[i] = xn - xp;
    xz_x[i] = sigma * xn - q[i];
  } else if (i < n + m) {
    int j = i - n;
    double zt = ztv[j], zp = z[j], yj = y[j];
    double zh = alpha * zt + (1.0 - alpha) * zp;
    double zn = zh + rho_inv[j] * yj;
    zn = fmin(fmax(zn, l[j]), u[j]);
    z[j] = zn;
    double dy = rho[j] * (zh - zn);
    delta_y[j] = dy;
    const double yn = yj + dy;
    y[j] = yn;
    const double rz = zn - rho_inv[j] * yn;
    t[j] = rho[j] * rz;
  }
}
void admm_update2_rhs(int n, int m, double alpha, const double *xt, const double *zt, const double *rho, const double *rho_inv, const double *l,
                      const double *u, double *x, double *z, double *y, double *delta_x, double *delta_y, double sigma, const double *q,
                      double *xz_x, double *t, double *slots, hipStream_t s) {
  OQ_LAUNCH(k_admm_update_rhs, dim3(blocks_for((int64_t)n + m)), dim3(kBlock), 0, s, n, m, alpha, xt, zt, rho, rho_inv, l, u, x, z, y,
            delta_x, delta_y, sigma, q, xz_x, t, slots, g_skip);
}
void admm_update2(int n, int m, double alpha, const double *xt, const double *zt, const double *rho, const double *rho_inv, const double *l,
                  const double *u, double *x, double *z, double *y, double *delta_x, double *delta_y, hipStream_t s) {
  OQ_LAUNCH(k_admm_update, dim3(blocks_for((int64_t)n + m)), dim3(kBlock), 0, s, n, m, alpha, xt, zt, rho, rho_inv, l, u, x, z, y,
            delta_x, delta_y, g_skip);
}

// --------------------------------------------------------------------------
// K8: residual norms + objective pieces, one pass over the n- and m-vectors
// --------------------------------------------------------------------------
// One pass over the n- and m-vectors, 14 maxima + 2 sums per thread; per block ONE barrier: wavefront reductions, the four
// wave results meet in LDS, 16 threads write the block's partials [16 x kReduceBlocks].  A single-block finish kernel
// folds them (maxima are order independent, the sums keep the fixed two-stage order) -- no atomics: 14 contended
// atomicMax per block cost more than the whole pass (110 -> ~20 us at n = 5e5, m = 1e6).
__global__ __launch_bounds__(kBlock) void k_residual_norms(int n, int m, const double *__restrict__ x, const double *__restrict__ z,
                                                           const double *__restrict__ Ax, const double *__restrict__ Px,
                                                           const double *__restrict__ Aty, const double *__restrict__ q,
                                                           const double *__restrict__ Dinv, const double *__restrict__ Einv,
                                                           double *__restrict__ partials) {
  __shared__ double sm[4][16];
  double v[16];
#pragma unroll
  for (int k = 0; k < 16; k++) v[k] = 0.0;
  const int stride = gridDim.x * kBlock;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < m; i += stride) {
    double ax = Ax[i], zi = z[i], e = Einv[i], r = ax - zi;
    v[S_PRI] = nanmax(v[S_PRI], fabs(r));   v[S_PRI_UNS] = nanmax(v[S_PRI_UNS], fabs(e * r));
    v[S_Z] = nanmax(v[S_Z], fabs(zi));      v[S_AX] = nanmax(v[S_AX], fabs(ax));
    v[S_Z_UNS] = nanmax(v[S_Z_UNS], fabs(e * zi)); v[S_AX_UNS] = nanmax(v[S_AX_UNS], fabs(e * ax));
  }
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    double px = Px[i], qi = q[i], at = m > 0 ? Aty[i] : 0.0, d = Dinv[i], xi = x[i];
    double r = (qi + px) + at;
    v[S_DUA] = nanmax(v[S_DUA], fabs(r));   v[S_DUA_UNS] = nanmax(v[S_DUA_UNS], fabs(d * r));
    v[S_Q] = nanmax(v[S_Q], fabs(qi));      v[S_ATY] = nanmax(v[S_ATY], fabs(at));   v[S_PX] = nanmax(v[S_PX], fabs(px));
    v[S_Q_UNS] = nanmax(v[S_Q_UNS], fabs(d * qi)); v[S_ATY_UNS] = nanmax(v[S_ATY_UNS], fabs(d * at));
    v[S_PX_UNS] = nanmax(v[S_PX_UNS], fabs(d * px));
    v[S_XPX] += xi * px; v[S_QX] += qi * xi;
  }
#pragma unroll
  for (int k = 0; k < 14; k++) v[k] = wave_max(v[k]);
  v[S_XPX] = wave_sum(v[S_XPX]);
  v[S_QX] = wave_sum(v[S_QX]);
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 16; k++) sm[threadIdx.x >> 6][k] = v[k];
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    const int k = threadIdx.x;
    const double a = sm[0][k], b = sm[1][k], c = sm[2][k], d = sm[3][k];
    partials[(size_t)k * kReduceBlocks + blockIdx.x] = k < 14 ? nanmax(nanmax(a, b), nanmax(c, d)) : (a + b) + (c + d);
  }
}
// slots[k] = max / sum over the blocks' partials (k < 14: max; 14, 15: the two-stage sum of sum_partials)
__global__ __launch_bounds__(kBlock) void k_residual_finish(const double *__restrict__ partials, double *__restrict__ slots) {
  for (int k = 0; k < 14; k++) {
    double mx = 0.0;
    for (int i = threadIdx.x; i < kReduceBlocks; i += kBlock) mx = nanmax(mx, partials[(size_t)k * kReduceBlocks + i]);
    mx = block_max(mx);
    if (threadIdx.x == 0) slots[k] = mx;
  }
  const double a = sum_partials(partials + (size_t)S_XPX * kReduceBlocks);
  const double b = sum_partials(partials + (size_t)S_QX * kReduceBlocks);
  if (threadIdx.x == 0) { slots[S_XPX] = a; slots[S_QX] = b; }
}
__global__ __launch_bounds__(kBlock) void k_sum_partials2(const double *__restrict__ partials, double *__restrict__ s0, double *__restrict__ s1, const int *__restrict__ skip) {
  if (skip && *skip) return;
  double a = sum_partials(partials);
  double b = sum_partials(partials + kReduceBlocks);
  if (threadIdx.x == 0) { *s0 = a; *s1 = b; }
}
void residual_norms(int n, int m, const double *x, const double *z, const double *Ax, const double *Px, const double *Aty,
                    const double *q, const double *Dinv, const double *Einv, double *slots, double *partials, hipStream_t s) {
  // partials: 16 * kReduceBlocks doubles
  OQ_LAUNCH(k_residual_norms, dim3(kReduceBlocks), dim3(kBlock), 0, s, n, m, x, z, Ax, Px, Aty, q, Dinv, Einv, partials);
  OQ_LAUNCH(k_residual_finish, dim3(1), dim3(kBlock), 0, s, partials, slots);
}

// --------------------------------------------------------------------------
// K10: infeasibility tests
// --------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_prim_infeas_prep(int m, double *__restrict__ dy, const double *__restrict__ l,
                                                             const double *__restrict__ u, const double *__restrict__ E,
                                                             double *__restrict__ slots, double *__restrict__ partials) {
  double mx = 0.0, sum = 0.0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < m; i += gridDim.x * kBlock) {
    double d = dy[i], li = l[i], ui = u[i];
    if (ui > INF_SCALED) { if (li < -INF_SCALED) d = 0.0; else d = fmin(d, 0.0); }
    else if (li < -INF_SCALED) d = fmax(d, 0.0);
    dy[i] = d;
    mx = nanmax(mx, fabs(E ? E[i] * d : d));
    sum += ui * fmax(d, 0.0) + li * fmin(d, 0.0);
  }
  mx = block_max(mx);
  sum = block_sum(sum);
  if (threadIdx.x == 0) { atomic_max_nonneg(&slots[S_T0], mx); partials[blockIdx.x] = sum; }
}
void prim_infeas_prep(int m, double *dy, const double *l, const double *u, const double *E, double *slots, double *partials,
                      hipStream_t s) {
  HIP_CHECK(hipMemsetAsync(slots + S_T0, 0, sizeof(double) * 6, s));
  OQ_LAUNCH(k_prim_infeas_prep, dim3(kReduceBlocks), dim3(kBlock), 0, s, m, dy, l, u, E, slots, partials);
  OQ_LAUNCH(k_sum_partials, dim3(1), dim3(kBlock), 0, s, partials, slots + S_T1, g_skip);
}
__global__ __launch_bounds__(kBlock) void k_dual_infeas_rows(int m, const double *__restrict__ Adx, const double *__restrict__ Einv,
                                                             const double *__restrict__ l, const double *__restrict__ u, double thr,
                                                             double *__restrict__ slots) {
  double bad = 0.0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < m; i += gridDim.x * kBlock) {
    double a = Einv ? Einv[i] * Adx[i] : Adx[i];
    if ((u[i] < INF_SCALED && a > thr) || (l[i] > -INF_SCALED && a < -thr) || a != a) bad = 1.0;
  }
  bad = block_max(bad);
  if (threadIdx.x == 0 && bad > 0.0) atomic_max_nonneg(&slots[S_T2], bad);
}
void dual_infeas_rows(int m, const double *Adx, const double *Einv, const double *l, const double *u, double thr, double *slots,
                      hipStream_t s) {
  HIP_CHECK(hipMemsetAsync(slots + S_T2, 0, sizeof(double), s));
  if (m <= 0) return;
  int grid = blocks_for(m); if (grid > kReduceBlocks) grid = kReduceBlocks;
  OQ_LAUNCH(k_dual_infeas_rows, dim3(grid), dim3(kBlock), 0, s, m, Adx, Einv, l, u, thr, slots);
}

// --------------------------------------------------------------------------
// K9: PCG pieces
// --------------------------------------------------------------------------
// one 8-lane group per row j of A' (column j of A): d = sigma + P_jj + sum rho_i A_ij^2
__global__ __launch_bounds__(kBlock) void k_pcg_precond(int n, const int64_t *__restrict__ atp, const int *__restrict__ ati,
                                                        const double *__restrict__ atx, const int64_t *__restrict__ pp,
                                                        const int *__restrict__ pi, const double *__restrict__ px,
                                                        const double *__restrict__ rho, double sigma, double *__restrict__ dinv,
                                                        int row0) {
  constexpr int G = 8;
  const int lane = threadIdx.x & (G - 1);
  const int64_t j = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / G;
  if (j >= n) return;
  double acc = 0.0;
  if (atp) for (int64_t k = atp[j] + lane; k < atp[j + 1]; k += G) { double a = atx[k]; acc += rho[ati[k]] * a * a; }
  for (int64_t k = pp[j] + lane; k < pp[j + 1]; k += G) if (pi[k] == (int)j + row0) acc += px[k];
#pragma unroll
  for (int o = G >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) dinv[j] = 1.0 / (sigma + acc);
}
// one piece of the Jacobi diagonal from CSR arrays (the other matrix being compact): out = (accumulate ? out : 0) + piece
__global__ __launch_bounds__(kBlock) void k_precond_piece(int n, const int64_t *__restrict__ atp, const int *__restrict__ ati,
                                                          const double *__restrict__ atx, const int64_t *__restrict__ pp,
                                                          const int *__restrict__ pi, const double *__restrict__ px,
                                                          const double *__restrict__ rho, double *__restrict__ out, int row0, int accumulate) {
  constexpr int G = 8;
  const int lane = threadIdx.x & (G - 1);
  const int64_t j = ((int64_t)blockIdx.x * kBlock + threadIdx.x) / G;
  if (j >= n) return;
  double acc = 0.0;
  if (atp) for (int64_t k = atp[j] + lane; k < atp[j + 1]; k += G) { double a = atx[k]; acc += rho[ati[k]] * a * a; }
  if (pp) for (int64_t k = pp[j] + lane; k < pp[j + 1]; k += G) if (pi[k] == (int)j + row0) acc += px[k];
#pragma unroll
  for (int o = G >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) out[j] = accumulate ? out[j] + acc : acc;
}
// compact matrices: dinv = 1 / (sigma + diag(P) + (A' .* A') rho), both pieces from the sliced-ELL copies
__global__ __launch_bounds__(kBlock) void k_precond_finish(int n, double sigma, const double *__restrict__ acc, double *__restrict__ dinv) {
  int j = blockIdx.x * kBlock + threadIdx.x;
  if (j < n) dinv[j] = 1.0 / (sigma + acc[j]);
}
void pcg_precond(const DevCsr &At, const DevCsr &Pf, const double *rho, double sigma, double *dinv, hipStream_t s, int row0) {
  int n = Pf.rows;
  if (At.compact || Pf.compact) {  // each piece from the copy its matrix still has
    const bool hasA = At.rows > 0 && At.cols > 0;
    if (Pf.compact) panel_diag(Pf, dinv, row0, s);                           // dinv <- diag(P)
    else OQ_LAUNCH(k_precond_piece, dim3(blocks_for((int64_t)n * 8)), dim3(kBlock), 0, s, n, (const int64_t *)nullptr, (const int *)nullptr,
                   (const double *)nullptr, Pf.rowptr.get(), Pf.col.get(), Pf.val.get(), rho, dinv, row0, 0);
    if (hasA) {
      if (At.compact) spmv_panel_squared(At, rho, dinv, 1.0, dinv, s);      // dinv <- (A' .* A') rho + diag(P)
      else OQ_LAUNCH(k_precond_piece, dim3(blocks_for((int64_t)n * 8)), dim3(kBlock), 0, s, n, At.rowptr.get(), At.col.get(), At.val.get(),
                     (const int64_t *)nullptr, (const int *)nullptr, (const double *)nullptr, rho, dinv, row0, 1);
    }
    OQ_LAUNCH(k_precond_finish, dim3(blocks_for(n)), dim3(kBlock), 0, s, n, sigma, dinv, dinv);
    return;
  }
  OQ_LAUNCH(k_pcg_precond, dim3(blocks_for((int64_t)n * 8)), dim3(kBlock), 0, s, n,
                     At.cols > 0 && At.rows > 0 ? At.rowptr.get() : (const int64_t *)nullptr, At.col.get(), At.val.get(),
                     Pf.rowptr.get(), Pf.col.get(), Pf.val.get(), rho, sigma, dinv, row0);
}

__global__ __launch_bounds__(kBlock) void k_pcg_init(int n, const double *__restrict__ b, const double *__restrict__ w,
                                                     const double *__restrict__ dinv, double *__restrict__ r, double *__restrict__ zz,
                                                     double *__restrict__ p, double *__restrict__ partials, double *__restrict__ slot_rn, const int *__restrict__ skip) {
  if (skip && *skip) return;
  double rz = 0.0, mx = 0.0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    double ri = b[i] - w[i];
    double zi = dinv[i] * ri;
    r[i] = ri; zz[i] = zi; p[i] = zi;
    rz += ri * zi;
    mx = nanmax(mx, fabs(ri));
  }
  rz = block_sum(rz);
  mx = block_max(mx);
  if (threadIdx.x == 0) { partials[blockIdx.x] = rz; atomic_max_nonneg(slot_rn, mx); }
}
void pcg_init_residual(int n, const double *b, const double *w, const double *dinv, double *r, double *zz, double *p,
                       double *partials, double *slot_rz, double *slot_rn, hipStream_t s) {
  // *slot_rn must be zero on entry (the caller clears its slot block once per solve)
  OQ_LAUNCH(k_pcg_init, dim3(kReduceBlocks), dim3(kBlock), 0, s, n, b, w, dinv, r, zz, p, partials, slot_rn, g_skip);
  OQ_LAUNCH(k_sum_partials, dim3(1), dim3(kBlock), 0, s, partials, slot_rz, g_skip);
}
__global__ __launch_bounds__(kBlock) void k_pcg_update_xr(int n, const double *__restrict__ slot_rz, const double *__restrict__ slot_pw,
                                                          double *__restrict__ x, const double *__restrict__ p, double *__restrict__ r,
                                                          const double *__restrict__ w, const double *__restrict__ dinv,
                                                          double *__restrict__ zz, double *__restrict__ partials,
                                                          double *__restrict__ slot_rn, const int *__restrict__ skip) {
  if (skip && *skip) return;
  const double alpha = *slot_rz / *slot_pw;
  double rz = 0.0, mx = 0.0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    x[i] += alpha * p[i];
    double ri = r[i] - alpha * w[i];
    double zi = dinv[i] * ri;
    r[i] = ri; zz[i] = zi;
    rz += ri * zi;
    mx = nanmax(mx, fabs(ri));
  }
  rz = block_sum(rz);
  mx = block_max(mx);
  if (threadIdx.x == 0) { partials[blockIdx.x] = rz; atomic_max_nonneg(slot_rn, mx); }
}
__global__ void k_fill_slots(double *slots, int count, double value, const int *__restrict__ skip) {
  if (skip && *skip) return;
  if ((int)threadIdx.x < count) slots[threadIdx.x] = value;
}
void fill_slots(double *slots, int count, double value, hipStream_t s) {
  OQ_LAUNCH(k_fill_slots, dim3(1), dim3(64), 0, s, slots, count, value, g_skip);
}
void pcg_update_xr(int n, const double *slot_rz, const double *slot_pw, double *x, const double *p, double *r, const double *w,
                   const double *dinv, double *zz, double *partials, double *slot_rz_new, double *slot_rn, hipStream_t s) {
  fill_slots(slot_rn, 1, 0.0, s);
  OQ_LAUNCH(k_pcg_update_xr, dim3(kReduceBlocks), dim3(kBlock), 0, s, n, slot_rz, slot_pw, x, p, r, w, dinv, zz, partials, slot_rn, g_skip);
  OQ_LAUNCH(k_sum_partials, dim3(1), dim3(kBlock), 0, s, partials, slot_rz_new, g_skip);
}
__global__ __launch_bounds__(kBlock) void k_pcg_update_p(int n, const double *__restrict__ slot_rz_new, const double *__restrict__ slot_rz,
                                                         const double *__restrict__ zz, double *__restrict__ p, const int *__restrict__ skip) {
  if (skip && *skip) return;
  const double beta = *slot_rz_new / *slot_rz;
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) p[i] = zz[i] + beta * p[i];
}
void pcg_update_p(int n, const double *slot_rz_new, const double *slot_rz, const double *zz, double *p, hipStream_t s) {
  OQ_LAUNCH(k_pcg_update_p, dim3(blocks_for(n)), dim3(kBlock), 0, s, n, slot_rz_new, slot_rz, zz, p, g_skip);
}
// Start vector of a CG solve from the last two solutions x1 (newest), x0: the point x1 + theta (x1 - x0) that is
// closest to the new solution in the energy norm, theta = e'r / e'Me with e = x1 - x0, r = b - M x1 (M x1, M x0
// are known).  num = e'r, den = e'(M x1 - M x0) -> slots; k_extrapolate3_dev applies theta = num / den (0 when the
// two solutions coincide, clamped to [-1, 4]) to x, M x and A x and rotates the pairs.
__global__ __launch_bounds__(kBlock) void k_extrap_dots(int n, const double *__restrict__ x1, const double *__restrict__ x0,
                                                        const double *__restrict__ Mx1, const double *__restrict__ Mx0,
                                                        const double *__restrict__ b, double *__restrict__ partials, const int *__restrict__ skip) {
  if (skip && *skip) return;
  double num = 0.0, den = 0.0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const double e = x1[i] - x0[i], m1 = Mx1[i];
    num += e * (b[i] - m1);
    den += e * (m1 - Mx0[i]);
  }
  num = block_sum(num);
  den = block_sum(den);
  if (threadIdx.x == 0) { partials[blockIdx.x] = num; partials[kReduceBlocks + blockIdx.x] = den; }
}
void pcg_extrap_dots(int n, const double *x1, const double *x0, const double *Mx1, const double *Mx0, const double *b,
                     double *partials, double *slot_num, double *slot_den, hipStream_t s) {
  OQ_LAUNCH(k_extrap_dots, dim3(kReduceBlocks), dim3(kBlock), 0, s, n, x1, x0, Mx1, Mx0, b, partials, g_skip);
  OQ_LAUNCH(k_sum_partials2, dim3(1), dim3(kBlock), 0, s, partials, slot_num, slot_den, g_skip);
}
// the three vector pairs of a CG start (x, M x over n; A x over m) in one launch
__global__ __launch_bounds__(kBlock) void k_extrapolate3_dev(double *__restrict__ a1, double *__restrict__ a0, double *__restrict__ b1,
                                                             double *__restrict__ b0, int n, double *__restrict__ c1,
                                                             double *__restrict__ c0, int m, const double *__restrict__ num,
                                                             const double *__restrict__ den, const int *__restrict__ skip) {
  if (skip && *skip) return;
  const double d = *den;
  double theta = d > 0.0 ? *num / d : 0.0;
  theta = theta != theta ? 0.0 : fmin(fmax(theta, -1.0), 4.0);
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) {
    double cur = a1[i], old = a0[i]; a0[i] = cur; a1[i] = cur + theta * (cur - old);
    cur = b1[i]; old = b0[i]; b0[i] = cur; b1[i] = cur + theta * (cur - old);
  } else if (i < n + m) {
    const int j = i - n;
    const double cur = c1[j], old = c0[j]; c0[j] = cur; c1[j] = cur + theta * (cur - old);
  }
}
void pcg_extrapolate3(double *x1, double *x0, double *Mx1, double *Mx0, int n, double *Ax1, double *Ax0, int m,
                      const double *slot_num, const double *slot_den, hipStream_t s) {
  OQ_LAUNCH(k_extrapolate3_dev, dim3(blocks_for((int64_t)n + m)), dim3(kBlock), 0, s, x1, x0, Mx1, Mx0, n, Ax1, Ax0, m, slot_num, slot_den, g_skip);
}
__global__ __launch_bounds__(kBlock) void k_copy2(double *__restrict__ d1, const double *__restrict__ s1, int n1, double *__restrict__ d2,
                                                  const double *__restrict__ s2, int n2, const int *__restrict__ skip) {
  if (skip && *skip) return;
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n1) d1[i] = s1[i];
  else if (i < n1 + n2) d2[i - n1] = s2[i - n1];
}
void vec_copy2(double *d1, const double *s1, int n1, double *d2, const double *s2, int n2, hipStream_t s) {
  if (n1 + n2 <= 0) return;
  OQ_LAUNCH(k_copy2, dim3(blocks_for((int64_t)n1 + n2)), dim3(kBlock), 0, s, d1, s1, n1, d2, s2, n2, g_skip);
}
// y1 += (num/den) x1 over n1 and y2 += (num/den) x2 over n2 in one launch
__global__ __launch_bounds__(kBlock) void k_axpy2_dev(double *__restrict__ y1, const double *__restrict__ x1, int n1, double *__restrict__ y2,
                                                      const double *__restrict__ x2, int n2, const double *__restrict__ num,
                                                      const double *__restrict__ den, const int *__restrict__ skip) {
  if (skip && *skip) return;
  const double a = *num / *den;
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n1) y1[i] += a * x1[i];
  else if (i < n1 + n2) y2[i - n1] += a * x2[i - n1];
}
void vec_axpy2_dev(double *y1, const double *x1, int n1, double *y2, const double *x2, int n2, const double *slot_num,
                   const double *slot_den, hipStream_t s) {
  if (n1 + n2 <= 0) return;
  OQ_LAUNCH(k_axpy2_dev, dim3(blocks_for((int64_t)n1 + n2)), dim3(kBlock), 0, s, y1, x1, n1, y2, x2, n2, slot_num, slot_den, g_skip);
}

// ---------------- row blocks of a CSR matrix and rank-ordered scalar combination (sharded path, row N4) ----------------
__global__ __launch_bounds__(kBlock) void k_rebase_rowptr(int count, const int64_t *__restrict__ in, int64_t base,
                                                          int64_t *__restrict__ out) {
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < count) out[i] = in[i] - base;
}
void csr_slice_rows(DevCsr &M, int r0, int r1, hipStream_t s) {
  if (r0 < 0 || r1 > M.rows || r1 < r0) throw Error(6, "csr_slice_rows: bad range");
  int64_t be[2];
  HIP_CHECK(hipMemcpyAsync(&be[0], M.rowptr.get() + r0, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipMemcpyAsync(&be[1], M.rowptr.get() + r1, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  DevCsr out;
  out.rows = r1 - r0; out.cols = M.cols; out.nnz = be[1] - be[0];
  out.rowptr.alloc((size_t)out.rows + 1);
  OQ_LAUNCH(k_rebase_rowptr, dim3(blocks_for(out.rows + 1)), dim3(kBlock), 0, s, out.rows + 1, M.rowptr.get() + r0, be[0],
            out.rowptr.get());
  out.col.alloc((size_t)out.nnz); out.val.alloc((size_t)out.nnz);
  if (out.nnz > 0) {
    HIP_CHECK(hipMemcpyAsync(out.col.get(), M.col.get() + be[0], sizeof(int) * (size_t)out.nnz, hipMemcpyDeviceToDevice, s));
    HIP_CHECK(hipMemcpyAsync(out.val.get(), M.val.get() + be[0], sizeof(double) * (size_t)out.nnz, hipMemcpyDeviceToDevice, s));
  }
  out.group = pick_group(out.rows, out.nnz);
  HIP_CHECK(hipStreamSynchronize(s));
  M = std::move(out);
}

// One wavefront; every rank runs the same loop over the same gathered copies in rank order, so all ranks end up
// with bit-identical scalars (the host control flow of the ranks must never diverge).
__global__ void k_combine_rank_slots(const double *__restrict__ gathered, int world, int count, unsigned sum_mask,
                                     double *__restrict__ out) {
  const int k = threadIdx.x;
  if (k >= count) return;
  const bool is_sum = (sum_mask >> k) & 1u;
  double acc = gathered[k];
  for (int r = 1; r < world; r++) {
    const double v = gathered[(size_t)r * count + k];
    acc = is_sum ? acc + v : nanmax(acc, v);
  }
  out[k] = acc;
}
void combine_rank_slots(const double *gathered, int world, int count, unsigned sum_mask, double *out, hipStream_t s) {
  if (count > 32) throw Error(6, "combine_rank_slots: at most 32 slots");
  OQ_LAUNCH(k_combine_rank_slots, dim3(1), dim3(64), 0, s, gathered, world, count, sum_mask, out);
}

}  // namespace oq
