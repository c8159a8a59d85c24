// common.hpp -- shared declarations of the HIP engine (libosqp_amd.so).
// MI355X / gfx950 only: 64-lane wavefronts, 256 CUs in 8 XCDs, HBM3E.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/osqp_amd.h"

namespace oq {

// ---- error handling: exceptions inside, return codes at the C boundary ----
struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
void set_last_error(const std::string &m);

#define HIP_CHECK(expr)                                                                             \
  do {                                                                                              \
    hipError_t _e = (expr);                                                                         \
    if (_e != hipSuccess)                                                                           \
      throw oq::Error(6, std::string("HIP error: ") + hipGetErrorString(_e) + " at " + __FILE__ +   \
                             ":" + std::to_string(__LINE__));                                       \
  } while (0)

// Every kernel launch goes through OQ_LAUNCH: launch errors are always checked;
// with OSQP_AMD_DEBUG=1 each launch is announced on stderr and synchronised, so a
// faulting kernel is the last name printed.
extern int g_debug_sync;
inline void post_launch(const char *name, hipStream_t s) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) throw Error(6, std::string("launch of ") + name + " failed: " + hipGetErrorString(e));
  if (g_debug_sync) {
    fprintf(stderr, "[osqp-amd] %s ... ", name);
    fflush(stderr);
    e = hipStreamSynchronize(s);
    fprintf(stderr, "%s\n", e == hipSuccess ? "ok" : hipGetErrorString(e));
    if (e != hipSuccess) throw Error(6, std::string("kernel ") + name + " failed: " + hipGetErrorString(e));
  }
}
#define OQ_LAUNCH(kern, grid, block, shmem, stream, ...)                    \
  do {                                                                      \
    hipLaunchKernelGGL(kern, grid, block, shmem, stream, __VA_ARGS__);      \
    oq::post_launch(#kern, stream);                                         \
  } while (0)

// ---- device buffers -------------------------------------------------------
extern size_t g_device_bytes;  // bytes currently allocated through DevBuf

template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  explicit DevBuf(size_t count) { alloc(count); }
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  DevBuf &operator=(DevBuf &&o) noexcept {
    if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  void alloc(size_t count) {
    release();
    n = count;
    size_t bytes = (count ? count : 1) * sizeof(T);
    HIP_CHECK(hipMalloc((void **)&p, bytes));
    g_device_bytes += bytes;
  }
  void release() {
    if (p) {
      (void)hipFree(p);
      g_device_bytes -= (n ? n : 1) * sizeof(T);
      p = nullptr;
      n = 0;
    }
  }
  void zero(hipStream_t s) { HIP_CHECK(hipMemsetAsync(p, 0, (n ? n : 1) * sizeof(T), s)); }
  void upload(const T *h, size_t count, hipStream_t s) {
    if (count) HIP_CHECK(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, s));
  }
  void download(T *h, size_t count, hipStream_t s) const {
    if (count) HIP_CHECK(hipMemcpyAsync(h, p, count * sizeof(T), hipMemcpyDeviceToHost, s));
  }
  T *get() const { return p; }
};

// ---- device CSR matrix: 64-bit row pointers, 32-bit column indices, fp64 values
// Column-panel copy of a CSR matrix for the LDS-staged SpMV (panel.hip): the columns are cut into
// B panels of W = 2^shift columns; entries are stored panel-major, row-minor, with 16-bit local column
// indices, so that a workgroup can keep its panel of x (W doubles) in LDS and gather from there.
struct DevPanel {
  bool active = false;
  int W = 0, shift = 0, B = 0, R = 0, T = 0;  // R rows per workgroup tile, T = ceil(rows / R) tiles
  DevBuf<uint32_t> pptr;                     // [B * rows + 1] start of (panel b, row i)
  DevBuf<uint16_t> pcol;                     // [nnz] column index inside the panel
  DevBuf<double> pval;                       // [nnz]
  DevBuf<double> partial;                    // [B * rows] per-panel row sums, reduced in fixed order
  // workgroup tiles: tile k = rows [tile_r0[k], tile_r1[k]) of panel tile_b[k], cut at ~equal nnz
  DevBuf<int> tile_b, tile_r0, tile_r1;
  int ntiles = 0;
  // sub-chunks of a tile for the streaming kernel: tile k owns entries [tile_sub0[k], tile_sub0[k] + tile_nsub[k]]
  // of sub_row / sub_k (row boundaries and their non-zero offsets, <= kPanelChunk non-zeros and < 1024 rows each)
  DevBuf<int> tile_sub0, tile_nsub, sub_row;
  DevBuf<uint32_t> sub_k;
  // sliced-ELL copy of every tile (panel_sell.hip): slice = 64 rows of the tile sorted by length, stored
  // column-major (element k of lane l at slice base + 64 k + l); tile k owns slices [tile_sub0[k], +tile_nsub[k]),
  // sub_k = slice base offset, sub_row = slice length, slice_rows = the 64 row ids of the slice (-1: empty lane)
  bool sell = false;
  DevBuf<uint32_t> cellbase;                 // [B * rows] base offset of (panel, row) inside sval / scol
  DevBuf<double> sval;
  DevBuf<uint16_t> scol;
  DevBuf<int> slice_rows;
  size_t padded = 0;
};

struct DevCsr {
  int rows = 0, cols = 0;
  int64_t nnz = 0;
  DevBuf<int64_t> rowptr;
  DevBuf<int> col;
  DevBuf<double> val;
  DevPanel panel;
  int group = 64;  // lanes per row chosen for the SpMV kernel (1..64)
  double spmv_bytes() const {  // algorithmic bytes of one y = M x
    return 12.0 * (double)nnz + 4.0 * ((double)rows + 1.0) + 8.0 * ((double)rows + (double)cols);
  }
};

// scalar slots written by the reduction kernels (device array of doubles)
enum Slot {
  S_PRI = 0, S_PRI_UNS, S_Z, S_AX, S_Z_UNS, S_AX_UNS,
  S_DUA, S_DUA_UNS, S_Q, S_ATY, S_PX, S_Q_UNS, S_ATY_UNS, S_PX_UNS,
  S_XPX, S_QX,
  S_T0, S_T1, S_T2, S_T3, S_T4, S_T5,   // scratch slots (infeasibility tests, scaling, PCG)
  S_COUNT = 32
};

constexpr int kBlock = 256;
constexpr int kReduceBlocks = 1024;  // fixed grid of the two-stage (deterministic) sum reductions

inline int blocks_for(int64_t n, int per_block = kBlock) {
  int64_t b = (n + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > 2147483647LL) throw Error(6, "grid too large");
  return (int)b;
}

}  // namespace oq
