// common.hpp -- shared declarations of the HIP engine (libosqp_amd.so).
// MI355X / gfx950 only: 64-lane wavefronts, 256 CUs in 8 XCDs, HBM3E.
#pragma once
#include <chrono>
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
// panel_build declines a matrix BEFORE it has released or rewritten anything (a layout it does not support, a copy that
// would be mostly padding): the caller keeps the matrix on its CSR arrays.  Any other Error out of panel_build (allocation,
// launch, synchronisation) can come after the CSR columns went and the nnz-index maps were folded into slot ids -- that one
// must reach the caller of osqp_setup as a failure, not be swallowed.
struct PanelRefused : Error {
  using Error::Error;
};
// a wait inside k_sn_tree timed out (direct.hip): the iterations since the last residual evaluation cannot be trusted.
// Engine::solve catches it, cold-starts and runs the solve again on the per-level form of the triangular solves.
// the columns of A handed to osqp_setup do not list their rows in ascending order (or repeat one): the ABI's setup entry point
// catches exactly this type and repeats the setup on a sorted host copy, as libosqp accepts such columns (abi.hip setup_from_host)
struct UnsortedColumns : Error {
  UnsortedColumns() : Error(1, "the row indices inside a column of A must ascend (and not repeat)") {}
};
extern thread_local bool g_unsorted_columns;  // set where UnsortedColumns is thrown, read (and cleared) by setup_from_host
struct TreeFault : Error {
  TreeFault() : Error(6, "internal: a supernode of the triangular solve waited 200 ms for its children (the solves fall back to one launch per level)") {}
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
extern size_t g_device_peak;   // high-water mark of the above since the process started
extern double g_alloc_s, g_free_s;  // wall time spent inside hipMalloc / hipFree by DevBuf (OSQP_AMD_SETUP_TRACE prints them)
extern size_t g_cache_bytes;  // blocks parked by dev_free inside a DevCacheScope (counted in the peak: they are still this process's)
inline double wall_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Device blocks of a setup.  A setup allocates and releases ~150 GB of temporaries for a 62 GB peak (rand-1e6); on the
// MI355X boxes of this pool a hipFree of a multi-GB block costs ~8 ms/GB and one hipMalloc in a few stalls for 1-5 s behind
// the driver's deferred release of earlier frees (profiles/r03_setup_alloc_stalls.txt).  Inside a DevCacheScope released
// blocks of >= 1 MiB are parked by size class (64 classes per octave, <= 1.6 % padding) and handed to the next request of
// the class; the scope's end (or dev_cache_trim) returns what is parked to the driver.  All work of a scope is on one
// stream (or separated by a device synchronisation), so a parked block is reused in stream order.
size_t dev_size_class(size_t bytes);
void *dev_alloc(size_t bytes, size_t &granted);
void dev_free(void *p, size_t granted);
void dev_cache_trim();
size_t dev_va_reserved();  // bytes of device address space reserved so far and never handed back (ranges are not reused: devmem.hip)
struct DevCacheScope {
  DevCacheScope();
  ~DevCacheScope();
  DevCacheScope(const DevCacheScope &) = delete;
  DevCacheScope &operator=(const DevCacheScope &) = delete;
};

template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  size_t cap = 0;  // bytes granted by dev_alloc (the size class of the request)
  DevBuf() = default;
  explicit DevBuf(size_t count) { alloc(count); }
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr; o.n = 0; o.cap = 0; }
  DevBuf &operator=(DevBuf &&o) noexcept {
    if (this != &o) { release(); p = o.p; n = o.n; cap = o.cap; o.p = nullptr; o.n = 0; o.cap = 0; }
    return *this;
  }
  ~DevBuf() { release(); }
  void alloc(size_t count) {
    release();
    n = count;
    p = (T *)dev_alloc((count ? count : 1) * sizeof(T), cap);
  }
  void release() {
    if (p) {
      dev_free(p, cap);
      p = nullptr;
      n = 0;
      cap = 0;
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
// Column-panel copy of a CSR matrix for the LDS-staged SpMV (panel.hip): the columns are cut into B panels of
// W = 2^shift columns so that a workgroup can keep a panel of x (W doubles) in LDS and gather from there; Gp consecutive
// panels form a group (NG groups).  Workgroup tile t = rows [tile_r0[t], tile_r1[t]) of group tile_g[t], cut at ~equal
// non-zero counts; unit t * Gp + j = the tile inside the j-th panel of its group, stored as sliced ELL: slice = 64 rows
// of the unit ordered by length, column-major (element k of lane l at slice_base + 64 k + l), 16-bit column ids local
// to the panel.
struct DevPanel {
  bool active = false;
  int W = 0, shift = 0, B = 0, Gp = 1, NG = 0, ntiles = 0;
  DevBuf<int> tile_g, tile_r0, tile_r1;
  DevBuf<int> unit_s0, unit_ns;              // unit u owns slices [unit_s0[u], unit_s0[u] + unit_ns[u])
  DevBuf<uint32_t> slice_base;               // offset of the slice inside sval / scol
  DevBuf<int> slice_len;                     // padded row length of the slice
  DevBuf<int> slice_rows;                    // [64 per slice] row id of every lane (-1: empty lane)
  DevBuf<uint32_t> cellbase;                 // [B * rows] slot of the first entry of (panel, row): value refresh
  DevBuf<double> sval;
  DevBuf<uint16_t> scol;
  bool wide = false;                         // panels of 2^18 columns gathered through L2 instead of staged in LDS
  DevBuf<uint32_t> scol32;                   // their local column ids
  DevBuf<double> partial;                    // [NG * rows] per-group row sums, reduced in fixed order
  size_t padded = 0;                         // stored entries including padding
};

struct DevCsr {
  int rows = 0, cols = 0;
  int64_t nnz = 0;
  DevBuf<int64_t> rowptr;
  DevBuf<int> col;
  DevBuf<double> val;
  DevPanel panel;
  bool compact = false;  // col / val released: the sliced-ELL copy in `panel` is the only one (panel.hip, compact mode)
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

// Where a kernel that has just produced scalars the host is waiting for puts them: pinned host memory mapped into the
// device's address space, a sequence number raised last (Engine::read_slots / begin_publish / wait_publish).
struct Publish {
  double *host_slots = nullptr;                  // the host mirror of the slot array (device-visible address); null: do not publish
  volatile unsigned long long *host_seq = nullptr;
  unsigned long long seq = 0;
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
