// devmem.hip -- where DevBuf's bytes come from.
//
// Large blocks (>= 256 MiB) are ranges of reserved device address space backed by 64 MiB physical chunks
// (hipMemCreate / hipMemMap): a released block gives its chunks back to a per-device pool, and the next block -- of ANY
// size -- is mapped from the pool.  A setup of rand-1e6 allocates and releases ~150 GB of nnz-sized temporaries for a
// 62 GB peak; through hipMalloc / hipFree the driver wipes released VRAM before it hands it out again, and on the MI355X
// boxes of this pool one hipMalloc in a few then waits 1-5 s for it (profiles/r03_setup_alloc_stalls.txt: a bare
// "hipMalloc 8 GB, hipFree, hipMalloc 8 GB" reproduces it).  With the pool nothing goes back to the driver before the
// end of the setup, the physical footprint is the live bytes rounded up to chunks, and mapping 8 GB costs ~0.5 ms.
// Small blocks stay with hipMalloc; inside a DevCacheScope those of >= 1 MiB are parked by size class and reused.
#include "common.hpp"
#include <map>
#include <mutex>

namespace oq {

size_t g_device_bytes = 0;
size_t g_device_peak = 0;
size_t g_cache_bytes = 0;
double g_alloc_s = 0., g_free_s = 0.;

namespace {
std::mutex g_mu;
int g_depth = 0;               // open scopes in the process: released chunks of mapped blocks go to the (shared) chunk pool
thread_local int t_depth = 0;  // open scopes of THIS thread: only its own frees are parked in its (thread-local) list -- a thread
                               // outside any scope would otherwise park blocks nobody trims (its list is not the scope owner's)
constexpr size_t kParkMin = size_t(1) << 20;
constexpr size_t kChunk = size_t(64) << 20;
const size_t kVmMin = getenv("OSQP_AMD_VMM_MIN_MB") ? (size_t)atol(getenv("OSQP_AMD_VMM_MIN_MB")) << 20 : size_t(256) << 20;
// debugging aid: every block handed out is filled with 0xA5 first (nothing may count on the zeros of a fresh hipMalloc)
const bool g_poison = getenv("OSQP_AMD_POISON") && atoi(getenv("OSQP_AMD_POISON")) == 1;

struct VmBlock { size_t size; int dev; std::vector<hipMemGenericAllocationHandle_t> chunks; };
std::map<void *, VmBlock> g_vm;                                           // live mapped ranges by base address
std::map<int, std::vector<hipMemGenericAllocationHandle_t>> g_chunk_pool;  // device -> unmapped physical chunks
thread_local std::map<std::pair<int, size_t>, std::vector<void *>> g_parked;  // (device, size class) -> small hipMalloc blocks; per
                                                                               // thread: a block is reused in ITS thread's stream order
int g_vm_state = (getenv("OSQP_AMD_VMM") && atoi(getenv("OSQP_AMD_VMM")) == 0) ? -1 : 0;  // 0 unprobed, 1 in use, -1 off

// A reserved range is never given back (and never handed to another block): on ROCm 7.0 / 7.2 a range that is unmapped,
// freed and reserved again -- or simply mapped again onto other chunks -- makes hipMemcpy / kernels through the new
// block hit the OLD chunks now and then (profiles/r03_setup_alloc_stalls.txt: osqp_update_P after a setup scatters into
// the wrong memory, silently).  Address space is what leaks: 128 TiB of it per process, i.e. ~800 setups of rand-1e6;
// when a reservation fails the block comes from hipMalloc.  OSQP_AMD_VMM_VA=0 frees ranges at once (the defect, for the record).
const bool g_va_free_at_once = getenv("OSQP_AMD_VMM_VA") && atoi(getenv("OSQP_AMD_VMM_VA")) == 0;

size_t g_va_reserved = 0;  // address space taken so far (never returned: see above); OSQP_AMD_ALLOC_TRACE prints it
// test hook: the reservation "fails" once this many MiB of address space have been taken (tests/test_devmem_gpu.py drives the
// allocator into its hipMalloc fallback in the middle of a setup and compares the results bit for bit)
const size_t g_va_limit = getenv("OSQP_AMD_VMM_VA_LIMIT_MB") ? (size_t)atol(getenv("OSQP_AMD_VMM_VA_LIMIT_MB")) << 20 : ~size_t(0);
void *va_get(size_t size) {
  void *va = nullptr;
  if (g_va_reserved + size > g_va_limit || hipMemAddressReserve(&va, size, 0, nullptr, 0) != hipSuccess) {
    (void)hipGetLastError();
    // Said once, loudly: from here on large blocks come from hipMalloc again -- correct, but with the driver's 1-5 s stalls
    // this allocator exists to avoid.  A long-lived process that sets up thousands of large workspaces gets here.
    static bool told = false;
    if (!told) {
      told = true;
      fprintf(stderr, "[osqp_amd] device allocator: no address range of %.2f GB left after %.1f TB of reservations; large blocks fall "
                      "back to hipMalloc for the rest of this process (slower setups, same results)\n", size / 1e9, g_va_reserved / 1e12);
    }
    return nullptr;
  }
  g_va_reserved += size;
  return va;
}
void va_retire(void *va, size_t size) {
  if (g_va_free_at_once) (void)hipMemAddressFree(va, size);
}
const bool g_alloc_trace = getenv("OSQP_AMD_ALLOC_TRACE") && atoi(getenv("OSQP_AMD_ALLOC_TRACE")) == 1;
void trace_block(const char *what, size_t granted) {
  if (g_alloc_trace && granted >= (size_t(64) << 20))
    fprintf(stderr, "[alloc] %-6s %7.3f GB   live %6.2f  pooled %6.2f\n", what, granted / 1e9, g_device_bytes / 1e9, g_cache_bytes / 1e9);
}
void note_peak() {
  if (g_device_bytes + g_cache_bytes > g_device_peak) g_device_peak = g_device_bytes + g_cache_bytes;
}

bool vm_usable(int dev) {
  if (g_vm_state == 0) {
    int ok = 0;
    g_vm_state = (hipDeviceGetAttribute(&ok, hipDeviceAttributeVirtualMemoryManagementSupported, dev) == hipSuccess && ok) ? 1 : -1;
    if (g_vm_state < 0) (void)hipGetLastError();
  }
  return g_vm_state > 0;
}

void trim_locked() {
  const double t0 = wall_now();
  for (auto &kv : g_parked)
    for (void *q : kv.second) { (void)hipFree(q); g_cache_bytes -= kv.first.second; }
  g_parked.clear();
  for (auto &kv : g_chunk_pool)
    for (auto h : kv.second) { (void)hipMemRelease(h); g_cache_bytes -= kChunk; }
  g_chunk_pool.clear();
  g_free_s += wall_now() - t0;
}

void vm_drop(void *va, size_t mapped, VmBlock &b) {  // undo a partly built block
  if (mapped) (void)hipMemUnmap(va, mapped);
  va_retire(va, b.size);
  auto &pool = g_chunk_pool[b.dev];
  for (auto h : b.chunks) { pool.push_back(h); g_cache_bytes += kChunk; }
}

void *vm_alloc(size_t bytes, size_t &granted, int dev) {
  VmBlock b;
  b.size = (bytes + kChunk - 1) / kChunk * kChunk;
  b.dev = dev;
  const size_t nch = b.size / kChunk;
  void *va = va_get(b.size);
  if (!va) return nullptr;  // address space exhausted: the caller falls back to hipMalloc
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  auto &pool = g_chunk_pool[dev];
  size_t fresh = 0;
  for (size_t i = 0; i < nch; i++) {
    hipMemGenericAllocationHandle_t h;
    if (!pool.empty()) {
      h = pool.back();
      pool.pop_back();
      g_cache_bytes -= kChunk;
    } else {
      hipError_t e = hipMemCreate(&h, kChunk, &prop, 0);
      if (e != hipSuccess && !g_parked.empty()) {  // what this thread has parked may be what is missing
        (void)hipGetLastError();
        for (auto &kv : g_parked)
          for (void *q : kv.second) { (void)hipFree(q); g_cache_bytes -= kv.first.second; }
        g_parked.clear();
        e = hipMemCreate(&h, kChunk, &prop, 0);
      }
      if (e != hipSuccess) {
        (void)hipGetLastError();
        vm_drop(va, i * kChunk, b);
        throw Error(6, std::string("device memory exhausted: hipMemCreate failed (") + hipGetErrorString(e) + ") with " +
                           std::to_string(g_device_bytes >> 20) + " MiB live");
      }
      fresh++;
    }
    b.chunks.push_back(h);
    hipError_t e = hipMemMap((char *)va + i * kChunk, kChunk, 0, h, 0);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      vm_drop(va, i * kChunk, b);
      throw Error(6, std::string("hipMemMap failed: ") + hipGetErrorString(e));
    }
  }
  hipMemAccessDesc acc{};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  hipError_t e = hipMemSetAccess(va, b.size, &acc, 1);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    vm_drop(va, b.size, b);
    throw Error(6, std::string("hipMemSetAccess failed: ") + hipGetErrorString(e));
  }
  granted = b.size;
  g_device_bytes += granted;
  note_peak();
  trace_block(fresh ? (fresh == nch ? "fresh" : "part") : "pool", granted);
  g_vm.emplace(va, std::move(b));
  return va;
}

bool vm_free(void *p) {
  auto it = g_vm.find(p);
  if (it == g_vm.end()) return false;
  VmBlock &b = it->second;
  // the range disappears at once: everything queued on ITS device that may touch it has to be through (hipFree waits too).
  // dev_free has drained the device BEFORE taking the allocator's lock (vm_drain), so that other threads' allocations do
  // not wait behind a device drain.
  int cur = b.dev;
  (void)hipGetDevice(&cur);
  if (cur != b.dev) (void)hipSetDevice(b.dev);
  hipError_t e1 = hipMemUnmap(p, b.size);
  if (cur != b.dev) (void)hipSetDevice(cur);
  if (e1 != hipSuccess) fprintf(stderr, "[osqp-amd] hipMemUnmap: %s\n", hipGetErrorString(e1));
  va_retire(p, b.size);
  g_device_bytes -= b.size;
  if (g_depth > 0) {
    auto &pool = g_chunk_pool[b.dev];
    for (auto h : b.chunks) pool.push_back(h);
    g_cache_bytes += b.size;
    trace_block("unmap", b.size);
  } else {
    for (auto h : b.chunks) (void)hipMemRelease(h);
    trace_block("free", b.size);
  }
  g_vm.erase(it);
  return true;
}
}  // namespace

size_t dev_size_class(size_t bytes) {
  if (bytes < kParkMin) return (bytes + 255) & ~size_t(255);
  const int lg = 63 - __builtin_clzll((unsigned long long)bytes);
  const size_t g = size_t(1) << (lg - 6);  // 64 classes per octave: <= 1.6 % padding
  return (bytes + g - 1) & ~(g - 1);
}

static void *dev_alloc_raw(size_t bytes, size_t &granted);
void *dev_alloc(size_t bytes, size_t &granted) {
  void *p = dev_alloc_raw(bytes, granted);
  if (g_poison) {
    HIP_CHECK(hipMemset(p, 0xA5, granted));
    HIP_CHECK(hipDeviceSynchronize());
  }
  return p;
}
static void *dev_alloc_raw(size_t bytes, size_t &granted) {
  std::lock_guard<std::mutex> lock(g_mu);
  int dev = 0;
  HIP_CHECK(hipGetDevice(&dev));
  const double t0 = wall_now();
  if (bytes >= kVmMin && vm_usable(dev)) {
    void *p = vm_alloc(bytes, granted, dev);
    if (p) {
      g_alloc_s += wall_now() - t0;
      return p;
    }
  }
  granted = dev_size_class(bytes);
  void *p = nullptr;
  if (granted >= kParkMin && !g_parked.empty()) {
    auto it = g_parked.find({dev, granted});
    if (it != g_parked.end() && !it->second.empty()) {
      p = it->second.back();
      it->second.pop_back();
      g_cache_bytes -= granted;
      g_device_bytes += granted;
      return p;
    }
  }
  hipError_t e = hipMalloc(&p, granted);
  if (e == hipErrorOutOfMemory && g_cache_bytes) {  // what is parked may be what is missing
    (void)hipGetLastError();
    trim_locked();
    e = hipMalloc(&p, granted);
  }
  g_alloc_s += wall_now() - t0;
  HIP_CHECK(e);
  g_device_bytes += granted;
  note_peak();
  trace_block("fresh", granted);
  return p;
}

// a mapped block is about to be unmapped: wait, outside the lock, for the work queued on its device
static void vm_drain(void *p) {
  int dev = -1;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_vm.find(p);
    if (it != g_vm.end()) dev = it->second.dev;
  }
  if (dev < 0) return;
  int cur = dev;
  (void)hipGetDevice(&cur);
  if (cur != dev) (void)hipSetDevice(dev);
  (void)hipDeviceSynchronize();
  if (cur != dev) (void)hipSetDevice(cur);
}

void dev_free(void *p, size_t granted) {
  if (!p) return;
  vm_drain(p);
  std::lock_guard<std::mutex> lock(g_mu);
  const double t0 = wall_now();
  if (vm_free(p)) {
    g_free_s += wall_now() - t0;
    return;
  }
  g_device_bytes -= granted;
  if (t_depth > 0 && granted >= kParkMin) {
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) {  // reused in stream order: a scope's work is on one stream
      g_parked[{dev, granted}].push_back(p);
      g_cache_bytes += granted;
      return;
    }
  }
  (void)hipFree(p);
  g_free_s += wall_now() - t0;
}

size_t dev_va_reserved() { return g_va_reserved; }
void dev_cache_trim() {
  std::lock_guard<std::mutex> lock(g_mu);
  trim_locked();
}

DevCacheScope::DevCacheScope() {
  std::lock_guard<std::mutex> lock(g_mu);
  ++g_depth;
  ++t_depth;
}
DevCacheScope::~DevCacheScope() {
  std::lock_guard<std::mutex> lock(g_mu);
  --g_depth;
  --t_depth;
  trim_locked();
}

}  // namespace oq
