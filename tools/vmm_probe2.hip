// probe: do hipMemset / hipMemcpy work across the physical chunks of one mapped range?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k_fill(long *p, size_t n, long v) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; size_t st = (size_t)gridDim.x * blockDim.x; for (; i < n; i += st) p[i] = v + (long)i; }
__global__ void k_count_ne(const long *p, size_t n, long v, int ramp, unsigned long long *out) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; size_t st = (size_t)gridDim.x * blockDim.x; unsigned long long c = 0; for (; i < n; i += st) c += p[i] != v + (ramp ? (long)i : 0); if (c) atomicAdd(out, c); }
static void *vmalloc(size_t total, size_t chunk, std::vector<hipMemGenericAllocationHandle_t> &h) {
  hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  void *va = nullptr; if (hipMemAddressReserve(&va, total, 0, nullptr, 0) != hipSuccess) return nullptr;
  for (size_t o = 0; o < total; o += chunk) { hipMemGenericAllocationHandle_t x; if (hipMemCreate(&x, chunk, &prop, 0) != hipSuccess) return nullptr; h.push_back(x); if (hipMemMap((char *)va + o, chunk, 0, x, 0) != hipSuccess) return nullptr; }
  if (hipMemSetAccess(va, total, &acc, 1) != hipSuccess) return nullptr;
  return va;
}
int main() {
  CK(hipSetDevice(0));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const size_t chunk = size_t(64) << 20, total = 4 * chunk, n = total / 8;
  std::vector<hipMemGenericAllocationHandle_t> ha, hb;
  long *a = (long *)vmalloc(total, chunk, ha), *b = (long *)vmalloc(total, chunk, hb);
  if (!a || !b) { printf("vmalloc failed\n"); return 1; }
  unsigned long long *cnt; CK(hipMalloc(&cnt, 8));
  auto count = [&](const long *p, size_t len, long v, int ramp) { hipMemset(cnt, 0, 8); hipLaunchKernelGGL(k_count_ne, dim3(1024), dim3(256), 0, 0, p, len, v, ramp, cnt); unsigned long long c = 0; hipMemcpy(&c, cnt, 8, hipMemcpyDeviceToHost); return c; };
  hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, a, n, 7L); CK(hipDeviceSynchronize());
  printf("kernel fill across chunks: %llu mismatches\n", count(a, n, 7, 1));
  hipError_t e = hipMemsetAsync(a, 0, total, s); printf("hipMemsetAsync whole range: %s\n", hipGetErrorString(e)); CK(hipStreamSynchronize(s));
  printf("  after memset: %llu non-zero words of %zu\n", count(a, n, 0, 0), n);
  hipLaunchKernelGGL(k_fill, dim3(1024), dim3(256), 0, 0, a, n, 7L); CK(hipDeviceSynchronize());
  e = hipMemsetAsync((char *)a + chunk / 2, 0, chunk, s); printf("hipMemsetAsync straddling one boundary: %s\n", hipGetErrorString(e)); CK(hipStreamSynchronize(s));
  printf("  words zero in the straddling window: %llu non-zero of %zu\n", count(a + chunk / 16, chunk / 8, 0, 0), chunk / 8);
  std::vector<long> host(n); for (size_t i = 0; i < n; i++) host[i] = 100 + (long)i;
  e = hipMemcpyAsync(a, host.data(), total, hipMemcpyHostToDevice, s); printf("H2D whole range: %s\n", hipGetErrorString(e)); CK(hipStreamSynchronize(s));
  printf("  after H2D: %llu mismatches\n", count(a, n, 100, 1));
  e = hipMemcpyAsync(b, a, total, hipMemcpyDeviceToDevice, s); printf("D2D whole range: %s\n", hipGetErrorString(e)); CK(hipStreamSynchronize(s));
  printf("  after D2D: %llu mismatches\n", count(b, n, 100, 1));
  e = hipMemcpyAsync((char *)b + chunk / 2, (char *)a + chunk + chunk / 4, chunk, hipMemcpyDeviceToDevice, s); printf("D2D straddling: %s\n", hipGetErrorString(e)); CK(hipStreamSynchronize(s));
  std::vector<long> back(n, -1);
  e = hipMemcpyAsync(back.data(), b, total, hipMemcpyDeviceToHost, s); printf("D2H whole range: %s\n", hipGetErrorString(e)); CK(hipStreamSynchronize(s));
  size_t bad = 0; for (size_t i = 0; i < n; i++) { long want = 100 + (long)i; size_t lo = chunk / 16, hi = lo + chunk / 8; if (i >= lo && i < hi) want = 100 + (long)(i - lo + chunk / 8 + chunk / 32); bad += back[i] != want; }
  printf("  after D2H (with the straddling D2D applied): %zu mismatches\n", bad);
  return 0;
}
