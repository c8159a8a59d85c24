// probe: cost of the HIP virtual-memory calls on this box (chunked physical memory mapped into reserved ranges)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_fill(double *p, size_t n, double v) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; size_t st = (size_t)gridDim.x * blockDim.x; for (; i < n; i += st) p[i] = v; }
__global__ void k_sum(const double *p, size_t n, double *out) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; size_t st = (size_t)gridDim.x * blockDim.x; double s = 0; for (; i < n; i += st) s += p[i]; if (s == 12345.678) out[0] = s; }
int main() {
  int dev = 0; CK(hipSetDevice(0));
  hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
  size_t gmin = 0, grec = 0;
  CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
  CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
  printf("granularity min %zu recommended %zu\n", gmin, grec);
  hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  double *out; CK(hipMalloc(&out, 8));
  for (size_t chunk : {size_t(64) << 20, size_t(256) << 20, size_t(1) << 30}) {
    const size_t total = size_t(8) << 30; const int nch = (int)(total / chunk);
    std::vector<hipMemGenericAllocationHandle_t> h(nch);
    double t0 = now();
    for (int i = 0; i < nch; i++) CK(hipMemCreate(&h[i], chunk, &prop, 0));
    double t1 = now();
    for (int rep = 0; rep < 3; rep++) {
      void *va = nullptr; double a0 = now();
      CK(hipMemAddressReserve(&va, total, 0, nullptr, 0));
      double a1 = now();
      for (int i = 0; i < nch; i++) CK(hipMemMap((char *)va + i * chunk, chunk, 0, h[i], 0));
      double a2 = now();
      CK(hipMemSetAccess(va, total, &acc, 1));
      double a3 = now();
      hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (double *)va, total / 8, 1.0); CK(hipDeviceSynchronize());
      double a4 = now();
      hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (double *)va, total / 8, 2.0); CK(hipDeviceSynchronize());
      double a5 = now();
      hipLaunchKernelGGL(k_sum, dim3(4096), dim3(256), 0, 0, (const double *)va, total / 8, out); CK(hipDeviceSynchronize());
      double a6 = now();
      CK(hipMemUnmap(va, total));
      double a7 = now();
      CK(hipMemAddressFree(va, total));
      double a8 = now();
      printf("chunk %4zu MB rep %d: reserve %.2f ms, map x%d %.2f ms, setaccess %.2f ms, first fill %.2f ms (%.0f GB/s), second fill %.2f ms (%.0f GB/s), read %.2f ms (%.0f GB/s), unmap %.2f ms, addrfree %.2f ms\n",
             chunk >> 20, rep, 1e3 * (a1 - a0), nch, 1e3 * (a2 - a1), 1e3 * (a3 - a2), 1e3 * (a4 - a3), total / (a4 - a3) / 1e9, 1e3 * (a5 - a4), total / (a5 - a4) / 1e9,
             1e3 * (a6 - a5), total / (a6 - a5) / 1e9, 1e3 * (a7 - a6), 1e3 * (a8 - a7));
    }
    double t2 = now();
    for (int i = 0; i < nch; i++) CK(hipMemRelease(h[i]));
    double t3 = now();
    printf("chunk %4zu MB: create x%d %.2f ms, release %.2f ms\n", chunk >> 20, nch, 1e3 * (t1 - t0), 1e3 * (t3 - t2));
  }
  // the same with hipMalloc
  for (int rep = 0; rep < 3; rep++) {
    const size_t total = size_t(8) << 30; void *p; double a0 = now(); CK(hipMalloc(&p, total)); double a1 = now();
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (double *)p, total / 8, 1.0); CK(hipDeviceSynchronize()); double a2 = now();
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (double *)p, total / 8, 2.0); CK(hipDeviceSynchronize()); double a3 = now();
    hipLaunchKernelGGL(k_sum, dim3(4096), dim3(256), 0, 0, (const double *)p, total / 8, out); CK(hipDeviceSynchronize()); double a4 = now();
    CK(hipFree(p)); double a5 = now();
    printf("hipMalloc 8 GB rep %d: malloc %.2f ms, first fill %.2f ms, second fill %.2f ms (%.0f GB/s), read %.2f ms (%.0f GB/s), free %.2f ms\n", rep, 1e3 * (a1 - a0), 1e3 * (a2 - a1), 1e3 * (a3 - a2),
           total / (a3 - a2) / 1e9, 1e3 * (a4 - a3), total / (a4 - a3) / 1e9, 1e3 * (a5 - a4));
  }
  return 0;
}
