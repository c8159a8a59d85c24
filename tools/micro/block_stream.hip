// How fast can a wavefront-per-block kernel stream packed blocks of a few KB each?  (direct.hip k_sn_level_w reads 246 MB of
// inverted diagonal blocks -- 34 000 of 6.5 KB on average -- in 75 us = 3.3 TB/s whatever the form of the loads.)
// Variants: doubles per block, waves per workgroup, loads in flight, with / without a dependent pointer load in front.
//   hipcc --offload-arch=gfx950 -O3 -o block_stream block_stream.hip && ./block_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int U, bool kChain>
__global__ __launch_bounds__(256) void k_stream(int nblk, int per, const long long *__restrict__ off, const double *__restrict__ W,
                                                double *__restrict__ out) {
  const int wv = threadIdx.x >> 6, gl = threadIdx.x & 63;
  const int J = blockIdx.x * (blockDim.x >> 6) + wv;
  if (J >= nblk) return;
  const double *p = W + (kChain ? off[J] : (long long)J * per);
  double acc = 0.0;
  int e = gl;
  for (; e + (U - 1) * 64 < per; e += U * 64) {
    double v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = p[e + u * 64];
#pragma unroll
    for (int u = 0; u < U; u++) acc += v[u];
  }
  for (; e < per; e += 64) acc += p[e];
  if (acc == 123.456) out[J] = acc;  // never (the data are zeros): no store traffic, the loads stay
}

// the access pattern of the block product: a packed lower triangle of s rows read column by column, lanes j..s-1 of load j
template <int U>
__global__ __launch_bounds__(256) void k_tri(int nblk, int s, const double *__restrict__ W, double *__restrict__ out) {
  const int wv = threadIdx.x >> 6, gl = threadIdx.x & 63;
  const int J = blockIdx.x * (blockDim.x >> 6) + wv;
  if (J >= nblk) return;
  const double *p = W + (long long)J * (s * (s + 1) / 2);
  auto w = [&](int j) { return (gl < s && j <= gl) ? p[j * s - j * (j - 1) / 2 + (gl - j)] : 0.0; };
  double acc = 0.0;
  int j = 0;
  for (; j + U - 1 < s; j += U) {
    double v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = w(j + u);
#pragma unroll
    for (int u = 0; u < U; u++) acc += v[u];
  }
  for (; j < s; j++) acc += w(j);
  if (acc == 123.456) out[J] = acc;
}
// ... the same triangle FOLDED: lane a < s / 2 owns rows a and s - 1 - a (s + 1 entries together), a half wavefront per block,
// s + 1 loads of s / 2 lanes each, none masked
template <int U>
__global__ __launch_bounds__(256) void k_fold(int nblk, int s, const double *__restrict__ W, double *__restrict__ out) {
  const int wv = threadIdx.x >> 6, gl = threadIdx.x & 63, half = gl >> 5, a = gl & 31;
  const int J = (blockIdx.x * (blockDim.x >> 6) + wv) * 2 + half;
  const int h = (s + 1) / 2;
  const bool live = J < nblk && a < h;
  const double *p = W + (long long)(live ? J : 0) * (s * (s + 1) / 2);
  double acc = 0.0;
  int k = 0;
  for (; k + U - 1 < s + 1; k += U) {
    double v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = live ? p[(k + u) * h + a] : 0.0;
#pragma unroll
    for (int u = 0; u < U; u++) acc += v[u];
  }
  for (; k < s + 1; k++) acc += live ? p[k * h + a] : 0.0;
  if (acc == 123.456) out[J] = acc;
}
template <int U, bool kFold>
float run_tri(int nblk, int s, const double *W, double *out) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int waves = kFold ? (nblk + 1) / 2 : nblk, grid = (waves + 3) / 4;
  for (int i = 0; i < 3; i++) { if (kFold) k_fold<U><<<grid, 256>>>(nblk, s, W, out); else k_tri<U><<<grid, 256>>>(nblk, s, W, out); }
  hipEventRecord(a);
  for (int i = 0; i < 20; i++) { if (kFold) k_fold<U><<<grid, 256>>>(nblk, s, W, out); else k_tri<U><<<grid, 256>>>(nblk, s, W, out); }
  hipEventRecord(b); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  return ms / 20.0f;
}

template <int U, bool kChain>
float run(int nblk, int per, int wpb, const long long *off, const double *W, double *out) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int grid = (nblk + wpb - 1) / wpb;
  for (int i = 0; i < 3; i++) k_stream<U, kChain><<<grid, wpb * 64>>>(nblk, per, off, W, out);
  hipEventRecord(a);
  for (int i = 0; i < 20; i++) k_stream<U, kChain><<<grid, wpb * 64>>>(nblk, per, off, W, out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  return ms / 20.0f;
}

int main() {
  const size_t total = (size_t)1 << 27;  // 1 GB of doubles: several times the 256 MB of last-level cache
  double *W, *out; long long *off;
  CK(hipMalloc(&W, total * 8)); CK(hipMemset(W, 0, total * 8));
  CK(hipMalloc(&out, 8 << 20)); CK(hipMalloc(&off, 8 << 20));
  for (int s : {16, 24, 32, 40, 48, 64}) {
    const int per = s * (s + 1) / 2;
    for (int nblk : {34000, 340000}) {
      if ((size_t)nblk * per > total) continue;
      const double gb = (double)nblk * per * 8 / 1e9;
      const float t4 = run_tri<4, false>(nblk, s, W, out), t8 = run_tri<8, false>(nblk, s, W, out);
      const float f4 = run_tri<4, true>(nblk, s, W, out), f8 = run_tri<8, true>(nblk, s, W, out);
      printf("triangle of %2d rows x %6d (%.0f MB): by columns 4 in flight %.0f GB/s (%.1f us), 8: %.0f; folded 4: %.0f GB/s (%.1f us), 8: %.0f\n", s, nblk,
             gb * 1e3, gb / t4 * 1e3, t4 * 1e3, gb / t8 * 1e3, gb / f4 * 1e3, f4 * 1e3, gb / f8 * 1e3);
    }
  }
  for (int per : {832, 2080}) {
    const int nblk = (int)(total / per) > (1 << 20) ? (1 << 20) : (int)(total / per);
    std::vector<long long> h(nblk);
    for (int j = 0; j < nblk; j++) h[j] = (long long)j * per;
    CK(hipMemcpy(off, h.data(), nblk * 8, hipMemcpyHostToDevice));
    const double gb = (double)nblk * per * 8 / 1e9;
    for (int wpb : {1, 4}) {
      const float t4 = run<4, false>(nblk, per, wpb, off, W, out), t8 = run<8, false>(nblk, per, wpb, off, W, out);
      const float t16 = run<16, false>(nblk, per, wpb, off, W, out), c8 = run<8, true>(nblk, per, wpb, off, W, out);
      printf("block %6d doubles x %7d, %d waves/workgroup: 4 in flight %.0f GB/s, 8: %.0f, 16: %.0f, 8 behind a pointer load: %.0f\n", per, nblk,
             wpb, gb / t4 * 1e3, gb / t8 * 1e3, gb / t16 * 1e3, gb / c8 * 1e3);
    }
  }
  return 0;
}
