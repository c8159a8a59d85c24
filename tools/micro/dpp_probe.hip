// dpp_probe.hip -- micro-measurements behind the round-4 batched kernel (csrc/batch2.hip):
//   (a) rate of v_fmac_f64_dpp row_newbcast against plain v_fma_f64 (VGPR and SGPR operand) and against ds_read_b128-broadcast + fma
//   (b) cost of a two-wavefront workgroup barrier with an LDS hand-over (write -> s_barrier -> read)
// build: hipcc --offload-arch=gfx950 -O3 dpp_probe.hip -o dpp_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define FMAC_DPP(acc, b, m, L) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #L " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(b), "v"(m))
#define ROW16(acc, b, m, o) \
  FMAC_DPP(acc[0], b, m[o+0], 0); FMAC_DPP(acc[1], b, m[o+1], 1); FMAC_DPP(acc[2], b, m[o+2], 2); FMAC_DPP(acc[3], b, m[o+3], 3); \
  FMAC_DPP(acc[0], b, m[o+4], 4); FMAC_DPP(acc[1], b, m[o+5], 5); FMAC_DPP(acc[2], b, m[o+6], 6); FMAC_DPP(acc[3], b, m[o+7], 7); \
  FMAC_DPP(acc[0], b, m[o+8], 8); FMAC_DPP(acc[1], b, m[o+9], 9); FMAC_DPP(acc[2], b, m[o+10], 10); FMAC_DPP(acc[3], b, m[o+11], 11); \
  FMAC_DPP(acc[0], b, m[o+12], 12); FMAC_DPP(acc[1], b, m[o+13], 13); FMAC_DPP(acc[2], b, m[o+14], 14); FMAC_DPP(acc[3], b, m[o+15], 15);

// mode 0: dpp, 1: plain fma with VGPR b, 2: fma with SGPR b (readfirstlane once), 3: ds_read_b128 broadcast + fma
template <int MODE>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_rate(double *out, const double *in, int iters, long long *cyc) {
  __shared__ double lb[128];
  double m[96];
#pragma unroll
  for (int u = 0; u < 96; u++) m[u] = in[(threadIdx.x + u * 128) & 4095];
  double b[6];
#pragma unroll
  for (int k = 0; k < 6; k++) b[k] = in[(threadIdx.x & 15) + 16 * k];
  if (threadIdx.x < 128) lb[threadIdx.x] = in[threadIdx.x];
  __syncthreads();
  double acc[4] = {0, 0, 0, 0};
  long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {
      ROW16(acc, b[0], m, 0) ROW16(acc, b[1], m, 16) ROW16(acc, b[2], m, 32) ROW16(acc, b[3], m, 48) ROW16(acc, b[4], m, 64) ROW16(acc, b[5], m, 80)
    } else if (MODE == 1) {
#pragma unroll
      for (int u = 0; u < 96; u++) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(b[u >> 4]), "v"(m[u]));
    } else if (MODE == 2) {
      double sb[6];
#pragma unroll
      for (int k = 0; k < 6; k++) { union { double d; int i[2]; } q; q.d = b[k]; q.i[0] = __builtin_amdgcn_readfirstlane(q.i[0]); q.i[1] = __builtin_amdgcn_readfirstlane(q.i[1]); sb[k] = q.d; }
#pragma unroll
      for (int u = 0; u < 96; u++) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "s"(sb[u >> 4]), "v"(m[u]));
    } else {
      typedef double d2 __attribute__((ext_vector_type(2)));
      d2 bv[48];
#pragma unroll
      for (int u = 0; u < 48; u++) bv[u] = *(volatile d2 *)(lb + 2 * u);
#pragma unroll
      for (int u = 0; u < 96; u++) acc[u & 3] = __builtin_fma(m[u], (u & 1) ? bv[u >> 1].y : bv[u >> 1].x, acc[u & 3]);
    }
#pragma unroll
    for (int k = 0; k < 6; k++) b[k] += acc[0] * 1e-300;  // keep a loop-carried dependence so nothing is hoisted
  }
  long long t1 = clock64();
  out[blockIdx.x * 128 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// two-wavefront hand-over: wave (it & 1) writes 100 doubles (one lane, 50 x b128), barrier, everybody reads 7 doubles
__global__ __launch_bounds__(128) void k_barrier(double *out, int iters, long long *cyc) {
  __shared__ __attribute__((aligned(16))) double buf[2][128];
  const int t = threadIdx.x;
  double v = t;
  long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
    const int p = (it * 37) & 127;
    if (t == p) {
#pragma unroll
      for (int u = 0; u < 100; u++) buf[it & 1][u] = v + u;
    }
    __syncthreads();
    double s = 0;
#pragma unroll
    for (int k = 0; k < 7; k++) s += buf[it & 1][(t & 15) + 16 * k];
    v = v * 0.5 + s * 1e-3;
  }
  long long t1 = clock64();
  out[blockIdx.x * 128 + t] = v;
  if (t == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// numeric check of the row_newbcast semantics: out[lane] = m * b[lane of the 16-row broadcast]
__global__ void k_sem(double *out, const double *in) {
  double acc = 0.0, b = in[threadIdx.x], m = 1.0;
  FMAC_DPP(acc, b, m, 5);
  out[threadIdx.x] = acc;
}

int main() {
  double *in, *out; long long *cyc;
  CK(hipMalloc(&in, 4096 * 8)); CK(hipMalloc(&out, 1 << 22)); CK(hipMalloc(&cyc, 64));
  std::vector<double> h(4096);
  for (int i = 0; i < 4096; i++) h[i] = 1.0 + 1e-3 * i;
  CK(hipMemcpy(in, h.data(), 4096 * 8, hipMemcpyHostToDevice));
  {
    k_sem<<<1, 64>>>(out, in);
    double r[64]; CK(hipMemcpy(r, out, 64 * 8, hipMemcpyDeviceToHost));
    printf("row_newbcast:5 -> lanes 0,1,17,40,63 read %.3f %.3f %.3f %.3f %.3f (expect in[5], in[5], in[21], in[37], in[53] = %.3f %.3f %.3f %.3f)\n", r[0], r[1], r[17], r[40], r[63], h[5], h[21], h[37], h[53]);
  }
  const int iters = 2000;
  const char *names[4] = {"v_fmac_f64_dpp row_newbcast", "v_fma_f64 vgpr b", "v_fma_f64 sgpr b", "ds_read_b128 bcast + fma"};
  for (int grid : {256, 2048}) {  // 1 workgroup (2 waves) per CU / 8 per CU (2 waves per SIMD)
    for (int mode = 0; mode < 4; mode++) {
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(e0));
        if (mode == 0) k_rate<0><<<grid, 128>>>(out, in, iters, cyc);
        if (mode == 1) k_rate<1><<<grid, 128>>>(out, in, iters, cyc);
        if (mode == 2) k_rate<2><<<grid, 128>>>(out, in, iters, cyc);
        if (mode == 3) k_rate<3><<<grid, 128>>>(out, in, iters, cyc);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      }
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
      printf("grid %4d  %-30s  %.3f ms   %.1f clock64 ticks per 96 FMAs (wave 0)   %.2f ns per wave-FMA per SIMD-slot\n", grid, names[mode], ms, (double)c / iters,
             ms * 1e6 / ((double)iters * 96 * (grid * 2 / 1024.0 < 1 ? 1 : grid * 2 / 1024.0)));
    }
  }
  for (int grid : {256, 1024}) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; rep++) { CK(hipEventRecord(e0)); k_barrier<<<grid, 128>>>(out, 2000, cyc); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    printf("grid %4d  two-wave hand-over (100-double row, barrier, 7 reads): %.3f ms, %.1f ticks, %.1f ns per round\n", grid, ms, (double)c / 2000, ms * 1e6 / 2000);
  }
  return 0;
}
