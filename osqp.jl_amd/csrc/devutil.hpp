// devutil.hpp -- device-side reduction helpers shared by the kernel files (kernels.hip, pcg.hip).
// Sums are always formed in one fixed order (wavefront xor-shuffle tree, then the four waves of a 256-thread block, then
// the kReduceBlocks block partials strided over a block): every kernel that needs a scalar recombines the same partials
// with the same functions, so a fused kernel and the unfused sequence it replaces produce the same bits.
#pragma once
#include "common.hpp"

namespace oq {

__device__ __forceinline__ double nanmax(double a, double b) { return (a > b || a != a) ? a : b; }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = nanmax(v, __shfl_xor(v, o, 64));
  return v;
}
// all threads of a 256-thread block get the block total (fixed order -> deterministic)
__device__ __forceinline__ double block_sum(double v) {
  __shared__ double sm[4];
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}
__device__ __forceinline__ double block_max(double v) {
  __shared__ double smx[4];
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) smx[threadIdx.x >> 6] = v;
  __syncthreads();
  return nanmax(nanmax(smx[0], smx[1]), nanmax(smx[2], smx[3]));
}
// max of non-negative doubles through their (monotone) bit pattern; NaN sorts above +inf
__device__ __forceinline__ void atomic_max_nonneg(double *addr, double v) {
  atomicMax((unsigned long long *)addr, (unsigned long long)__double_as_longlong(v));
}
// sum of the kReduceBlocks partials, same order in every block
__device__ __forceinline__ double sum_partials(const double *partials) {
  double v = 0.0;
  for (int i = threadIdx.x; i < kReduceBlocks; i += kBlock) v += partials[i];
  return block_sum(v);
}


// max over kReduceBlocks block partials, same value in every thread of the block
__device__ __forceinline__ double max_partials(const double *partials) {
  double v = 0.0;
  for (int i = threadIdx.x; i < kReduceBlocks; i += kBlock) v = nanmax(v, partials[i]);
  return block_max(v);
}

}  // namespace oq
