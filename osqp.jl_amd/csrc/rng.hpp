// rng.hpp -- counter-based generator shared by the device problem generators
// (gen.hip, batch.hip).  Same bits as the host statement in oracle/gen.c.
#pragma once
#include <hip/hip_runtime.h>

namespace oq {

enum { G_AROW = 1, G_AVAL = 2, G_UROW = 3, G_UVAL = 4, G_Q = 5, G_L = 6, G_U = 7, G_PDIAG = 8,
       G_MPC_A = 9, G_MPC_B = 10, G_MPC_X0 = 11, G_MPC_REF = 12 };

__host__ __device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ unsigned long long rnd(unsigned long long seed, unsigned long long stream, unsigned long long idx) {
  unsigned long long k = mix64(seed + 0x9E3779B97F4A7C15ULL * (stream + 1));
  return mix64(k ^ (idx * 0xD1B54A32D192ED03ULL + 0x8CB92BA72F3D8DD7ULL));
}
__host__ __device__ __forceinline__ double u01(unsigned long long seed, unsigned long long stream, unsigned long long idx) {
  return ((double)(rnd(seed, stream, idx) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}
#define OQ_GAUSS_K (1.7320508075688772 / 65536.0)
__host__ __device__ __forceinline__ long long gauss_int(unsigned long long r) {
  return (long long)((r & 0xFFFF) + ((r >> 16) & 0xFFFF) + ((r >> 32) & 0xFFFF) + ((r >> 48) & 0xFFFF)) - 131070;
}
__host__ __device__ __forceinline__ double gauss(unsigned long long seed, unsigned long long stream, unsigned long long idx) {
  return (double)gauss_int(rnd(seed, stream, idx)) * OQ_GAUSS_K;
}

}  // namespace oq
