// Shared device/host helpers for libicm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/icm_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define ICM_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return ICM_ERR_LAUNCH;            \
  } while (0)

namespace icm {

__device__ __forceinline__ float gelu_f(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float dgelu_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
  return cdf + x * pdf;
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == ICM_ACT_GELU) return gelu_f(v);
  if (act == ICM_ACT_SQUARE) return v * v;
  return v;
}

// exact n / d for n < 2^16 * ... (n * d < 2^32 region used here: n < 65536, d < 65536)
struct FastDiv {
  uint32_t d, magic;
};
inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  f.magic = (d <= 1) ? 0u : (uint32_t)(((1ull << 32) + d - 1) / d);
  return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
  return f.d <= 1 ? n : __umulhi(n, f.magic);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

inline int ceil_log2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}
inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace icm
