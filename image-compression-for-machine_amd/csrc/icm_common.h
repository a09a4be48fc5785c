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

// ---- halo-patch staging shared by the implicit-GEMM and wgrad kernels ------------------------------------
// The patch of `nch` channels x TI images x PH x PW input pixels is written to LDS as
//   dst[c*CS + ti*PP + py*PWrow + colmap(px)],  colmap = identity (S!=2) or parity-split (S==2)
// with zeros for padding / out-of-range channels, and `act` applied on the way (virtual activations).
struct PatchGeom {
  int PW, PH, PWrow, PWh, PP, CS, S, TIPH;
  FastDiv dPW, dTIPH, dPH;
  int H, W, N, C, act;
  long long bs;
};
template <int U>
__device__ __forceinline__ void stage_patch_u(const float* __restrict__ src, const PatchGeom& g, int c0, int nch,
                                              int n0, int iyb, int ixb, float* __restrict__ dst, int ltid,
                                              int nthreads) {
  // U independent global loads are issued per thread before any LDS store, so a loader wave keeps U*64 requests
  // in flight: the staging is latency-bound (L2 / HBM round trips), not bandwidth-bound.
  const int total = nch * g.TIPH * g.PW;
  const int HW = g.H * g.W;
  for (int base = 0; base < total; base += nthreads * U) {
    float v[U];
    int la[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = base + u * nthreads + ltid;
      v[u] = 0.0f;
      la[u] = -1;
      if (e < total) {
        const uint32_t r = fdiv((uint32_t)e, g.dPW);
        const int px = e - (int)r * g.PW;
        const uint32_t c8 = fdiv(r, g.dTIPH);
        const uint32_t rem = r - c8 * (uint32_t)g.TIPH;
        const uint32_t ti = fdiv(rem, g.dPH);
        const int py = (int)(rem - ti * (uint32_t)g.PH);
        const int c = c0 + (int)c8, n = n0 + (int)ti, iy = iyb + py, ix = ixb + px;
        if (c < g.C && n < g.N && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W)
          v[u] = src[(long long)n * g.bs + (long long)c * HW + iy * g.W + ix];
        const int col = (g.S == 2) ? ((px & 1) * g.PWh + (px >> 1)) : px;
        la[u] = (int)c8 * g.CS + (int)ti * g.PP + py * g.PWrow + col;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (la[u] >= 0) dst[la[u]] = apply_act(v[u], g.act);
  }
}
__device__ __forceinline__ void stage_patch(const float* __restrict__ src, const PatchGeom& g, int c0, int nch,
                                            int n0, int iyb, int ixb, float* __restrict__ dst, int ltid,
                                            int nthreads) {
  stage_patch_u<8>(src, g, c0, nch, n0, iyb, ixb, dst, ltid, nthreads);
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
