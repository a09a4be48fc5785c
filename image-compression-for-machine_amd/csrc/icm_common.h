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

// Kernels that may use more than the default 64 KB of dynamic LDS get the attribute raised ONCE (first launch of
// that kernel in the process), not on every launch.  Thread-safe; returns false if the runtime refuses.
bool ensure_max_lds(const void* fn);

// erf, branch-free.  libm's erff takes one of two polynomial paths behind a per-lane branch; inside the loaders (one wave
// activating a batch of staged values) that control flow serialises the elements into ~40-instruction dependent
// chains and cost a third of the throughput of every convolution / weight gradient with a virtual-GELU operand.
// Both paths are evaluated here (13 FMAs + one v_exp_f32) and selected, so straight-line code of several elements
// interleaves.  Coefficients: the classic single-precision pair (|a| <= 0.927734375: odd polynomial in a; beyond:
// 1 - exp(-t (1 + P(t)))); measured against double-precision erf over [-6, 6]: max abs error 5.8e-8, max rel 8.8e-8.
__device__ __forceinline__ float erf_bf(float a) {
  const float t = fabsf(a), s = a * a;
  float r = fmaf(-1.72853470e-5f, t, 3.83197126e-4f);
  const float u = fmaf(-3.88396438e-3f, t, 2.42546219e-2f);
  r = fmaf(r, s, u);
  r = fmaf(r, t, -1.06777877e-1f);
  r = fmaf(r, t, -6.34846687e-1f);
  r = fmaf(r, t, -1.28717512e-1f);
  r = fmaf(r, t, -t);
  const float big = copysignf(1.0f - __builtin_amdgcn_exp2f(r * 1.4426950408889634f), a);
  float q = -5.96761703e-4f;
  q = fmaf(q, s, 4.99119423e-3f);
  q = fmaf(q, s, -2.67681349e-2f);
  q = fmaf(q, s, 1.12819925e-1f);
  q = fmaf(q, s, -3.76125336e-1f);
  q = fmaf(q, s, 1.28379166e-1f);
  const float small = fmaf(q, a, a);
  return t > 0.927734375f ? big : small;
}
__device__ __forceinline__ float gelu_f(float x) {
  return 0.5f * x * (1.0f + erf_bf(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float dgelu_f(float x) {
  const float cdf = 0.5f * (1.0f + erf_bf(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170f * x * x);   // exp(-x^2 / 2)
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
  int vec4;   // halo-free, 4-pixel-aligned patch: stage with 16-byte loads / LDS stores (host-checked)
  int pipe;   // weight-gradient loaders: two register sets, next batch in flight while this one is stored
  int dma;    // stage by LDS-DMA (no activation, linear patch layout; host-checked): stage_planes_dma
  int v4R, v4P2;   // vec4 with a small plane: one load instruction covers v4R channels, v4P2 lanes each (set_v4_pack)
  int seg_len, seg_gap;   // blocked channel map (0 = none): logical channel c -> plane c + (c / seg_len) * seg_gap
  FastDiv dseg;
};
// physical channel plane of logical channel c (wave-uniform in every caller)
__device__ __forceinline__ int phys_ch(const PatchGeom& g, int c) {
  return g.seg_len ? c + (int)fdiv((uint32_t)c, g.dseg) * g.seg_gap : c;
}
// vec4 staging of small planes (1x1 convolutions on 64..128-pixel tiles fill only 16..32 of the 64 lanes with one
// channel): pack R = 64 / P2 channels into every load instruction, P2 = lanes per channel (power of two >= plane4)
inline void set_v4_pack(PatchGeom& g) {
  g.v4R = 1;
  g.v4P2 = 64;
  if (!g.vec4) return;
  const int plane4 = g.TIPH * (g.PW >> 2);
  if (plane4 > 32) return;
  int p2 = 1;
  while (p2 < plane4) p2 <<= 1;
  g.v4P2 = p2;
  g.v4R = 64 / p2;
}
// Plane-sweep staging: a loader wave owns whole channels; its 64 lanes sweep the (TI x PH x PW) plane of the
// patch linearly.  The per-lane plane offsets (global and LDS) depend only on the tile, so they are computed
// once (PlaneMap) and every channel afterwards costs one address add, one load and one LDS store per element,
// with up to MAXJ loads in flight per lane.
#define ICM_MAXJ 12
struct PlaneMap {
  int goff[ICM_MAXJ];   // offset inside the channel plane (incl. image offset n*bs), -1 = zero fill
  int loff[ICM_MAXJ];   // offset inside the LDS channel slab, -1 = beyond the plane
  int lc;               // vec4 channel packing: this lane's channel inside a group of v4R channels (else 0)
};
__device__ __forceinline__ void plane_map_init(PlaneMap& m, const PatchGeom& g, int n0, int iyb, int ixb, int lane) {
  const int plane = g.TIPH * g.PW;
  m.lc = 0;
#pragma unroll
  for (int j = 0; j < ICM_MAXJ; ++j) {
    const int e = lane + 64 * j;
    m.goff[j] = -1;
    m.loff[j] = -1;
    if (e < plane) {
      const uint32_t r = fdiv((uint32_t)e, g.dPW);
      const int px = e - (int)r * g.PW;
      const uint32_t ti = fdiv(r, g.dPH);
      const int py = (int)(r - ti * (uint32_t)g.PH);
      const int n = n0 + (int)ti, iy = iyb + py, ix = ixb + px;
      if (n < g.N && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W)
        m.goff[j] = ((int)((long long)n * g.bs) + iy * g.W + ix) * 4;   // byte offset (host checks N*bs*4 < 2^31)
      const int col = (g.S == 2) ? ((px & 1) * g.PWh + (px >> 1)) : px;
      m.loff[j] = ((int)ti * g.PP + py * g.PWrow + col) * 4;
    }
  }
}
// stage local channels cl = lw, lw+4, ... < nch (global channel c0 + cl) of the patch into dst.
// NJR = plane slots per lane rounded up to a divisor of 12; 12/NJR channels are in flight together so that a
// lane always has 12 independent loads outstanding, whatever the plane size.
// ---- 16-byte variant for halo-free patches (1x1 convolutions / Linear layers): PW, W, the patch origin and
// all strides are multiples of 4 pixels, so a lane moves 4 consecutive pixels per load and per LDS store.
__device__ __forceinline__ void plane_map_init_v4(PlaneMap& m, const PatchGeom& g, int n0, int iyb, int ixb, int lane) {
  const int pw4 = g.PW >> 2;
  const int plane4 = g.TIPH * pw4;
  const bool packed = g.v4R > 1;
  m.lc = packed ? lane / g.v4P2 : 0;
  const int lane_e = packed ? lane - m.lc * g.v4P2 : lane;
  const int HWb = g.H * g.W * 4;
#pragma unroll
  for (int j = 0; j < ICM_MAXJ; ++j) {
    const int e = lane_e + 64 * j;
    m.goff[j] = -1;
    m.loff[j] = -1;
    if (e < plane4 && !(packed && j > 0)) {
      const int r = e / pw4;
      const int px = (e - r * pw4) << 2;
      const int ti = r / g.PH;
      const int py = r - ti * g.PH;
      const int n = n0 + ti, iy = iyb + py, ix = ixb + px;
      if (n < g.N && (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W)
        m.goff[j] = ((int)((long long)n * g.bs) + iy * g.W + ix) * 4 + m.lc * HWb;
      m.loff[j] = (ti * g.PP + py * g.PWrow + px) * 4 + m.lc * g.CS * 4;
    }
  }
}
// stage local channels of the patch: loader wave lw takes channel GROUPS lw, lw+4, ... (a group = v4R consecutive
// channels moved by one load instruction; v4R = 1 unless the plane is small)
template <int NJR, int MJ, bool PIPE>
__device__ __forceinline__ void stage_planes_t_v4(const float* __restrict__ src, const PlaneMap& m,
                                                  const PatchGeom& g, int c0, int nch, float* __restrict__ dst,
                                                  int lw) {
  constexpr int CPB = MJ / NJR;
  const long long HWb = (long long)g.H * g.W * 4;
  const int R = g.v4R;
  const int ngroups = (nch + R - 1) / R;
  const int nk = (ngroups - lw + 3) >> 2;
  const char* srcb = reinterpret_cast<const char*>(src);
  char* dstb = reinterpret_cast<char*>(dst);
  auto load = [&](f32x4 (&v)[CPB * NJR], int kb) {
#pragma unroll
    for (int u = 0; u < CPB * NJR; ++u) {
      const int k = kb + u / NJR, j = u % NJR;
      const int cl = (lw + 4 * k) * R;            // first local channel of the group (wave-uniform)
      const char* base = srcb + (long long)phys_ch(g, c0 + cl) * HWb;   // (a v4R group never straddles a segment: host-checked)
      if (k < nk && cl + m.lc < nch && c0 + cl + m.lc < g.C && m.goff[j] >= 0)
        v[u] = *reinterpret_cast<const f32x4*>(base + m.goff[j]);
      else v[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
  };
  auto store = [&](const f32x4 (&v)[CPB * NJR], int kb) {
#pragma unroll
    for (int u = 0; u < CPB * NJR; ++u) {
      const int k = kb + u / NJR, j = u % NJR;
      const int cl = (lw + 4 * k) * R;
      if (k < nk && cl + m.lc < nch && m.loff[j] >= 0) {
        f32x4 w;
        w[0] = apply_act(v[u][0], g.act); w[1] = apply_act(v[u][1], g.act);
        w[2] = apply_act(v[u][2], g.act); w[3] = apply_act(v[u][3], g.act);
        char* o = dstb + cl * g.CS * 4 + m.loff[j];
        if ((g.CS & 3) == 0) {
          *reinterpret_cast<f32x4*>(o) = w;
        } else {   // odd channel stride (conflict-free wgrad B fragments): four dword stores
          float* of = reinterpret_cast<float*>(o);
          of[0] = w[0]; of[1] = w[1]; of[2] = w[2]; of[3] = w[3];
        }
      }
    }
  };
  if constexpr (PIPE) {
    // two register sets: batch b + 1 is in flight while batch b is activated and stored (see stage_planes_t)
    f32x4 va[CPB * NJR], vb[CPB * NJR];
    load(va, 0);
    for (int kb = 0; kb < nk; kb += 2 * CPB) {
      if (kb + CPB < nk) load(vb, kb + CPB);
      store(va, kb);
      if (kb + 2 * CPB < nk) load(va, kb + 2 * CPB);
      if (kb + CPB < nk) store(vb, kb + CPB);
    }
  } else {
    for (int kb = 0; kb < nk; kb += CPB) {
      f32x4 v[CPB * NJR];
      load(v, kb);
      store(v, kb);
    }
  }
}
template <int MJ, bool PIPE = false>
__device__ __forceinline__ void stage_planes_v4(const float* __restrict__ src, const PlaneMap& m, const PatchGeom& g,
                                                int c0, int nch, float* __restrict__ dst, int lw) {
  const int nj = (g.TIPH * (g.PW >> 2) + 63) >> 6;
  if (nj <= 1) stage_planes_t_v4<1, MJ, PIPE>(src, m, g, c0, nch, dst, lw);
  else if (nj == 2) stage_planes_t_v4<2, MJ, PIPE>(src, m, g, c0, nch, dst, lw);
  else stage_planes_t_v4<4, MJ, PIPE>(src, m, g, c0, nch, dst, lw);   // host guarantees nj <= 4 when vec4 is set
}

template <int NJR, int MJ, bool PIPE>
__device__ __forceinline__ void stage_planes_t(const float* __restrict__ src, const PlaneMap& m, const PatchGeom& g,
                                               int c0, int nch, float* __restrict__ dst, int lw) {
  // lw must be wave-uniform (readfirstlane): channel bases then live in SGPRs and every load is
  // global_load_dword v, v_off32, s[base] -- no 64-bit vector address arithmetic in the loader.
  // MJ = loads in flight per lane (the loaders are latency-bound: L2 / HBM round trips per batch).
  // Two register sets: the loads of batch b + 1 are issued BEFORE the values of batch b are activated and stored, so the
  // activation (35 VALU instructions per GELU) and the LDS stores run inside the next batch's memory latency instead
  // of after it (s_waitcnt vmcnt(N) retires the batches in issue order).
  constexpr int CPB = MJ / NJR;
  const long long HWb = (long long)g.H * g.W * 4;
  const int nk = (nch - lw + 3) >> 2;   // channels of this loader wave
  const char* srcb = reinterpret_cast<const char*>(src);
  char* dstb = reinterpret_cast<char*>(dst);
  auto load = [&](float (&v)[CPB * NJR], int kb) {
#pragma unroll
    for (int u = 0; u < CPB * NJR; ++u) {
      const int k = kb + u / NJR, j = u % NJR;
      const int c = c0 + lw + 4 * k;
      const char* base = srcb + (long long)phys_ch(g, c) * HWb;
      v[u] = (k < nk && c < g.C && m.goff[j] >= 0) ? *reinterpret_cast<const float*>(base + m.goff[j]) : 0.0f;
    }
  };
  auto store = [&](const float (&v)[CPB * NJR], int kb) {
#pragma unroll
    for (int u = 0; u < CPB * NJR; ++u) {
      const int k = kb + u / NJR, j = u % NJR;
      if (k < nk && m.loff[j] >= 0)
        *reinterpret_cast<float*>(dstb + (lw + 4 * k) * g.CS * 4 + m.loff[j]) = apply_act(v[u], g.act);
    }
  };
  if constexpr (PIPE) {   // weight-gradient kernels (256-register budget); the conv kernels are held to 128 VGPRs
    float va[CPB * NJR], vb[CPB * NJR];
    load(va, 0);
    for (int kb = 0; kb < nk; kb += 2 * CPB) {
      if (kb + CPB < nk) load(vb, kb + CPB);
      store(va, kb);
      if (kb + 2 * CPB < nk) load(va, kb + 2 * CPB);
      if (kb + CPB < nk) store(vb, kb + CPB);
    }
  } else {
    for (int kb = 0; kb < nk; kb += CPB) {
      float v[CPB * NJR];
      load(v, kb);
      store(v, kb);
    }
  }
}
template <int MJ, bool PIPE = false>
__device__ __forceinline__ void stage_planes(const float* __restrict__ src, const PlaneMap& m, const PatchGeom& g,
                                             int c0, int nch, float* __restrict__ dst, int lw) {
  const int nj = (g.TIPH * g.PW + 63) >> 6;
  if (nj <= 1) stage_planes_t<1, MJ, PIPE>(src, m, g, c0, nch, dst, lw);
  else if (nj == 2) stage_planes_t<2, MJ, PIPE>(src, m, g, c0, nch, dst, lw);
  else if (nj == 3) stage_planes_t<3, MJ, PIPE>(src, m, g, c0, nch, dst, lw);
  else if (nj == 4) stage_planes_t<4, MJ, PIPE>(src, m, g, c0, nch, dst, lw);
  else if (nj <= 6) stage_planes_t<6, MJ, PIPE>(src, m, g, c0, nch, dst, lw);
  else if (nj <= 8) stage_planes_t<8, 16, PIPE>(src, m, g, c0, nch, dst, lw);   // two channels (16 loads) in flight
  else stage_planes_t<12, (MJ < 12 ? 12 : MJ), PIPE>(src, m, g, c0, nch, dst, lw);
}

// ---- LDS-DMA staging (global_load_lds_dword: memory -> LDS without a VGPR round trip) -------------------------------
// The register path keeps 12-16 loads per lane in flight and is latency-bound (measured: staging alone takes as long
// as the MFMAs it feeds); a DMA load occupies no destination register, so a loader wave can have up to 64 of them
// outstanding (the vmcnt counter's width).  Usable when the operand needs no activation on the way (the DMA cannot
// apply one) and the LDS patch layout is linear in the lane sweep (identity column map, PWrow == PW): element
// e = lane + 64 j of a channel plane lands at slab + e, exactly where a wave-wide DMA puts lane `lane` of its j-th
// instruction.  Padding / out-of-range elements read from a zero page; lanes beyond the plane are masked off (EXEC).
// The issuing wave must drain vmcnt before the barrier that publishes the stage (barriers do not wait for DMAs).
static __device__ __attribute__((aligned(16))) float icm_zero_page[256];   // zero-initialised when the code object is loaded
                                                                            // (one copy per TU; 16 bytes per lane for the wide DMAs)

template <int NJR>
__device__ __forceinline__ void stage_planes_dma_t(const float* __restrict__ src, const PlaneMap& m, const PatchGeom& g,
                                                   int c0, int nch, float* __restrict__ dst, int lw, int lane) {
  const long long HWb = (long long)g.H * g.W * 4;
  const int nk = (nch - lw + 3) >> 2;   // channels of this loader wave
  const char* srcb = reinterpret_cast<const char*>(src);
  const float* zero = icm_zero_page + lane;
  for (int k = 0; k < nk; ++k) {
    const int cl = lw + 4 * k, c = c0 + cl;
    const char* base = srcb + (long long)phys_ch(g, c) * HWb;
    float* slab = dst + cl * g.CS;
#pragma unroll
    for (int j = 0; j < NJR; ++j) {
      if (m.loff[j] >= 0) {
        const float* p = (c < g.C && m.goff[j] >= 0) ? reinterpret_cast<const float*>(base + m.goff[j]) : zero;
        __builtin_amdgcn_global_load_lds(p, slab + 64 * j, 4, 0, 0);
      }
    }
  }
}
// in-place activation of the channel slabs THIS loader wave filled by DMA (call after its s_waitcnt vmcnt(0): a wave's
// own landed DMAs are visible to its LDS reads); padding zeros stay zero under GELU and x^2
__device__ __forceinline__ void act_planes_inplace(const PatchGeom& g, int nch, float* __restrict__ dst, int lw, int lane) {
  const int plane = g.TIPH * g.PW;
  const int nk = (nch - lw + 3) >> 2;
  for (int k = 0; k < nk; ++k) {
    float* slab = dst + (lw + 4 * k) * g.CS;
    for (int e0 = 0; e0 < plane; e0 += 256) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + 64 * u + lane;
        v[u] = e < plane ? slab[e] : 0.0f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + 64 * u + lane;
        if (e < plane) slab[e] = apply_act(v[u], g.act);
      }
    }
  }
}

__device__ __forceinline__ void stage_planes_dma(const float* __restrict__ src, const PlaneMap& m, const PatchGeom& g,
                                                 int c0, int nch, float* __restrict__ dst, int lw, int lane) {
  const int nj = (g.TIPH * g.PW + 63) >> 6;
  if (nj <= 1) stage_planes_dma_t<1>(src, m, g, c0, nch, dst, lw, lane);
  else if (nj == 2) stage_planes_dma_t<2>(src, m, g, c0, nch, dst, lw, lane);
  else if (nj == 3) stage_planes_dma_t<3>(src, m, g, c0, nch, dst, lw, lane);
  else if (nj == 4) stage_planes_dma_t<4>(src, m, g, c0, nch, dst, lw, lane);
  else if (nj <= 6) stage_planes_dma_t<6>(src, m, g, c0, nch, dst, lw, lane);
  else if (nj <= 8) stage_planes_dma_t<8>(src, m, g, c0, nch, dst, lw, lane);
  else stage_planes_dma_t<12>(src, m, g, c0, nch, dst, lw, lane);
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
