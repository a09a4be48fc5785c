// HBM-bound pointwise / reduction kernels of the WACNN hot path (gfx950).
// All are grid-stride, 16-B vectorised where the layout allows, one pass over their operands.
#include <algorithm>
#include "icm_common.h"

#include <mutex>
#include <unordered_set>

namespace icm {

bool ensure_max_lds(const void* fn) {
  static std::mutex mu;
  static std::unordered_set<const void*> done;
  std::lock_guard<std::mutex> lk(mu);
  if (done.count(fn)) return true;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  done.insert(fn);
  return true;
}

static inline int grid_for(long long n, int per_thread = 4) {
  long long b = (n + 256LL * per_thread - 1) / (256LL * per_thread);
  return (int)std::max<long long>(1, std::min<long long>(b, 256 * 8));
}

#define GRID_STRIDE(i, n) \
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long long)gridDim.x * blockDim.x)

// ---------------------------------------------------------------- deterministic block reduction
// No float atomics anywhere in the library: every rank of a data-parallel job must derive bit-identical sums from
// bit-identical inputs (clip coefficient, bias / LayerNorm / table gradients), whatever the workgroup arrival order.
// Fixed tree: lanes by wave_sum (xor butterfly), waves in index order.
__device__ __forceinline__ float block_sum256(float v, float* red /* >= 4 floats of LDS */) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// ---------------------------------------------------------------- channel sum: out[c] = sum_{n,p} x[n,c,p]
// grid (C, S): workgroup (c, s) reduces pixel chunk s of channel c (strided assignment: fixed for a given shape).
// S == 1: the workgroup writes out[c] itself; S > 1: partials go to ws[s][c] and channel_sum_finish adds them in
// split order.  float4 loads when the plane allows.
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ x, long long bs, int N, int C,
                                                          int HW, float* __restrict__ out, float* __restrict__ ws,
                                                          int accum) {
  __shared__ float red[4];
  const int c = blockIdx.x, S = gridDim.y, sp = blockIdx.y;
  float s = 0.0f;
  if ((HW & 3) == 0 && (bs & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    const int HW4 = HW >> 2;
    const long long total = (long long)N * HW4;
    for (long long i = (long long)sp * 256 + threadIdx.x; i < total; i += (long long)S * 256) {
      const int n = (int)(i / HW4), p = (int)(i - (long long)n * HW4);
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + n * bs + (long long)c * HW + 4 * p);
      s += (v[0] + v[1]) + (v[2] + v[3]);
    }
  } else {
    const long long total = (long long)N * HW;
    for (long long i = (long long)sp * 256 + threadIdx.x; i < total; i += (long long)S * 256) {
      const int n = (int)(i / HW), p = (int)(i - (long long)n * HW);
      s += x[n * bs + (long long)c * HW + p];
    }
  }
  s = block_sum256(s, red);
  if (threadIdx.x == 0) {
    if (S == 1) out[c] = accum ? out[c] + s : s;
    else ws[(long long)sp * C + c] = s;
  }
}
// out[k*C + c] (+)= sum_s ws[(k*S + s)*C + c]   (K independent quantities, S partials each, added in split order)
__global__ void split_sum_finish_kernel(const float* __restrict__ ws, float* __restrict__ out0,
                                        float* __restrict__ out1, int C, int S, int K, int accum) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C * K) return;
  const int k = i / C, c = i - k * C;
  float s = 0.0f;
  for (int j = 0; j < S; ++j) s += ws[((long long)k * S + j) * C + c];
  float* o = (k == 0 ? out0 : out1) + c;
  *o = accum ? *o + s : s;
}

// the same sum for MANY partials (S up to ~1000 rows: one per workgroup of the fused LayerNorm backward): one wave per
// output, lane l adds rows l, l + 64, ... in order, then a fixed xor tree over the lanes -- deterministic, and 64 loads
// in flight per output instead of one serial chain of S dependent additions (which cost 0.1 ms per launch at S = 1024)
__global__ __launch_bounds__(256) void split_sum_finish_wave_kernel(const float* __restrict__ ws, float* __restrict__ out0,
                                                                    float* __restrict__ out1, int C, int S, int K,
                                                                    int accum) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= C * K) return;   // wave-uniform
  const int k = i / C, c = i - k * C;
  float s = 0.0f;
  for (int j = lane; j < S; j += 64) s += ws[((long long)k * S + j) * C + c];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) {
    float* o = (k == 0 ? out0 : out1) + c;
    *o = accum ? *o + s : s;
  }
}

// ---------------------------------------------------------------- NonNegativeParametrizer
__global__ void nonneg_fwd_kernel(const float* p, float* out, long long n, float bound, float ped) {
  GRID_STRIDE(i, n) {
    const float v = fmaxf(p[i], bound);
    out[i] = v * v - ped;
  }
}
__global__ void nonneg_bwd_kernel(const float* p, const float* g, float* dp, long long n, float bound, int accum) {
  GRID_STRIDE(i, n) {
    const float pv = p[i];
    const float gg = g[i] * 2.0f * fmaxf(pv, bound);
    float r = (pv >= bound || gg < 0.0f) ? gg : 0.0f;
    if (accum) r += dp[i];
    dp[i] = r;
  }
}

// ---------------------------------------------------------------- GDN backward pre-pass
__global__ void gdn_bwd_pre_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                   const float* __restrict__ nrm, float* __restrict__ dn, float* __restrict__ t1,
                                   long long n, int inverse) {
  GRID_STRIDE(i, n) {
    const float nv = nrm[i], gv = g[i], xv = x[i];
    if (inverse) {
      const float s = sqrtf(nv);
      t1[i] = gv * s;
      dn[i] = 0.5f * gv * xv / s;
    } else {
      const float r = rsqrtf(nv);
      t1[i] = gv * r;
      dn[i] = -0.5f * gv * xv * r / nv;
    }
  }
}

// ---------------------------------------------------------------- GELU / gate / grad plumbing
__global__ void gelu_fwd_kernel(const float* x, float* y, long long n) {
  GRID_STRIDE(i, n) y[i] = gelu_f(x[i]);
}
__global__ void gate_fwd_kernel(const float* a_pre, const float* b, const float* x, float* out, long long n) {
  GRID_STRIDE(i, n) out[i] = gelu_f(a_pre[i]) * sigmoid_f(b[i]) + x[i];
}
__global__ void gate_bwd_kernel(const float* g, const float* a_pre, const float* b, float* da, float* db, float* dx,
                                long long n, int accum_da, int accum_dx) {
  GRID_STRIDE(i, n) {
    const float gv = g[i], ap = a_pre[i], s = sigmoid_f(b[i]);
    float va = gv * s * dgelu_f(ap);
    if (accum_da) va += da[i];
    da[i] = va;
    db[i] = gv * gelu_f(ap) * s * (1.0f - s);
    float vx = gv;
    if (accum_dx) vx += dx[i];
    dx[i] = vx;
  }
}
__global__ void add_grad_kernel(const float* src, const float* pre, float* dst, long long n, int accum) {
  GRID_STRIDE(i, n) {
    float v = src[i];
    if (pre) v *= dgelu_f(pre[i]);
    if (accum) v += dst[i];
    dst[i] = v;
  }
}
__global__ void ste_round_offset_kernel(const float* z, const float* quant, float* zh, int N, int C, int HW) {
  const long long n = (long long)N * C * HW;
  GRID_STRIDE(i, n) {
    const int c = (int)((i / HW) % C);
    const float med = quant[c * 3 + 1];
    const float t = z[i] - med;
    zh[i] = ((rintf(t) - t) + t) + med;
  }
}
#define ICM_GATHER_MAX 64
struct GatherPtrs {
  const float* p[ICM_GATHER_MAX];
};
// dst[i * len + e] = srcs[i][e]: the first-layer biases of all slice chains as one vector (cnn.py:89-127)
__global__ __launch_bounds__(256) void gather_vectors_kernel(const GatherPtrs P, int len, float* __restrict__ dst) {
  const float* src = P.p[blockIdx.x];
  for (int e = threadIdx.x; e < len; e += 256) dst[(long long)blockIdx.x * len + e] = src[e];
}

__global__ void copy_strided_kernel(const float* src, long long sbs, float* dst, long long dbs, int N, int C, int HW,
                                    int accum) {
  const long long per = (long long)C * HW, n = (long long)N * per;
  GRID_STRIDE(i, n) {
    const long long b = i / per, r = i - b * per;
    float v = src[b * sbs + r];
    if (accum) v += dst[b * dbs + r];
    dst[b * dbs + r] = v;
  }
}
// dst[b][a][k] (+)= src[a][b][K-1-k]: a Conv2d weight [Cout][Cin][KH*KW] as the ConvTranspose2d weight of the same map
// (stride 1: conv(x, W, pad p) = convT(x, W', pad K-1-p), W'[ci][co][ky][kx] = W[co][ci][K-1-ky][K-1-kx]), and the
// gradient of W' back into the layout of W (the same permutation with A and B exchanged)
__global__ void permute_flip_kernel(const float* src, float* dst, int A, int B, int K, int accum) {
  const long long n = (long long)A * B * K;
  GRID_STRIDE(i, n) {
    const int k = (int)(i % K);
    const long long ab = i / K;
    const int a = (int)(ab % A), b = (int)(ab / A);
    float v = src[((long long)a * B + b) * K + (K - 1 - k)];
    if (accum) v += dst[i];
    dst[i] = v;
  }
}
// dpre[n,c,p] = g[n,c,p] * 0.5 * (1 - t^2)   (LRP tail backward, cnn.py:177-178)
__global__ void lrp_bwd_kernel(const float* g, long long gbs, const float* t, long long tbs, float* dpre, long long dbs,
                               int N, int C, int HW) {
  const long long per = (long long)C * HW, n = (long long)N * per;
  GRID_STRIDE(i, n) {
    const long long b = i / per, r = i - b * per;
    const float tv = t[b * tbs + r];
    dpre[b * dbs + r] = g[b * gbs + r] * 0.5f * (1.0f - tv * tv);
  }
}
// dst[n][c*4 + dy*2 + dx][y][x] = src[n][c][2y+dy][2x+dx]   (inverse of nn.PixelShuffle(2))
__global__ void pixel_unshuffle2_kernel(const float* src, float* dst, int N, int C, int H, int W) {
  const long long n = (long long)N * C * 4 * H * W;
  GRID_STRIDE(i, n) {
    const int x = (int)(i % W);
    long long q = i / W;
    const int y = (int)(q % H); q /= H;
    const int c4 = (int)(q % (C * 4));
    const int b = (int)(q / (C * 4));
    const int c = c4 >> 2, dy = (c4 >> 1) & 1, dx = c4 & 1;
    dst[i] = src[(((long long)b * C + c) * (2 * H) + 2 * y + dy) * (2 * W) + 2 * x + dx];
  }
}
// ---------------------------------------------------------------- LayerNorm over channels on NCHW (stf.py: nn.LayerNorm(C) on tokens)
// workgroup = 64 pixels x 4 channel slices: lane = pixel (every load/store of a wave is one 256-B row of a channel
// plane), wave = channel slice c = w, w+4, ...; the per-pixel channel sums of the four slices meet in LDS.
// Two-pass statistics (mean, then centred variance) like the reference's nn.LayerNorm.
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, long long xbs,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            long long ybs, float* __restrict__ mean,
                                                            float* __restrict__ rstd, int N, int C, int HW, float eps) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, cs = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long total = (long long)N * HW;
  const long long ntiles = (total + 63) / 64;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long i = tile * 64 + lane;
    const bool valid = i < total;
    const int n = valid ? (int)(i / HW) : 0, p = valid ? (int)(i - (long long)n * HW) : 0;
    const float* xp = x + n * xbs + p;
    float s = 0.0f;
    if (valid)
      for (int c = cs; c < C; c += 4) s += xp[(long long)c * HW];
    red[cs][lane] = s;
    __syncthreads();
    const float m = ((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane])) / (float)C;
    __syncthreads();
    float v = 0.0f;
    if (valid)
      for (int c = cs; c < C; c += 4) {
        const float d = xp[(long long)c * HW] - m;
        v += d * d;
      }
    red[cs][lane] = v;
    __syncthreads();
    const float r = rsqrtf(((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane])) / (float)C + eps);
    __syncthreads();
    if (valid) {
      if (cs == 0 && mean) { mean[i] = m; rstd[i] = r; }
      float* yp = y + n * ybs + p;
      for (int c = cs; c < C; c += 4) yp[(long long)c * HW] = (xp[(long long)c * HW] - m) * r * gamma[c] + beta[c];
    }
  }
}
// dx = rstd * (g*gamma - mean_c(g*gamma) - xhat * mean_c(g*gamma*xhat))
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, long long xbs,
                                                            const float* __restrict__ dy, long long dbs,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ dx,
                                                            long long dxbs, int N, int C, int HW, int accum,
                                                            const float* __restrict__ extra, long long ebs) {
  __shared__ float red[2][4][64];
  const int lane = threadIdx.x & 63, cs = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long total = (long long)N * HW;
  const long long ntiles = (total + 63) / 64;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long i = tile * 64 + lane;
    const bool valid = i < total;
    const int n = valid ? (int)(i / HW) : 0, p = valid ? (int)(i - (long long)n * HW) : 0;
    const float* xp = x + n * xbs + p;
    const float* gp = dy + n * dbs + p;
    const float m = valid ? mean[i] : 0.0f, r = valid ? rstd[i] : 0.0f;
    float s1 = 0.0f, s2 = 0.0f;
    if (valid)
      for (int c = cs; c < C; c += 4) {
        const float gg = gp[(long long)c * HW] * gamma[c];
        s1 += gg;
        s2 += gg * (xp[(long long)c * HW] - m) * r;
      }
    red[0][cs][lane] = s1;
    red[1][cs][lane] = s2;
    __syncthreads();
    s1 = ((red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane])) / (float)C;
    s2 = ((red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane])) / (float)C;
    __syncthreads();
    if (valid) {
      float* dp = dx + n * dxbs + p;
      const float* ep = extra ? extra + n * ebs + p : nullptr;
      for (int c = cs; c < C; c += 4) {
        const float xh = (xp[(long long)c * HW] - m) * r;
        float v = r * (gp[(long long)c * HW] * gamma[c] - s1 - xh * s2);
        if (ep) v += ep[(long long)c * HW];   // identity-path gradient of the residual add around this LayerNorm
        if (accum) v += dp[(long long)c * HW];
        dp[(long long)c * HW] = v;
      }
    }
  }
}
// Register-cached variants for C = NW * CPT in {48, 96, 192, 384} (all stf levels and patch_embed): NW waves per
// workgroup, wave cs holds channels cs, cs + NW, ... of 64 pixels; every thread keeps its CPT channel values of x (and
// dy) in registers, so each tensor is read from HBM once instead of three times.  With FUSE the backward kernel also
// carries the parameter gradients: every lane accumulates dy * xhat and dy of its channels over all tiles of its
// workgroup, one cross-lane sum at the end, partials pws[{gamma, beta}][workgroup][c] for split_sum_finish_kernel
// (fixed order) -- the separate pass over x and dy of layernorm_bwd_params_kernel is gone for C <= 192 (CPT <= 24; at
// C = 384 the 2 x 48 extra accumulators do not fit next to the 96 cached values: that level keeps the separate pass).
template <int VW>
__device__ __forceinline__ float ln_sum_waves(const float (*red)[64], int px) {
  float t[VW];
#pragma unroll
  for (int v = 0; v < VW; ++v) t[v] = red[v][px];
#pragma unroll
  for (int w = 1; w < VW; w *= 2)
#pragma unroll
    for (int v = 0; v + w < VW; v += 2 * w) t[v] += t[v + w];   // fixed pairwise order
  return t[0];
}
template <int CPT, int NW>
__global__ __launch_bounds__(64 * NW) void layernorm_fwd_cached_kernel(const float* __restrict__ x, long long xbs,
                                                                      const float* __restrict__ gamma,
                                                                      const float* __restrict__ beta,
                                                                      float* __restrict__ y, long long ybs,
                                                                      float* __restrict__ mean, float* __restrict__ rstd,
                                                                      int N, int HW, float eps) {
  constexpr int VW = NW, LW = 64, C = VW * CPT;
  __shared__ float red[VW][LW];
  const int lane = threadIdx.x & 63, px = lane;
  const int vs = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long total = (long long)N * HW;
  const long long ntiles = (total + LW - 1) / LW;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long i = tile * LW + px;
    const bool valid = i < total;
    const int n = valid ? (int)(i / HW) : 0, p = valid ? (int)(i - (long long)n * HW) : 0;
    const float* xp = x + n * xbs + p;
    float xv[CPT];
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
      xv[k] = valid ? xp[(long long)(vs + VW * k) * HW] : 0.0f;
      s += xv[k];
    }
    red[vs][px] = s;
    __syncthreads();
    const float m = ln_sum_waves<VW>(red, px) / (float)C;
    __syncthreads();
    float v = 0.0f;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
      const float dd = xv[k] - m;
      v += dd * dd;
    }
    red[vs][px] = v;
    __syncthreads();
    const float r = rsqrtf(ln_sum_waves<VW>(red, px) / (float)C + eps);
    __syncthreads();
    if (valid) {
      if (vs == 0 && mean) { mean[i] = m; rstd[i] = r; }
      float* yp = y + n * ybs + p;
#pragma unroll
      for (int k = 0; k < CPT; ++k) {
        const int c = vs + VW * k;
        yp[(long long)c * HW] = (xv[k] - m) * r * gamma[c] + beta[c];
      }
    }
  }
}
template <int CPT, int NW, bool FUSE>
__global__ __launch_bounds__(64 * NW) void layernorm_bwd_cached_kernel(const float* __restrict__ x, long long xbs,
                                                                      const float* __restrict__ dy, long long dbs,
                                                                      const float* __restrict__ gamma,
                                                                      const float* __restrict__ mean,
                                                                      const float* __restrict__ rstd,
                                                                      float* __restrict__ dx, long long dxbs, int N,
                                                                      int HW, int accum, const float* __restrict__ extra,
                                                                      long long ebs, float* __restrict__ pws) {
  constexpr int VW = NW, LW = 64, C = VW * CPT;
  __shared__ float red[2][VW][LW];
  const int lane = threadIdx.x & 63, px = lane;
  const int vs = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long total = (long long)N * HW;
  const long long ntiles = (total + LW - 1) / LW;
  float ag[FUSE ? CPT : 1], ab[FUSE ? CPT : 1];   // parameter-gradient partials of this lane's pixels
#pragma unroll
  for (int k = 0; k < (FUSE ? CPT : 1); ++k) ag[k] = ab[k] = 0.0f;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long long i = tile * LW + px;
    const bool valid = i < total;
    const int n = valid ? (int)(i / HW) : 0, p = valid ? (int)(i - (long long)n * HW) : 0;
    const float* xp = x + n * xbs + p;
    const float* gp = dy + n * dbs + p;
    const float m = valid ? mean[i] : 0.0f, r = valid ? rstd[i] : 0.0f;
    float xh[CPT], gg[CPT];
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
      const int c = vs + VW * k;
      xh[k] = valid ? (xp[(long long)c * HW] - m) * r : 0.0f;
      const float graw = valid ? gp[(long long)c * HW] : 0.0f;
      if constexpr (FUSE) {
        ag[k] += graw * xh[k];
        ab[k] += graw;
      }
      gg[k] = graw * gamma[c];
      s1 += gg[k];
      s2 += gg[k] * xh[k];
    }
    red[0][vs][px] = s1;
    red[1][vs][px] = s2;
    __syncthreads();
    s1 = ln_sum_waves<VW>(red[0], px) / (float)C;
    s2 = ln_sum_waves<VW>(red[1], px) / (float)C;
    __syncthreads();
    if (valid) {
      float* dp = dx + n * dxbs + p;
      const float* ep = extra ? extra + n * ebs + p : nullptr;
#pragma unroll
      for (int k = 0; k < CPT; ++k) {
        const int c = vs + VW * k;
        float v = r * (gg[k] - s1 - xh[k] * s2);
        if (ep) v += ep[(long long)c * HW];
        if (accum) v += dp[(long long)c * HW];
        dp[(long long)c * HW] = v;
      }
    }
  }
  if constexpr (FUSE) {
    const int S = gridDim.x;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
      float a = ag[k], b = ab[k];
#pragma unroll
      for (int o = LW / 2; o >= 1; o >>= 1) {   // within the virtual wave's LW lanes
        a += __shfl_xor(a, o, 64);
        b += __shfl_xor(b, o, 64);
      }
      if (px == 0) {
        const int c = vs + VW * k;
        pws[(long long)blockIdx.x * C + c] = a;
        pws[((long long)S + blockIdx.x) * C + c] = b;
      }
    }
  }
}
// dgamma[c] (+)= sum_{n,p} dy * xhat ; dbeta[c] (+)= sum dy.  grid (C, S): workgroup (c, s) reduces pixel chunk s of
// channel c; S == 1 writes the outputs, S > 1 writes partials ws[{gamma,beta}][s][c] for split_sum_finish_kernel
// (deterministic: no float atomics)
__global__ __launch_bounds__(256) void layernorm_bwd_params_kernel(const float* __restrict__ x, long long xbs,
                                                                   const float* __restrict__ dy, long long dbs,
                                                                   const float* __restrict__ mean,
                                                                   const float* __restrict__ rstd, float* dgamma,
                                                                   float* dbeta, float* __restrict__ ws, int N, int C,
                                                                   int HW, int accum) {
  __shared__ float red[4];
  const int c = blockIdx.x, S = gridDim.y, sp = blockIdx.y;
  float sg = 0.0f, sb = 0.0f;
  const long long total = (long long)N * HW;
  for (long long i = (long long)sp * 256 + threadIdx.x; i < total; i += (long long)S * 256) {
    const int n = (int)(i / HW), p = (int)(i - (long long)n * HW);
    const float g = dy[n * dbs + (long long)c * HW + p];
    const float xh = (x[n * xbs + (long long)c * HW + p] - mean[i]) * rstd[i];
    sg += g * xh;
    sb += g;
  }
  sg = block_sum256(sg, red);
  sb = block_sum256(sb, red);
  if (threadIdx.x == 0) {
    if (S == 1) {
      dgamma[c] = accum ? dgamma[c] + sg : sg;
      dbeta[c] = accum ? dbeta[c] + sb : sb;
    } else {
      ws[(long long)sp * C + c] = sg;
      ws[((long long)S + sp) * C + c] = sb;
    }
  }
}
// PatchMerging gather (stf.py:224-228): dst[n][k*C + c][y][x] = src[n][c][2y + (k&1)][2x + (k>>1)]; inverse = its gradient
__global__ void space_to_depth2_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int C, int H,
                                       int W, int inverse, int accum) {
  const int H2 = H / 2, W2 = W / 2;
  const long long total = (long long)N * 4 * C * H2 * W2;
  GRID_STRIDE(i, total) {
    const int x = (int)(i % W2);
    long long q = i / W2;
    const int y = (int)(q % H2); q /= H2;
    const int kc = (int)(q % (4 * C));
    const int n = (int)(q / (4 * C));
    const int k = kc / C, c = kc - k * C;
    const long long big = (((long long)n * C + c) * H + 2 * y + (k & 1)) * W + 2 * x + (k >> 1);
    if (!inverse) dst[i] = src[big];
    else {
      float v = src[i];
      if (accum) v += dst[big];
      dst[big] = v;
    }
  }
}
// DropPath residual (stf.py:190-191): out[n] = shortcut[n] + scale[n] * branch[n]; backward d_branch = scale[n]*g
__global__ void residual_scale_kernel(const float* __restrict__ shortcut, const float* __restrict__ branch,
                                      const float* __restrict__ scale, float* __restrict__ out, long long per,
                                      long long n) {
  GRID_STRIDE(i, n) {
    const float s = scale[i / per];
    out[i] = (shortcut ? shortcut[i] : 0.0f) + s * branch[i];
  }
}
// ---------------------------------------------------------------- thin-channel convolutions as GEMM + (im2col | col2im)
// The 3-channel ends of the codec (g_a.0: conv 3->192, g_s.8: convT 192->3; cnn.py:32,51) waste 90 % of a 32-row MFMA
// tile in the implicit-GEMM kernel.  Their (channel, tap) pairs become the channel axis of a 1x1 GEMM instead:
//   im2col:  cols[n][c*KK + t][oy][ox] = x[n][c][oy*S - pad + kh][ox*S - pad + kw]        (0 outside)
//   col2im:  out[n][c][y][x] = bias[c] + sum_t cols[n][c*KK + t][(y + pad - kh)/S][(x + pad - kw)/S]   (adjoint map)
// Both are one pass over the 75-channel column tensor (HBM-bound, coalesced along x).
__global__ void im2col_kernel(const float* __restrict__ x, float* __restrict__ cols, int N, int C, int H, int W, int OH,
                              int OW, int K, int S, int pad) {
  const int KK = K * K;
  const long long total = (long long)N * C * KK * OH * OW;
  GRID_STRIDE(i, total) {
    const int ox = (int)(i % OW);
    long long q = i / OW;
    const int oy = (int)(q % OH); q /= OH;
    const int ct = (int)(q % (C * KK));
    const int n = (int)(q / (C * KK));
    const int c = ct / KK, t = ct - c * KK, kh = t / K, kw = t - kh * K;
    const int iy = oy * S - pad + kh, ix = ox * S - pad + kw;
    float v = 0.0f;
    if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = x[(((long long)n * C + c) * H + iy) * W + ix];
    cols[i] = v;
  }
}
__global__ void col2im_kernel(const float* __restrict__ cols, const float* __restrict__ bias, float* __restrict__ out,
                              int N, int C, int H, int W, int OH, int OW, int K, int S, int pad, int accum) {
  const int KK = K * K;
  const long long total = (long long)N * C * H * W;
  GRID_STRIDE(i, total) {
    const int x = (int)(i % W);
    long long q = i / W;
    const int y = (int)(q % H); q /= H;
    const int c = (int)(q % C);
    const int n = (int)(q / C);
    float v = bias ? bias[c] : 0.0f;
    const float* cp = cols + ((long long)n * C + c) * KK * OH * OW;
    for (int kh = 0; kh < K; ++kh) {
      const int ty = y + pad - kh;
      if (ty < 0 || ty % S != 0) continue;
      const int oy = ty / S;
      if (oy >= OH) continue;
      for (int kw = 0; kw < K; ++kw) {
        const int tx = x + pad - kw;
        if (tx < 0 || tx % S != 0) continue;
        const int ox = tx / S;
        if (ox >= OW) continue;
        v += cp[((long long)(kh * K + kw) * OH + oy) * OW + ox];
      }
    }
    if (accum) v += out[i];
    out[i] = v;
  }
}
__global__ void fill_kernel(float* p, long long n, float v) {
  GRID_STRIDE(i, n) p[i] = v;
}
__global__ void clamp_kernel(float* p, long long n, float lo, float hi) {
  GRID_STRIDE(i, n) p[i] = fminf(fmaxf(p[i], lo), hi);
}
// F.pad(x, (left, right, top, bottom), value) with negative pads = crop: dst[n,c,y,x] = src[n,c,y-top,x-left] or value
__global__ void pad2d_kernel(const float* __restrict__ src, int N, int C, int H, int W, float* __restrict__ dst, int OH,
                             int OW, int top, int left, float value) {
  const long long total = (long long)N * C * OH * OW;
  GRID_STRIDE(i, total) {
    const int x = (int)(i % OW);
    long long q = i / OW;
    const int y = (int)(q % OH);
    const long long nc = q / OH;
    const int sy = y - top, sx = x - left;
    dst[i] = ((unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W) ? src[(nc * H + sy) * W + sx] : value;
  }
}

// ---------------------------------------------------------------- zigzag block ordering (stf6.py:654-762)
// The latent [B, C, H, W] is a grid of ns x nH x nW contiguous blocks ([B, ns, C/ns, nH, H/nH, nW, W/nW] view); the
// zigzag tensor [B, N, C/ns, H/nH, W/nW] lists them in shell order (blk[n] = (c * nH + h) * nW + w of output block n).
// Pure index permutation, HBM-bound: one element per thread, both sides read / written in runs of W/nW floats.
struct ZigzagDesc {
  int blk[64];
};
__global__ void zigzag_kernel(const float* __restrict__ src, float* __restrict__ dst, long long x_bs, int B, int N, int Cs,
                              int nH, int Hb, int nW, int Wb, int reverse, const ZigzagDesc d) {
  const long long total = (long long)B * N * Cs * Hb * Wb;
  const int H = nH * Hb, W = nW * Wb;
  GRID_STRIDE(i, total) {
    const int q = (int)(i % Wb);
    long long t = i / Wb;
    const int r = (int)(t % Hb);
    t /= Hb;
    const int cs = (int)(t % Cs);
    t /= Cs;
    const int n = (int)(t % N);
    const long long b = t / N;
    const int id = d.blk[n];
    const int w = id % nW, h = (id / nW) % nH, c = id / (nW * nH);
    const long long xo = b * x_bs + ((long long)(c * Cs + cs) * H + h * Hb + r) * W + w * Wb + q;
    if (reverse) dst[xo] = src[i];
    else dst[i] = src[xo];
  }
}

// ---------------------------------------------------------------- R-D loss
// stage 1: per-workgroup partial sums of (squared error, log lik_y, log lik_z) into ws[3][nblk];
// stage 2 (one workgroup): adds the partials in block order and forms bpp / mse / loss.  Deterministic.
__global__ __launch_bounds__(256) void rd_reduce_kernel(const float* x, const float* xh, long long nx, const float* ly,
                                                        long long ny, const float* lz, long long nz, float* ws) {
  __shared__ float red[4];
  float se = 0.0f, sy = 0.0f, sz = 0.0f;
  GRID_STRIDE(i, nx) {
    const float d = xh[i] - x[i];
    se += d * d;
  }
  GRID_STRIDE(i, ny) sy += logf(ly[i]);
  GRID_STRIDE(i, nz) sz += logf(lz[i]);
  se = block_sum256(se, red);
  sy = block_sum256(sy, red);
  sz = block_sum256(sz, red);
  if (threadIdx.x == 0) {
    ws[blockIdx.x] = se;
    ws[gridDim.x + blockIdx.x] = sy;
    ws[2 * gridDim.x + blockIdx.x] = sz;
  }
}
// sum of nblk partials in fixed order by one 256-thread workgroup (thread t takes t, t+256, ...)
__device__ __forceinline__ float ordered_sum(const float* p, int nblk, float* red) {
  float s = 0.0f;
  for (int i = threadIdx.x; i < nblk; i += 256) s += p[i];
  return block_sum256(s, red);
}
__global__ __launch_bounds__(256) void rd_finish_kernel(const float* ws, int nblk, float* out, long long nx,
                                                        long long npix, float lmbda) {
  __shared__ float red[4];
  const float se = ordered_sum(ws, nblk, red);
  const float sy = ordered_sum(ws + nblk, nblk, red);
  const float sz = ordered_sum(ws + 2 * nblk, nblk, red);
  if (threadIdx.x == 0) {
    const float mse = se / (float)nx;
    const float denom = -0.69314718055994530942f * (float)npix;
    const float bpp = sy / denom + sz / denom;
    out[0] = bpp;
    out[1] = mse;
    out[2] = lmbda * 65025.0f * mse + bpp;
    out[3] = sy;
    out[4] = sz;
  }
}
__global__ void rd_bwd_kernel(const float* x, const float* xh, long long nx, const float* ly, long long ny,
                              const float* lz, long long nz, long long npix, float lmbda, float gscale, float* dxh,
                              float* dly, float* dlz) {
  const float cm = gscale * lmbda * 65025.0f * 2.0f / (float)nx;
  const float cl = gscale / (-0.69314718055994530942f * (float)npix);
  GRID_STRIDE(i, nx) dxh[i] = cm * (xh[i] - x[i]);
  GRID_STRIDE(i, ny) dly[i] = cl / ly[i];
  GRID_STRIDE(i, nz) dlz[i] = cl / lz[i];
}

// ---------------------------------------------------------------- optimiser
__global__ __launch_bounds__(256) void sqnorm_kernel(const float* g, long long n, float* ws) {
  __shared__ float red[4];
  float s = 0.0f;
  const f32x4* g4 = reinterpret_cast<const f32x4*>(g);
  const long long n4 = ((reinterpret_cast<uintptr_t>(g) & 15) == 0) ? n / 4 : 0;
  GRID_STRIDE(i, n4) {
    const f32x4 v = g4[i];
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  for (long long i = n4 * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    s += g[i] * g[i];
  s = block_sum256(s, red);
  if (threadIdx.x == 0) ws[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void sqnorm_finish_kernel(const float* ws, int nblk, float* out) {
  __shared__ float red[4];
  const float s = ordered_sum(ws, nblk, red);
  if (threadIdx.x == 0) out[0] = s;
}
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, float step_size, float b1, float b2, float omb1,
                            float omb2, float eps, float bc2_sqrt, const float* sqnorm, float max_norm, float gscale,
                            const float* __restrict__ hyper) {
  if (hyper) {   // step-dependent scalars from device memory (hipGraph replay: kernel arguments are frozen at capture)
    step_size = hyper[0];
    bc2_sqrt = hyper[1];
  }
  float coef = gscale;
  if (sqnorm) {
    const float total = sqrtf(*sqnorm) * gscale;
    coef *= fminf(1.0f, max_norm / (total + 1e-6f));
  }
  GRID_STRIDE(i, n) {
    const float gv = g[i] * coef;
    const float mv = b1 * m[i] + omb1 * gv;
    const float vv = b2 * v[i] + omb2 * gv * gv;
    m[i] = mv;
    v[i] = vv;
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mv / denom);
  }
}

}  // namespace icm

using namespace icm;
#define ST ((hipStream_t)stream)

extern "C" {

int icm_channel_sum(const float* x, int64_t x_bs, int N, int C, int HW, float* out, int accum, float* ws,
                    int64_t ws_floats, void* stream) {
  if (!x || !out || N <= 0 || C <= 0 || HW <= 0) return ICM_ERR_ARG;
  const long long per_c = (long long)N * HW;
  long long S = std::max<long long>(1, std::min<long long>(32, per_c / 16384));
  if (!ws) S = 1;
  else S = std::max<long long>(1, std::min<long long>(S, ws_floats / C));
  hipLaunchKernelGGL(channel_sum_kernel, dim3(C, (unsigned)S), dim3(256), 0, ST, x, (long long)x_bs, N, C, HW, out, ws,
                     accum);
  ICM_CHECK_LAUNCH();
  if (S > 4) {   // one wave per channel: the serial form is a chain of S dependent L2 round trips (~50 us at S = 32)
    hipLaunchKernelGGL(split_sum_finish_wave_kernel, dim3((C + 3) / 4), dim3(256), 0, ST, ws, out, out, C, (int)S, 1, accum);
    ICM_CHECK_LAUNCH();
  } else if (S > 1) {
    hipLaunchKernelGGL(split_sum_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, ST, ws, out, out, C, (int)S, 1,
                       accum);
    ICM_CHECK_LAUNCH();
  }
  return ICM_OK;
}
int icm_nonneg_fwd(const float* p, float* out, int64_t n, float bound, float pedestal, void* stream) {
  if (!p || !out || n <= 0) return ICM_ERR_ARG;
  hipLaunchKernelGGL(nonneg_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, ST, p, out, (long long)n, bound, pedestal);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}
int icm_nonneg_bwd(const float* p, const float* g_eff, float* dp, int64_t n, float bound, int accum, void* stream) {
  if (!p || !g_eff || !dp || n <= 0) return ICM_ERR_ARG;
  hipLaunchKernelGGL(nonneg_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, ST, p, g_eff, dp, (long long)n, bound, accum);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}
int icm_gdn_bwd_pre(const float* g, const float* x, const float* nrm, float* dn, float* t1, int64_t n, int inverse,
                    void* stream) {
  if (!g || !x || !nrm || !dn || !t1 || n <= 0) return ICM_ERR_ARG;
  hipLaunchKernelGGL(gdn_bwd_pre_kernel, dim3(grid_for(n)), dim3(256), 0, ST, g, x, nrm, dn, t1, (long long)n, inverse);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}
int icm_gelu_fwd(const float* x, float* y, int64_t n, void* stream) {
  if (!x || !y || n <= 0) return ICM_ERR_ARG;
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, ST, x, y, (long long)n);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}
int icm_gate_fwd(const float* a_pre, const float* b, const float* x, float* out, int64_t n, void* stream) {
  if (!a_pre || !b || !x || !out || n <= 0) return ICM_ERR_ARG;
  hipLaunchKernelGGL(gate_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, ST, a_pre, b, x, out, (long long)n);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}
int icm_gate_bwd(const float* g, const float* a_pre, const float* b, float* da_pre, float* db, float* dx, int64_t n,
                 int accum_da, int accum_dx, void* stream) {
  if (!g || !a_pre || !b || !da_pre || !db || !dx || n <= 0) return ICM_ERR_ARG;
  hipLaunchKernelGGL(gate_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, ST, g, a_pre, b, da_pre, db, dx, (long long)n,
                     accum_da, accum_dx);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}
int icm_add_grad(const float* src, const float* mul_dgelu_of, float* dst, int64_t n, int accum, void* stream) {
  if (!src || !dst || n <= 0) return ICM_ERR_ARG;
  hipLaunchKernelGGL(add_grad_kernel, dim3(grid_for(n)), dim3(256), 0, ST, src, mul_dgelu_of, dst, (long long)n, accum);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}
int icm_ste_round_offset(const float* z, const float* quantiles, float* z_hat, int N, int C, int HW, void* stream) {
  if (!z || !quantiles || !z_hat || N <= 0 || C <= 0 || HW <= 0) return ICM_ERR_ARG;
  hipLaunchKernelGGL(ste_round_offset_kernel, dim3(grid_for((long long)N * C * HW)), dim3(256), 0, ST, z, quantiles,
                     z_hat, N, C, HW);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}
int icm_copy_strided(const float* src, int64_t src_bs, float* dst, int64_t dst_bs, int N, int C, int HW, int accum,
                     void* stream) {
  if (!src || !dst || N <= 0 || C <= 0 || HW <= 0) return ICM_ERR_ARG;
  hipLaunchKernelGGL(copy_strided_kernel, dim3(grid_for((long long)N * C * HW)), dim3(256), 0, ST, src,
                     (long long)src_bs, dst, (long long)dst_bs, N, C, HW, accum);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}
int icm_permute_flip(const float* src, float* dst, int A, int B, int K, int accum, void* stream) {
  if (!src || !dst || A <= 0 || B <= 0 || K <= 0 || src == dst) return ICM_ERR_ARG;
  hipLaunchKernelGGL(permute_flip_kernel, dim3(grid_for((long long)A * B * K)), dim3(256), 0, ST, src, dst, A, B, K, accum);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}
int icm_gather_vectors(const float* const* srcs, int n, int len, float* dst, void* stream) {
  if (!srcs || !dst || n < 1 || n > ICM_GATHER_MAX || len <= 0) return ICM_ERR_ARG;
  icm::GatherPtrs P;
  for (int i = 0; i < ICM_GATHER_MAX; ++i) P.p[i] = srcs[i < n ? i : 0];
  for (int i = 0; i < n; ++i)
    if (!srcs[i]) return ICM_ERR_ARG;
  hipLaunchKernelGGL(icm::gather_vectors_kernel, dim3(n), dim3(256), 0, ST, P, len, dst);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}
int icm_lrp_bwd(const float* g, int64_t g_bs, const float* t, int64_t t_bs, float* dpre, int64_t d_bs, int N, int C,
                int HW, void* stream) {
  if (!g || !t || !dpre || N <= 0 || C <= 0 || HW <= 0) return ICM_ERR_ARG;
  hipLaunchKernelGGL(lrp_bwd_kernel, dim3(grid_for((long long)N * C * HW)), dim3(256), 0, ST, g, (long long)g_bs, t,
                     (long long)t_bs, dpre, (long long)d_bs, N, C, HW);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}
int icm_pixel_unshuffle2(const float* src, float* dst, int N, int C, int H, int W, void* stream) {
  if (!src || !dst || N <= 0 || C <= 0 || H <= 0 || W <= 0) return ICM_ERR_ARG;
  hipLaunchKernelGGL(pixel_unshuffle2_kernel, dim3(grid_for((long long)N * C * 4 * H * W)), dim3(256), 0, ST, src, dst,
                     N, C, H, W);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}
int icm_layernorm_fwd(const float* x, int64_t x_bs, const float* gamma, const float* beta, float* y, int64_t y_bs,
                      float* mean, float* rstd, int N, int C, int HW, float eps, void* stream) {
  if (!x || !gamma || !beta || !y || N <= 0 || C <= 0 || HW <= 0) return ICM_ERR_ARG;
  const long long tiles = ((long long)N * HW + 63) / 64;
  const dim3 grid((unsigned)std::min<long long>(tiles, 256 * 16));
#define ICM_LN_FWD(CPT, NW)                                                                                          \
  hipLaunchKernelGGL((layernorm_fwd_cached_kernel<CPT, NW>), grid, dim3(64 * NW), 0, ST, x, (long long)x_bs, gamma,   \
                     beta, y, (long long)y_bs, mean, rstd, N, HW, eps)
  if (C == 48) ICM_LN_FWD(12, 4);
  else if (C == 96) ICM_LN_FWD(24, 4);
  else if (C == 192) ICM_LN_FWD(24, 8);
  else if (C == 384) ICM_LN_FWD(48, 8);
  else
    hipLaunchKernelGGL(layernorm_fwd_kernel, grid, dim3(256), 0, ST, x, (long long)x_bs, gamma, beta, y, (long long)y_bs,
                       mean, rstd, N, C, HW, eps);
#undef ICM_LN_FWD
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}
int icm_layernorm_bwd(const float* x, int64_t x_bs, const float* dy, int64_t dy_bs, const float* gamma,
                      const float* mean, const float* rstd, float* dx, int64_t dx_bs, float* dgamma, float* dbeta,
                      int N, int C, int HW, int accum_dx, int accum_params, const float* dx_extra, int64_t dx_extra_bs,
                      float* ws, int64_t ws_floats, void* stream) {
  if (!x || !dy || !gamma || !mean || !rstd || N <= 0 || C <= 0 || HW <= 0) return ICM_ERR_ARG;
  const bool cached = C == 48 || C == 96 || C == 192 || C == 384;
  bool params_done = false;
  if (dx) {
    const long long tiles = ((long long)N * HW + 63) / 64;
    long long nblk = std::min<long long>(tiles, 256 * 16);
    // parameter gradients ride on the same pass when the workspace holds one partial row per workgroup
    float* pws = nullptr;
    static const bool nofuse = [] { const char* e = getenv("ICM_LN_NOFUSE"); return e && atoi(e) != 0; }();   // measurement only
    if (!nofuse && cached && C <= 192 && dgamma && dbeta && ws && ws_floats >= 2LL * 64 * C) {
      nblk = std::min<long long>(std::min<long long>(nblk, 1024), ws_floats / (2LL * C));
      pws = ws;
    }
    const dim3 grid((unsigned)nblk);
#define ICM_LN_BWD(CPT, NW, FUSE)                                                                                    \
  hipLaunchKernelGGL((layernorm_bwd_cached_kernel<CPT, NW, FUSE>), grid, dim3(64 * NW), 0, ST, x, (long long)x_bs, dy, \
                     (long long)dy_bs, gamma, mean, rstd, dx, (long long)dx_bs, N, HW, accum_dx, dx_extra,           \
                     (long long)dx_extra_bs, pws)
    if (C == 48) { if (pws) ICM_LN_BWD(12, 4, true); else ICM_LN_BWD(12, 4, false); }
    else if (C == 96) { if (pws) ICM_LN_BWD(24, 4, true); else ICM_LN_BWD(24, 4, false); }
    else if (C == 192) { if (pws) ICM_LN_BWD(24, 8, true); else ICM_LN_BWD(24, 8, false); }
    else if (C == 384) ICM_LN_BWD(48, 8, false);
    else
      hipLaunchKernelGGL(layernorm_bwd_kernel, grid, dim3(256), 0, ST, x, (long long)x_bs, dy, (long long)dy_bs, gamma,
                         mean, rstd, dx, (long long)dx_bs, N, C, HW, accum_dx, dx_extra, (long long)dx_extra_bs);
#undef ICM_LN_BWD
    ICM_CHECK_LAUNCH();
    if (pws) {
      hipLaunchKernelGGL(split_sum_finish_wave_kernel, dim3((2 * C + 3) / 4), dim3(256), 0, ST, ws, dgamma, dbeta, C,
                         (int)nblk, 2, accum_params);
      ICM_CHECK_LAUNCH();
      params_done = true;
    }
  }
  if (dgamma && dbeta && !params_done) {
    // enough pixel chunks to fill the chip whatever C is (C = 48 at the 128x128 level)
    const long long chunks = ((long long)N * HW + 4095) / 4096;
    long long S = std::max<long long>(1, std::min<long long>(chunks, (2048 + C - 1) / C));
    if (!ws) S = 1;
    else S = std::max<long long>(1, std::min<long long>(S, ws_floats / (2LL * C)));
    hipLaunchKernelGGL(layernorm_bwd_params_kernel, dim3(C, (unsigned)S), dim3(256), 0, ST, x, (long long)x_bs, dy,
                       (long long)dy_bs, mean, rstd, dgamma, dbeta, ws, N, C, HW, accum_params);
    ICM_CHECK_LAUNCH();
    if (S > 4) {
      hipLaunchKernelGGL(split_sum_finish_wave_kernel, dim3((2 * C + 3) / 4), dim3(256), 0, ST, ws, dgamma, dbeta, C,
                         (int)S, 2, accum_params);
      ICM_CHECK_LAUNCH();
    } else if (S > 1) {
      hipLaunchKernelGGL(split_sum_finish_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, ST, ws, dgamma, dbeta, C,
                         (int)S, 2, accum_params);
      ICM_CHECK_LAUNCH();
    }
  }
  return ICM_OK;
}
int icm_space_to_depth2(const float* src, float* dst, int N, int C, int H, int W, int inverse, int accum, void* stream) {
  if (!src || !dst || N <= 0 || C <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return ICM_ERR_ARG;
  hipLaunchKernelGGL(space_to_depth2_kernel, dim3(grid_for((long long)N * C * H * W)), dim3(256), 0, ST, src, dst, N, C,
                     H, W, inverse, accum);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}
int icm_residual_scale(const float* shortcut, const float* branch, const float* scale, float* out, int N,
                       int64_t per_sample, void* stream) {
  if (!branch || !scale || !out || N <= 0 || per_sample <= 0) return ICM_ERR_ARG;
  hipLaunchKernelGGL(residual_scale_kernel, dim3(grid_for((long long)N * per_sample)), dim3(256), 0, ST, shortcut,
                     branch, scale, out, (long long)per_sample, (long long)N * per_sample);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}
int icm_im2col(const float* x, float* cols, int N, int C, int H, int W, int K, int stride, int pad, void* stream) {
  if (!x || !cols || N <= 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0 || stride <= 0 || pad < 0) return ICM_ERR_ARG;
  const int OH = (H + 2 * pad - K) / stride + 1, OW = (W + 2 * pad - K) / stride + 1;
  if (OH <= 0 || OW <= 0) return ICM_ERR_ARG;
  hipLaunchKernelGGL(im2col_kernel, dim3(grid_for((long long)N * C * K * K * OH * OW)), dim3(256), 0, ST, x, cols, N, C, H,
                     W, OH, OW, K, stride, pad);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}
int icm_col2im(const float* cols, const float* bias, float* out, int N, int C, int H, int W, int K, int stride, int pad,
               int accum, void* stream) {
  if (!cols || !out || N <= 0 || C <= 0 || H <= 0 || W <= 0 || K <= 0 || stride <= 0 || pad < 0) return ICM_ERR_ARG;
  const int OH = (H + 2 * pad - K) / stride + 1, OW = (W + 2 * pad - K) / stride + 1;
  if (OH <= 0 || OW <= 0) return ICM_ERR_ARG;
  hipLaunchKernelGGL(col2im_kernel, dim3(grid_for((long long)N * C * H * W, 1)), dim3(256), 0, ST, cols, bias, out, N, C,
                     H, W, OH, OW, K, stride, pad, accum);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}
int icm_fill(float* p, int64_t n, float v, void* stream) {
  if (!p || n <= 0) return ICM_ERR_ARG;
  hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, ST, p, (long long)n, v);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}
int icm_clamp(float* p, int64_t n, float lo, float hi, void* stream) {
  if (!p || n <= 0 || !(lo <= hi)) return ICM_ERR_ARG;
  hipLaunchKernelGGL(clamp_kernel, dim3(grid_for(n)), dim3(256), 0, ST, p, (long long)n, lo, hi);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}
int icm_pad2d(const float* src, int N, int C, int H, int W, float* dst, int OH, int OW, int top, int left, float value,
              void* stream) {
  if (!src || !dst || N <= 0 || C <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return ICM_ERR_ARG;
  hipLaunchKernelGGL(pad2d_kernel, dim3(grid_for((long long)N * C * OH * OW)), dim3(256), 0, ST, src, N, C, H, W, dst, OH,
                     OW, top, left, value);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}
// Block order of ZigzagSplits / ZigzagReverse (stf6.py:671-696, 733-758): shell i = blocks whose largest index is i;
// inside a shell the channel-group index runs fastest, then the row-half index, then the column-half index.
int icm_zigzag_order(int num_slices, int num_h, int num_w, int32_t* order, int capacity) {
  if (num_slices <= 0 || num_h <= 0 || num_w <= 0) return -1;
  const int total = num_slices * num_h * num_w;
  if (!order) return total;
  if (capacity < total) return -1;
  int n = 0;
  const int shells = std::max(num_slices, std::max(num_h, num_w));
  for (int i = 0; i < shells; ++i)
    for (int w = 0; w < std::min(i + 1, num_w); ++w)
      for (int h = 0; h < std::min(i + 1, num_h); ++h)
        for (int c = 0; c < std::min(i + 1, num_slices); ++c)
          if (std::max(c, std::max(h, w)) == i) order[n++] = (c * num_h + h) * num_w + w;
  return n;
}
static int zigzag_run(const float* src, float* dst, int64_t x_bs, int B, int C, int H, int W, int ns, int nh, int nw,
                      int reverse, void* stream) {
  if (!src || !dst || B <= 0 || C <= 0 || H <= 0 || W <= 0 || ns <= 0 || nh <= 0 || nw <= 0) return ICM_ERR_ARG;
  if (C % ns || H % nh || W % nw) return ICM_ERR_ARG;   // the reference's view() (stf6.py:664) needs exact blocks
  if (ns * nh * nw > 64) return ICM_ERR_UNSUPPORTED;
  ZigzagDesc d;
  int32_t ord[64];
  const int N = icm_zigzag_order(ns, nh, nw, ord, 64);
  if (N != ns * nh * nw) return ICM_ERR_ARG;
  for (int i = 0; i < 64; ++i) d.blk[i] = i < N ? ord[i] : 0;
  const long long total = (long long)B * C * H * W;
  hipLaunchKernelGGL(zigzag_kernel, dim3(grid_for(total)), dim3(256), 0, ST, src, dst, (long long)x_bs, B, N, C / ns, nh,
                     H / nh, nw, W / nw, reverse, d);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}
int icm_zigzag_splits(const float* x, int64_t x_bs, float* z, int B, int C, int H, int W, int num_slices, int num_h,
                      int num_w, void* stream) {
  return zigzag_run(x, z, x_bs, B, C, H, W, num_slices, num_h, num_w, 0, stream);
}
int icm_zigzag_reverse(const float* z, float* x, int64_t x_bs, int B, int C, int H, int W, int num_slices, int num_h,
                       int num_w, void* stream) {
  return zigzag_run(z, x, x_bs, B, C, H, W, num_slices, num_h, num_w, 1, stream);
}
int icm_rd_loss_fwd(const float* x, const float* x_hat, int64_t n_img_elems, const float* lik_y, int64_t n_y,
                    const float* lik_z, int64_t n_z, int64_t num_pixels, float lmbda, float* out, float* ws,
                    void* stream) {
  if (!x || !x_hat || !lik_y || !lik_z || !out || !ws || n_img_elems <= 0 || num_pixels <= 0) return ICM_ERR_ARG;
  const int nblk = grid_for(n_img_elems, 8);   // <= 2048: 3 * nblk <= ICM_REDUCE_WS_FLOATS
  hipLaunchKernelGGL(rd_reduce_kernel, dim3(nblk), dim3(256), 0, ST, x, x_hat, (long long)n_img_elems, lik_y,
                     (long long)n_y, lik_z, (long long)n_z, ws);
  ICM_CHECK_LAUNCH();
  hipLaunchKernelGGL(rd_finish_kernel, dim3(1), dim3(256), 0, ST, ws, nblk, out, (long long)n_img_elems,
                     (long long)num_pixels, lmbda);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}
int icm_rd_loss_bwd(const float* x, const float* x_hat, int64_t n_img_elems, const float* lik_y, int64_t n_y,
                    const float* lik_z, int64_t n_z, int64_t num_pixels, float lmbda, float gscale, float* dx_hat,
                    float* dlik_y, float* dlik_z, void* stream) {
  if (!x || !x_hat || !lik_y || !lik_z || !dx_hat || !dlik_y || !dlik_z) return ICM_ERR_ARG;
  hipLaunchKernelGGL(rd_bwd_kernel, dim3(grid_for(n_img_elems)), dim3(256), 0, ST, x, x_hat, (long long)n_img_elems,
                     lik_y, (long long)n_y, lik_z, (long long)n_z, (long long)num_pixels, lmbda, gscale, dx_hat, dlik_y,
                     dlik_z);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}
int icm_grad_sqnorm(const float* g, int64_t n, float* out, float* ws, void* stream) {
  if (!g || !out || !ws || n <= 0) return ICM_ERR_ARG;
  const int nblk = grid_for(n, 16);   // <= 2048 <= ICM_REDUCE_WS_FLOATS
  hipLaunchKernelGGL(sqnorm_kernel, dim3(nblk), dim3(256), 0, ST, g, (long long)n, ws);
  ICM_CHECK_LAUNCH();
  hipLaunchKernelGGL(sqnorm_finish_kernel, dim3(1), dim3(256), 0, ST, ws, nblk, out);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}
int icm_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                  double eps, int step, const float* sqnorm, float max_norm, float gscale, void* stream) {
  if (!p || !g || !m || !v || n <= 0 || step < 1) return ICM_ERR_ARG;
  // scalars are formed in double exactly as torch.optim.Adam forms them on the host, then rounded to f32
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2s = sqrt(1.0 - pow(beta2, (double)step));
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, ST, p, g, m, v, (long long)n, (float)(lr / bc1),
                     (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)bc2s,
                     sqnorm, max_norm, gscale, (const float*)nullptr);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}
int icm_adam_step_hyper(float* p, const float* g, float* m, float* v, int64_t n, double beta1, double beta2, double eps,
                        const float* hyper, const float* sqnorm, float max_norm, float gscale, void* stream) {
  if (!p || !g || !m || !v || !hyper || n <= 0) return ICM_ERR_ARG;
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, ST, p, g, m, v, (long long)n, 0.0f, (float)beta1,
                     (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, 1.0f, sqnorm, max_norm, gscale,
                     hyper);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}

const char* icm_strerror(int code) {
  switch (code) {
    case ICM_OK: return "ok";
    case ICM_ERR_ARG: return "invalid argument";
    case ICM_ERR_LAUNCH: return "kernel launch failed";
    case ICM_ERR_UNSUPPORTED: return "unsupported configuration";
    default: return "unknown error";
  }
}
int icm_version(void) { return 1; }

}  // extern "C"
