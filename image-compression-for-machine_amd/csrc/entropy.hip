// EntropyBottleneck and GaussianConditional likelihood kernels (forward + backward), gfx950.
//
// EntropyBottleneck (entropy_models.py:400-433): one workgroup owns one channel: its 58 parameters are
// transformed once (softplus / tanh) into LDS, every lane evaluates both logits chains of its elements
// in registers, and the backward pass reduces the 58 parameter gradients inside the workgroup
// (wavefront shuffles + LDS) -- no atomics, bitwise reproducible.
// GaussianConditional (entropy_models.py:626-659) is fused with ste_round(y-mu)+mu (cnn.py:171-173).
#include <algorithm>
#include "icm_common.h"

namespace icm {

// per-channel parameter block in LDS (transformed)
struct EbChan {
  float sp0[3], b0[3], tf0[3];
  float sp[3][9], b[3][3], tf[3][3];
  float sp4[3], b4;
  float med;
};

__device__ __forceinline__ float softplus_f(float x) {
  // torch softplus (beta=1, threshold=20)
  return x > 20.0f ? x : log1pf(expf(x));
}

struct EbPtrs {
  const float* matrix[5];
  const float* bias[5];
  const float* factor[4];
  const float* quantiles;
};

__device__ __forceinline__ void eb_load_channel(const EbPtrs& P, int c, EbChan* ch, int tid) {
  if (tid < 3) {
    ch->sp0[tid] = softplus_f(P.matrix[0][c * 3 + tid]);
    ch->b0[tid] = P.bias[0][c * 3 + tid];
    ch->tf0[tid] = tanhf(P.factor[0][c * 3 + tid]);
    ch->sp4[tid] = softplus_f(P.matrix[4][c * 3 + tid]);
  }
  if (tid >= 64 && tid < 64 + 27) {
    const int q = tid - 64, k = q / 9, e = q % 9;
    ch->sp[k][e] = softplus_f(P.matrix[1 + k][c * 9 + e]);
  }
  if (tid >= 128 && tid < 128 + 9) {
    const int q = tid - 128, k = q / 3, e = q % 3;
    ch->b[k][e] = P.bias[1 + k][c * 3 + e];
    ch->tf[k][e] = tanhf(P.factor[1 + k][c * 3 + e]);
  }
  if (tid == 192) {
    ch->b4 = P.bias[4][c];
    ch->med = P.quantiles[c * 3 + 1];
  }
}

// forward chain keeping what backward needs: h[k][*] (layer inputs, k=1..4) and th[k][*] = tanh(u)
struct EbTrace {
  float hin[4][3];  // outputs of gated layers 0..3 (inputs to layers 1..4)
  float th[4][3];   // tanh(u) of gated layers
};

__device__ __forceinline__ float eb_chain(const EbChan& ch, float v, EbTrace* tr) {
  float h[3];
#pragma unroll
  for (int o = 0; o < 3; ++o) {
    const float u = ch.sp0[o] * v + ch.b0[o];
    const float t = tanhf(u);
    h[o] = u + ch.tf0[o] * t;
    if (tr) { tr->th[0][o] = t; tr->hin[0][o] = h[o]; }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float n[3];
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      // torch.matmul on [3x3]x[3xL]: fp32 dot in index order
      float u = ch.sp[k][o * 3 + 0] * h[0];
      u += ch.sp[k][o * 3 + 1] * h[1];
      u += ch.sp[k][o * 3 + 2] * h[2];
      u += ch.b[k][o];
      const float t = tanhf(u);
      n[o] = u + ch.tf[k][o] * t;
      if (tr) { tr->th[k + 1][o] = t; tr->hin[k + 1][o] = n[o]; }
    }
#pragma unroll
    for (int o = 0; o < 3; ++o) h[o] = n[o];
  }
  float out = ch.sp4[0] * h[0];
  out += ch.sp4[1] * h[1];
  out += ch.sp4[2] * h[2];
  return out + ch.b4;
}

__global__ __launch_bounds__(256) void eb_fwd_kernel(const float* __restrict__ z, const float* __restrict__ noise,
                                                     const EbPtrs P, float* __restrict__ lik, float* __restrict__ zt,
                                                     int N, int C, int HW, float lik_bound) {
  __shared__ EbChan ch;
  const int c = blockIdx.x, tid = threadIdx.x;
  eb_load_channel(P, c, &ch, tid);
  __syncthreads();
  const int L = N * HW;
  for (int e = tid; e < L; e += 256) {
    const int n = e / HW, p = e - n * HW;
    const long long idx = ((long long)n * C + c) * HW + p;
    float v;
    if (noise) v = z[idx] + noise[idx];
    else v = rintf(z[idx] - ch.med) + ch.med;
    const float lo = eb_chain(ch, v - 0.5f, nullptr);
    const float up = eb_chain(ch, v + 0.5f, nullptr);
    const float sum = lo + up;
    const float s = sum > 0.0f ? -1.0f : (sum < 0.0f ? 1.0f : 0.0f);
    const float l = fabsf(sigmoid_f(s * up) - sigmoid_f(s * lo));
    lik[idx] = fmaxf(l, lik_bound);
    if (zt) zt[idx] = v;
  }
}

struct EbGradPtrs {
  float* matrix[5];
  float* bias[5];
  float* factor[4];
  float* dmedian;
};

// accumulate the parameter-gradient contributions of one chain given d(out)
__device__ __forceinline__ float eb_chain_bwd(const EbChan& ch, float v, const EbTrace& tr, float dout, float* ga) {
  // ga layout (58): [0..2] dM0_sp, [3..5] db0, [6..8] dtf0, then per k: [9+15k .. ] dM_sp(9), db(3), dtf(3); [54..56] dM4_sp, [57] db4
  float dh[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    ga[54 + i] += dout * tr.hin[3][i];
    dh[i] = ch.sp4[i] * dout;
  }
  ga[57] += dout;
#pragma unroll
  for (int k = 2; k >= 0; --k) {
    float du[3];
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      const float t = tr.th[k + 1][o];
      du[o] = dh[o] * (1.0f + ch.tf[k][o] * (1.0f - t * t));
      ga[9 + 15 * k + 12 + o] += dh[o] * t;
      ga[9 + 15 * k + 9 + o] += du[o];
#pragma unroll
      for (int i = 0; i < 3; ++i) ga[9 + 15 * k + o * 3 + i] += du[o] * tr.hin[k][i];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
      dh[i] = ch.sp[k][0 * 3 + i] * du[0] + ch.sp[k][1 * 3 + i] * du[1] + ch.sp[k][2 * 3 + i] * du[2];
  }
  float dv = 0.0f;
#pragma unroll
  for (int o = 0; o < 3; ++o) {
    const float t = tr.th[0][o];
    const float du = dh[o] * (1.0f + ch.tf0[o] * (1.0f - t * t));
    ga[6 + o] += dh[o] * t;
    ga[3 + o] += du;
    ga[0 + o] += du * v;
    dv += ch.sp0[o] * du;
  }
  return dv;
}

__global__ __launch_bounds__(256) void eb_bwd_kernel(const float* __restrict__ z, const float* __restrict__ noise,
                                                     const EbPtrs P, const float* __restrict__ dlik,
                                                     float* __restrict__ dz, const EbGradPtrs G, int N, int C, int HW,
                                                     float lik_bound, int accum_dz) {
  __shared__ EbChan ch;
  __shared__ float red[4][60];
  const int c = blockIdx.x, tid = threadIdx.x;
  eb_load_channel(P, c, &ch, tid);
  __syncthreads();
  float ga[59];
#pragma unroll
  for (int i = 0; i < 59; ++i) ga[i] = 0.0f;  // ga[58] = d median (eval)
  const int L = N * HW;
  for (int e = tid; e < L; e += 256) {
    const int n = e / HW, p = e - n * HW;
    const long long idx = ((long long)n * C + c) * HW + p;
    float v;
    if (noise) v = z[idx] + noise[idx];
    else v = rintf(z[idx] - ch.med) + ch.med;
    EbTrace tl, tu;
    const float lo = eb_chain(ch, v - 0.5f, &tl);
    const float up = eb_chain(ch, v + 0.5f, &tu);
    const float sum = lo + up;
    const float s = sum > 0.0f ? -1.0f : (sum < 0.0f ? 1.0f : 0.0f);
    const float su = sigmoid_f(s * up), sl = sigmoid_f(s * lo);
    const float diff = su - sl;
    const float lraw = fabsf(diff);
    float g = dlik[idx];
    if (!(lraw >= lik_bound || g < 0.0f)) g = 0.0f;
    const float sgn = diff > 0.0f ? 1.0f : (diff < 0.0f ? -1.0f : 0.0f);
    const float dup = g * sgn * s * su * (1.0f - su);
    const float dlo = -g * sgn * s * sl * (1.0f - sl);
    float dv = eb_chain_bwd(ch, v - 0.5f, tl, dlo, ga);
    dv += eb_chain_bwd(ch, v + 0.5f, tu, dup, ga);
    if (noise) {
      float r = dv;
      if (accum_dz) r += dz[idx];
      dz[idx] = r;
    } else {
      ga[58] += dv;
      if (!accum_dz) dz[idx] = 0.0f;
    }
  }
  // block reduction of the 59 accumulators
#pragma unroll
  for (int i = 0; i < 59; ++i) {
    const float s = wave_sum(ga[i]);
    if ((tid & 63) == 0) red[tid >> 6][i] = s;
  }
  __syncthreads();
  if (tid < 59) {
    const float s = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    // chain rule through softplus (sigmoid) / tanh of the stored parameters
    if (tid < 3) {
      G.matrix[0][c * 3 + tid] = s * sigmoid_f(P.matrix[0][c * 3 + tid]);
    } else if (tid < 6) {
      G.bias[0][c * 3 + tid - 3] = s;
    } else if (tid < 9) {
      const float tf = ch.tf0[tid - 6];
      G.factor[0][c * 3 + tid - 6] = s * (1.0f - tf * tf);
    } else if (tid < 54) {
      const int q = tid - 9, k = q / 15, e = q % 15;
      if (e < 9) G.matrix[1 + k][c * 9 + e] = s * sigmoid_f(P.matrix[1 + k][c * 9 + e]);
      else if (e < 12) G.bias[1 + k][c * 3 + e - 9] = s;
      else {
        const float tf = ch.tf[k][e - 12];
        G.factor[1 + k][c * 3 + e - 12] = s * (1.0f - tf * tf);
      }
    } else if (tid < 57) {
      G.matrix[4][c * 3 + tid - 54] = s * sigmoid_f(P.matrix[4][c * 3 + tid - 54]);
    } else if (tid == 57) {
      G.bias[4][c] = s;
    } else if (G.dmedian) {
      G.dmedian[c] = s;
    }
  }
}

// aux loss: sum_c sum_q |F_detached(quantiles[c,q]) - target_q| ; gradient wrt quantiles only
__global__ __launch_bounds__(256) void eb_aux_kernel(const EbPtrs P, float* loss, float* dq, int C, float target) {
  __shared__ float red[4];
  float part = 0.0f;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < C * 3; e += gridDim.x * 256) {
    const int c = e / 3, q = e % 3;
    EbChan ch;
    for (int i = 0; i < 3; ++i) {
      ch.sp0[i] = softplus_f(P.matrix[0][c * 3 + i]);
      ch.b0[i] = P.bias[0][c * 3 + i];
      ch.tf0[i] = tanhf(P.factor[0][c * 3 + i]);
      ch.sp4[i] = softplus_f(P.matrix[4][c * 3 + i]);
      for (int k = 0; k < 3; ++k) {
        ch.b[k][i] = P.bias[1 + k][c * 3 + i];
        ch.tf[k][i] = tanhf(P.factor[1 + k][c * 3 + i]);
      }
    }
    for (int k = 0; k < 3; ++k)
      for (int i = 0; i < 9; ++i) ch.sp[k][i] = softplus_f(P.matrix[1 + k][c * 9 + i]);
    ch.b4 = P.bias[4][c];
    ch.med = 0.0f;
    const float v = P.quantiles[c * 3 + q];
    EbTrace tr;
    const float f = eb_chain(ch, v, &tr);
    const float tq = (q == 0) ? -target : (q == 1 ? 0.0f : target);
    const float diff = f - tq;
    part += fabsf(diff);
    const float sg = diff > 0.0f ? 1.0f : (diff < 0.0f ? -1.0f : 0.0f);
    float scratch[59];
    for (int i = 0; i < 59; ++i) scratch[i] = 0.0f;
    dq[e] = eb_chain_bwd(ch, v, tr, sg, scratch);
  }
  // ONE workgroup (C * 3 = 576 chain evaluations): fixed-order sum, no float atomics (every data-parallel rank
  // must see the same value)
  part = wave_sum(part);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) *loss = (red[0] + red[1]) + (red[2] + red[3]);
}

// ------------------------------------------------------------------------------ GaussianConditional
__device__ __forceinline__ float std_cdf(float t) { return 0.5f * erfcf(-0.70710678118654752440f * t); }
__device__ __forceinline__ float std_pdf(float t) { return 0.39894228040143267794f * expf(-0.5f * t * t); }

struct GcDesc {
  const float* y; const float* mu; const float* sc; const float* noise;
  float* lik; float* yh; float* yh2;
  long long y_bs, mu_bs, sc_bs, nz_bs, lik_bs, yh_bs, yh2_bs;
  int N, C, HW;
  float scale_bound, lik_bound;
};

__global__ void gc_fwd_kernel(const GcDesc d) {
  const long long per = (long long)d.C * d.HW, total = per * d.N;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / per, r = i - n * per;
    const float y = d.y[n * d.y_bs + r], mu = d.mu[n * d.mu_bs + r], sc = d.sc[n * d.sc_bs + r];
    const float t = y - mu;
    const float rt = rintf(t);
    const float out = d.noise ? (y + d.noise[n * d.nz_bs + r]) : (rt + mu);
    const float v = fabsf(out - mu);
    const float s = fmaxf(sc, d.scale_bound);
    const float up = std_cdf((0.5f - v) / s), lo = std_cdf((-0.5f - v) / s);
    d.lik[n * d.lik_bs + r] = fmaxf(up - lo, d.lik_bound);
    const float yh = ((rt - t) + t) + mu;
    if (d.yh) d.yh[n * d.yh_bs + r] = yh;
    if (d.yh2) d.yh2[n * d.yh2_bs + r] = yh;
  }
}

struct GcBwdDesc {
  const float* y; const float* mu; const float* sc; const float* noise; const float* dlik; const float* dyh;
  float* dy; float* dmu; float* dsc;
  long long y_bs, mu_bs, sc_bs, nz_bs, dl_bs, dyh_bs, dy_bs, dmu_bs, dsc_bs;
  int N, C, HW, accum_dy;
  float scale_bound, lik_bound;
};

__global__ void gc_bwd_kernel(const GcBwdDesc d) {
  const long long per = (long long)d.C * d.HW, total = per * d.N;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / per, r = i - n * per;
    const float y = d.y[n * d.y_bs + r], mu = d.mu[n * d.mu_bs + r], sc = d.sc[n * d.sc_bs + r];
    const float out = d.noise ? (y + d.noise[n * d.nz_bs + r]) : (rintf(y - mu) + mu);
    const float w = out - mu;
    const float v = fabsf(w);
    const float s = fmaxf(sc, d.scale_bound);
    const float u = (0.5f - v) / s, l = (-0.5f - v) / s;
    const float lraw = std_cdf(u) - std_cdf(l);
    float g = d.dlik[n * d.dl_bs + r];
    if (!(lraw >= d.lik_bound || g < 0.0f)) g = 0.0f;
    const float pu = std_pdf(u), pl = std_pdf(l);
    const float dv = g * (pl - pu) / s;
    float gs = g * (l * pl - u * pu) / s;
    if (!(sc >= d.scale_bound || gs < 0.0f)) gs = 0.0f;
    const float sg = w > 0.0f ? 1.0f : (w < 0.0f ? -1.0f : 0.0f);
    float gy = 0.0f, gmu = 0.0f;
    if (d.noise) {
      gy = dv * sg;
      gmu = -gy;
    }
    if (d.dyh) gy += d.dyh[n * d.dyh_bs + r];
    float* dyp = d.dy + n * d.dy_bs + r;
    if (d.accum_dy) gy += *dyp;
    *dyp = gy;
    d.dmu[n * d.dmu_bs + r] = gmu;
    d.dsc[n * d.dsc_bs + r] = gs;
  }
}

// ------------------------------------------------------------------------------ codec tables / symbols (update, compress)
// EntropyBottleneck.update() (entropy_models.py:354-393): per channel, minima = clamp(ceil(med - q0), 0),
// maxima = clamp(ceil(q2 - med), 0) (ints); then the pmf of the integer grid med - minima + k, k < max_length.
__global__ void eb_table_bounds_kernel(const float* __restrict__ quantiles, int C, int* __restrict__ minima,
                                       int* __restrict__ maxima) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float q0 = quantiles[c * 3], med = quantiles[c * 3 + 1], q2 = quantiles[c * 3 + 2];
  minima[c] = max((int)ceilf(med - q0), 0);
  maxima[c] = max((int)ceilf(q2 - med), 0);
}
__global__ __launch_bounds__(256) void eb_pmf_table_kernel(const EbPtrs P, const int* __restrict__ minima, int C,
                                                           int max_length, float* __restrict__ pmf,
                                                           float* __restrict__ tail) {
  __shared__ EbChan ch;
  const int c = blockIdx.x, tid = threadIdx.x;
  eb_load_channel(P, c, &ch, tid);
  __syncthreads();
  const float start = ch.med - (float)minima[c];
  for (int k = tid; k < max_length; k += 256) {
    const float v = (float)k + start;
    const float lo = eb_chain(ch, v - 0.5f, nullptr);
    const float up = eb_chain(ch, v + 0.5f, nullptr);
    const float sum = lo + up;
    const float s = sum > 0.0f ? -1.0f : (sum < 0.0f ? 1.0f : 0.0f);
    pmf[(long long)c * max_length + k] = fabsf(sigmoid_f(s * up) - sigmoid_f(s * lo));
  }
  if (tid == 0) {   // tail mass: sigmoid(lower of the first sample) + sigmoid(-upper of the last of max_length samples)
    const float lo0 = eb_chain(ch, start - 0.5f, nullptr);
    const float upL = eb_chain(ch, (float)(max_length - 1) + start + 0.5f, nullptr);
    tail[c] = sigmoid_f(lo0) + sigmoid_f(-upL);
  }
}
// GaussianConditional.update() (entropy_models.py:598-624): centers = ceil(scale * multiplier); pmf over |k - center|
__global__ void gc_table_centers_kernel(const float* __restrict__ table, int ns, float multiplier,
                                        int* __restrict__ centers) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ns) centers[i] = (int)ceilf(table[i] * multiplier);
}
__global__ void gc_pmf_table_kernel(const float* __restrict__ table, const int* __restrict__ centers, int ns,
                                    int max_length, float* __restrict__ pmf, float* __restrict__ tail) {
  const int i = blockIdx.y;
  const float sc = table[i];
  const int ctr = centers[i];
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < max_length; k += gridDim.x * blockDim.x) {
    const float v = (float)abs(k - ctr);
    const float up = std_cdf((0.5f - v) / sc), lo = std_cdf((-0.5f - v) / sc);
    pmf[(long long)i * max_length + k] = up - lo;
    if (k == 0) tail[i] = 2.0f * lo;
  }
}
// GaussianConditional.build_indexes (entropy_models.py:661-666): (ns - 1) - #{s in table[:-1] : max(scale, bound) <= s}
__global__ void gc_build_indexes_kernel(const float* __restrict__ scale, long long sbs, const float* __restrict__ table,
                                        int ns, float bound, int* __restrict__ idx, int N, int C, int HW) {
  const long long per = (long long)C * HW, total = per * N;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / per, r = i - n * per;
    const float s = fmaxf(scale[n * sbs + r], bound);
    int v = ns - 1;
    for (int k = 0; k < ns - 1; ++k) v -= (s <= table[k]) ? 1 : 0;
    idx[i] = v;
  }
}
// EntropyModel.quantize (entropy_models.py:126-150): r = round(x - mean); "symbols" -> int32, "dequantize" -> r + mean.
// mean address = means + n*m_bs + c*m_cs + p*m_ps (full tensors, or per-channel medians with m_bs = m_ps = 0).
__global__ void quantize_kernel(const float* __restrict__ x, long long xbs, const float* __restrict__ means,
                                long long m_bs, long long m_cs, long long m_ps, int* __restrict__ sym,
                                float* __restrict__ deq, int N, int C, int HW) {
  const long long per = (long long)C * HW, total = per * N;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / per, r = i - n * per;
    const int c = (int)(r / HW), p = (int)(r - (long long)c * HW);
    const float m = means ? means[n * m_bs + c * m_cs + p * m_ps] : 0.0f;
    const float q = rintf(x[n * xbs + r] - m);
    if (sym) sym[i] = (int)q;
    if (deq) deq[i] = q + m;
  }
}
// EntropyModel.dequantize (entropy_models.py:159-166): float(symbols) + mean
__global__ void dequantize_kernel(const int* __restrict__ sym, const float* __restrict__ means, long long m_bs,
                                  long long m_cs, long long m_ps, float* __restrict__ out, long long obs, int N, int C,
                                  int HW) {
  const long long per = (long long)C * HW, total = per * N;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / per, r = i - n * per;
    const int c = (int)(r / HW), p = (int)(r - (long long)c * HW);
    const float m = means ? means[n * m_bs + c * m_cs + p * m_ps] : 0.0f;
    out[n * obs + r] = (float)sym[i] + m;
  }
}

}  // namespace icm

using namespace icm;

static inline EbPtrs to_ptrs(const icm_eb_params* p) {
  EbPtrs P;
  for (int i = 0; i < 5; ++i) { P.matrix[i] = p->matrix[i]; P.bias[i] = p->bias[i]; }
  for (int i = 0; i < 4; ++i) P.factor[i] = p->factor[i];
  P.quantiles = p->quantiles;
  return P;
}
static inline bool eb_ok(const icm_eb_params* p) {
  if (!p || !p->quantiles) return false;
  for (int i = 0; i < 5; ++i) if (!p->matrix[i] || !p->bias[i]) return false;
  for (int i = 0; i < 4; ++i) if (!p->factor[i]) return false;
  return true;
}

extern "C" {

int icm_eb_likelihood_fwd(const float* z, const float* noise, const icm_eb_params* p, float* lik, float* zt, int N,
                          int C, int HW, float lik_bound, void* stream) {
  if (!z || !lik || !eb_ok(p) || N <= 0 || C <= 0 || HW <= 0) return ICM_ERR_ARG;
  hipLaunchKernelGGL(eb_fwd_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, z, noise, to_ptrs(p), lik, zt, N, C, HW,
                     lik_bound);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}

int icm_eb_likelihood_bwd(const float* z, const float* noise, const icm_eb_params* p, const float* dlik, float* dz,
                          const icm_eb_grads* g, int N, int C, int HW, float lik_bound, int accum_dz, void* stream) {
  if (!z || !dlik || !dz || !g || !eb_ok(p) || N <= 0 || C <= 0 || HW <= 0) return ICM_ERR_ARG;
  EbGradPtrs G;
  for (int i = 0; i < 5; ++i) { G.matrix[i] = g->matrix[i]; G.bias[i] = g->bias[i]; if (!G.matrix[i] || !G.bias[i]) return ICM_ERR_ARG; }
  for (int i = 0; i < 4; ++i) { G.factor[i] = g->factor[i]; if (!G.factor[i]) return ICM_ERR_ARG; }
  G.dmedian = g->dmedian;
  hipLaunchKernelGGL(eb_bwd_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, z, noise, to_ptrs(p), dlik, dz, G, N, C,
                     HW, lik_bound, accum_dz);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}

int icm_eb_aux_loss(const icm_eb_params* p, float* loss, float* dquantiles, int C, float target, void* stream) {
  if (!eb_ok(p) || !loss || !dquantiles || C <= 0) return ICM_ERR_ARG;
  hipLaunchKernelGGL(eb_aux_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream,
                     to_ptrs(p), loss, dquantiles, C, target);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}

static inline int ew_blocks(long long total) {
  return (int)std::max<long long>(1, std::min<long long>((total + 255) / 256, 2048));
}

int icm_eb_table_bounds(const float* quantiles, int C, int32_t* minima, int32_t* maxima, void* stream) {
  if (!quantiles || !minima || !maxima || C <= 0) return ICM_ERR_ARG;
  hipLaunchKernelGGL(eb_table_bounds_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, quantiles, C,
                     minima, maxima);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}
int icm_eb_pmf_table(const icm_eb_params* p, const int32_t* minima, int C, int max_length, float* pmf, float* tail_mass,
                     void* stream) {
  if (!eb_ok(p) || !minima || !pmf || !tail_mass || C <= 0 || max_length <= 0) return ICM_ERR_ARG;
  hipLaunchKernelGGL(eb_pmf_table_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, to_ptrs(p), minima, C, max_length,
                     pmf, tail_mass);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}
int icm_gc_table_centers(const float* scale_table, int ns, float multiplier, int32_t* centers, void* stream) {
  if (!scale_table || !centers || ns <= 0) return ICM_ERR_ARG;
  hipLaunchKernelGGL(gc_table_centers_kernel, dim3((ns + 255) / 256), dim3(256), 0, (hipStream_t)stream, scale_table, ns,
                     multiplier, centers);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}
int icm_gc_pmf_table(const float* scale_table, const int32_t* centers, int ns, int max_length, float* pmf,
                     float* tail_mass, void* stream) {
  if (!scale_table || !centers || !pmf || !tail_mass || ns <= 0 || max_length <= 0) return ICM_ERR_ARG;
  hipLaunchKernelGGL(gc_pmf_table_kernel, dim3((max_length + 255) / 256, ns), dim3(256), 0, (hipStream_t)stream,
                     scale_table, centers, ns, max_length, pmf, tail_mass);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}
int icm_gc_build_indexes(const float* scale, int64_t scale_bs, const float* scale_table, int ns, float scale_bound,
                         int32_t* indexes, int N, int C, int HW, void* stream) {
  if (!scale || !scale_table || !indexes || ns <= 0 || N <= 0 || C <= 0 || HW <= 0) return ICM_ERR_ARG;
  hipLaunchKernelGGL(gc_build_indexes_kernel, dim3(ew_blocks((long long)N * C * HW)), dim3(256), 0, (hipStream_t)stream,
                     scale, (long long)scale_bs, scale_table, ns, scale_bound, indexes, N, C, HW);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}
int icm_quantize(const float* x, int64_t x_bs, const float* means, int64_t m_bs, int64_t m_cs, int64_t m_ps,
                 int32_t* symbols, float* dequantized, int N, int C, int HW, void* stream) {
  if (!x || (!symbols && !dequantized) || N <= 0 || C <= 0 || HW <= 0) return ICM_ERR_ARG;
  hipLaunchKernelGGL(quantize_kernel, dim3(ew_blocks((long long)N * C * HW)), dim3(256), 0, (hipStream_t)stream, x,
                     (long long)x_bs, means, (long long)m_bs, (long long)m_cs, (long long)m_ps, symbols, dequantized, N,
                     C, HW);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}
int icm_dequantize(const int32_t* symbols, const float* means, int64_t m_bs, int64_t m_cs, int64_t m_ps, float* out,
                   int64_t out_bs, int N, int C, int HW, void* stream) {
  if (!symbols || !out || N <= 0 || C <= 0 || HW <= 0) return ICM_ERR_ARG;
  hipLaunchKernelGGL(dequantize_kernel, dim3(ew_blocks((long long)N * C * HW)), dim3(256), 0, (hipStream_t)stream,
                     symbols, means, (long long)m_bs, (long long)m_cs, (long long)m_ps, out, (long long)out_bs, N, C, HW);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}

int icm_gc_likelihood_ste_fwd(const float* y, int64_t y_bs, const float* mu, int64_t mu_bs, const float* scale,
                              int64_t sc_bs, const float* noise, int64_t nz_bs, float* lik, int64_t lik_bs, float* yh,
                              int64_t yh_bs, float* yh2, int64_t yh2_bs, int N, int C, int HW, float scale_bound,
                              float lik_bound, void* stream) {
  if (!y || !mu || !scale || !lik || N <= 0 || C <= 0 || HW <= 0 || scale_bound <= 0) return ICM_ERR_ARG;
  GcDesc d{y, mu, scale, noise, lik, yh, yh2, y_bs, mu_bs, sc_bs, nz_bs, lik_bs, yh_bs, yh2_bs, N, C, HW, scale_bound, lik_bound};
  const long long total = (long long)N * C * HW;
  const int blocks = (int)std::max<long long>(1, std::min<long long>((total + 255) / 256, 2048));
  hipLaunchKernelGGL(gc_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}

int icm_gc_likelihood_ste_bwd(const float* y, int64_t y_bs, const float* mu, int64_t mu_bs, const float* scale,
                              int64_t sc_bs, const float* noise, int64_t nz_bs, const float* dlik, int64_t dl_bs,
                              const float* dyh, int64_t dyh_bs, float* dy, int64_t dy_bs, float* dmu, int64_t dmu_bs,
                              float* dscale, int64_t dsc_bs, int N, int C, int HW, float scale_bound, float lik_bound,
                              int accum_dy, void* stream) {
  if (!y || !mu || !scale || !dlik || !dy || !dmu || !dscale || N <= 0 || C <= 0 || HW <= 0) return ICM_ERR_ARG;
  GcBwdDesc d{y, mu, scale, noise, dlik, dyh, dy, dmu, dscale, y_bs, mu_bs, sc_bs, nz_bs, dl_bs, dyh_bs, dy_bs, dmu_bs,
              dsc_bs, N, C, HW, accum_dy, scale_bound, lik_bound};
  const long long total = (long long)N * C * HW;
  const int blocks = (int)std::max<long long>(1, std::min<long long>((total + 255) / 256, 2048));
  hipLaunchKernelGGL(gc_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}

}  // extern "C"
