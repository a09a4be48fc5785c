// Weight-gradient kernel of the convolution family (f32 MFMA, gfx950).
//
//   dW[a][b][t] = sum_{n,p} actS(gs[n,a,p]) * actB(gb[n,b,p*stride - pad + t])
//
// GEMM view: M = a (channels of the small-grid tensor), N = b (channels of the big-grid tensor),
// K = pixels.  Per workgroup: 128 a x (32*TB) b x NT taps, accumulated over its share of the pixel
// tiles (64 small-grid pixels each).  Both operands are staged once per pixel tile into LDS -- the
// small-grid tile as [a][pixel] (row stride 65: conflict-free A fragments), the big-grid halo patch
// as [b][patch] (odd plane stride: conflict-free B fragments) -- and the patch is re-read for every
// tap.  Pixel-split partial sums go to [split][tap][a][b] slabs with 128-B coalesced plain stores and
// are reduced by a second kernel: bitwise reproducible, no float atomics (Guideline 12).
#include <algorithm>
#include "icm_common.h"

namespace icm {

#define WG_MAX_TAPS 32

struct WgDesc {
  const float* gs;
  const float* gb;
  float* ws;
  long long gs_bs, gb_bs;
  int Ca, OH, OW, Cb, H, W, N;
  int S, pad, ntaps, act_s, act_b;
  int lgTW, lgTH, lgTI, PH, PW, PPimg, PPo, lgPWp2;
  FastDiv dTIPH, dPH;
  int tiles_x, tiles_y, tiles_n, ntiles, nsplit;
  int natile, nbtile, ngroups;
  short tapoff[WG_MAX_TAPS];
};

template <int NT, int TB>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgDesc d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* gsT = smem;                       // [128][65]
  float* gbP = smem + 128 * 65;            // [32*TB][PPo]
  int* poff = reinterpret_cast<int*>(gbP + 32 * TB * d.PPo);  // [64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;

  int bid = blockIdx.x;
  const int at = bid % d.natile; bid /= d.natile;
  const int bt = bid % d.nbtile; bid /= d.nbtile;
  const int tg = bid % d.ngroups;
  const int split = bid / d.ngroups;
  const int a0 = at * 128, b0 = bt * 32 * TB, t0 = tg * NT;
  const int nt = min(NT, d.ntaps - t0);
  const int TWm = (1 << d.lgTW) - 1, THm = (1 << d.lgTH) - 1;

  if (tid < 64) {
    const int tx = tid & TWm, ty = (tid >> d.lgTW) & THm, ti = tid >> (d.lgTW + d.lgTH);
    poff[tid] = ti * d.PPimg + ty * d.S * d.PW + tx * d.S;
  }
  int toff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) toff[t] = d.tapoff[min(t0 + t, d.ntaps - 1)];

  f32x16 acc[NT][TB];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int b = 0; b < TB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][b][r] = 0.0f;

  const int OHW = d.OH * d.OW, HW = d.H * d.W;
  const int PWp2m = (1 << d.lgPWp2) - 1;
  const int total_e = (32 * TB * (int)d.dTIPH.d) << d.lgPWp2;

  for (int tile = split; tile < d.ntiles; tile += d.nsplit) {
    int q = tile;
    const int tx_i = q % d.tiles_x; q /= d.tiles_x;
    const int ty_i = q % d.tiles_y;
    const int tn_i = q / d.tiles_y;
    const int ox0 = tx_i << d.lgTW, oy0 = ty_i << d.lgTH, n0 = tn_i << d.lgTI;
    __syncthreads();
    // small-grid tile: 128 channels x 64 pixels
    for (int e = tid; e < 128 * 64; e += 256) {
      const int p = e & 63, a = e >> 6;
      const int tx = p & TWm, ty = (p >> d.lgTW) & THm, ti = p >> (d.lgTW + d.lgTH);
      const int n = n0 + ti, oy = oy0 + ty, ox = ox0 + tx, ca = a0 + a;
      float v = 0.0f;
      if (ca < d.Ca && n < d.N && oy < d.OH && ox < d.OW) {
        v = d.gs[(long long)n * d.gs_bs + (long long)ca * OHW + oy * d.OW + ox];
        v = apply_act(v, d.act_s);
      }
      gsT[a * 65 + p] = v;
    }
    // big-grid halo patch: 32*TB channels
    const int iyb = oy0 * d.S - d.pad, ixb = ox0 * d.S - d.pad;
    for (int e = tid; e < total_e; e += 256) {
      const int px = e & PWp2m;
      const uint32_t r = (uint32_t)e >> d.lgPWp2;
      if (px < d.PW) {
        const uint32_t cb = fdiv(r, d.dTIPH);
        const uint32_t rem = r - cb * d.dTIPH.d;
        const uint32_t ti = fdiv(rem, d.dPH);
        const uint32_t py = rem - ti * d.dPH.d;
        const int c = b0 + (int)cb, n = n0 + (int)ti, iy = iyb + (int)py, ix = ixb + px;
        float v = 0.0f;
        if (c < d.Cb && n < d.N && (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W) {
          v = d.gb[(long long)n * d.gb_bs + (long long)c * HW + iy * d.W + ix];
          v = apply_act(v, d.act_b);
        }
        gbP[cb * d.PPo + ti * d.PPimg + py * d.PW + px] = v;
      }
    }
    __syncthreads();
    const float* arow = gsT + (wave * 32 + l31) * 65 + h;
#pragma unroll 2
    for (int kp = 0; kp < 32; ++kp) {
      const int po = poff[2 * kp + h];
      const float av = arow[2 * kp];
#pragma unroll
      for (int tb = 0; tb < TB; ++tb) {
        const float* bp = gbP + (tb * 32 + l31) * d.PPo + po;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (t < nt) {
            const float bv = bp[toff[t]];
            acc[t][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t][tb], 0, 0, 0);
          }
        }
      }
    }
  }

#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t >= nt) continue;
#pragma unroll
    for (int tb = 0; tb < TB; ++tb) {
      const int b = b0 + tb * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int a = a0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (a < d.Ca && b < d.Cb)
          d.ws[(((long long)split * d.ntaps + t0 + t) * d.Ca + a) * d.Cb + b] = acc[t][tb][r];
      }
    }
  }
}

// dw[(a*Cb + b)*ntaps + t] (+)= sum_s ws[((s*ntaps + t)*Ca + a)*Cb + b]
__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int Ca, int Cb,
                                    int ntaps, int nsplit, int accum) {
  const long long total = (long long)Ca * Cb * ntaps;
  const long long slab = total;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    // i enumerates (t, a, b) with b fastest: coalesced slab reads
    const int b = (int)(i % Cb);
    const long long q = i / Cb;
    const int a = (int)(q % Ca);
    const int t = (int)(q / Ca);
    float s = 0.0f;
    for (int k = 0; k < nsplit; ++k) s += ws[k * slab + i];
    float* o = dw + ((long long)a * Cb + b) * ntaps + t;
    if (accum) s += *o;
    *o = s;
  }
}

struct WgPlan {
  int nt, tb;        // kernel variant
  int lgTW, lgTH, lgTI, PH, PW, PPimg, PPo;
  int tiles_x, tiles_y, tiles_n, ntiles, nsplit, natile, nbtile, ngroups;
  size_t lds;
};

static int plan_wgrad(const icm_wgrad_args& a, WgPlan& p) {
  if (!a.gs || !a.gb || a.N <= 0 || a.Ca <= 0 || a.Cb <= 0 || a.KH * a.KW > WG_MAX_TAPS) return ICM_ERR_ARG;
  if (a.stride != 1 && a.stride != 2) return ICM_ERR_UNSUPPORTED;
  if (a.OH != (a.H + 2 * a.pad - a.KH) / a.stride + 1 || a.OW != (a.W + 2 * a.pad - a.KW) / a.stride + 1)
    return ICM_ERR_ARG;
  const int ntaps = a.KH * a.KW;
  if (ntaps == 1) { p.nt = 1; p.tb = 4; }
  else if (ntaps <= 9) { p.nt = 9; p.tb = 1; }
  else { p.nt = 5; p.tb = 1; }
  p.lgTW = std::min(4, ceil_log2(a.OW));
  p.lgTH = std::min(6 - p.lgTW, ceil_log2(a.OH));
  p.lgTI = 6 - p.lgTW - p.lgTH;
  const int TW = 1 << p.lgTW, TH = 1 << p.lgTH, TI = 1 << p.lgTI;
  p.PW = (TW - 1) * a.stride + a.KW;
  p.PH = (TH - 1) * a.stride + a.KH;
  p.PPimg = p.PH * p.PW;
  p.PPo = TI * p.PPimg;
  if ((p.PPo & 1) == 0) p.PPo += 1;
  p.tiles_x = cdiv(a.OW, TW); p.tiles_y = cdiv(a.OH, TH); p.tiles_n = cdiv(a.N, TI);
  p.ntiles = p.tiles_x * p.tiles_y * p.tiles_n;
  p.natile = cdiv(a.Ca, 128); p.nbtile = cdiv(a.Cb, 32 * p.tb); p.ngroups = cdiv(ntaps, p.nt);
  const int base = p.natile * p.nbtile * p.ngroups;
  p.nsplit = std::max(1, std::min(p.ntiles, cdiv(1024, base)));
  p.lds = (size_t)(128 * 65 + 32 * p.tb * p.PPo) * 4 + 64 * 4;
  if (p.lds > 160 * 1024) return ICM_ERR_UNSUPPORTED;
  return ICM_OK;
}

}  // namespace icm

extern "C" {

int64_t icm_wgrad_workspace_floats(const icm_wgrad_args* a) {
  icm::WgPlan p;
  if (!a || icm::plan_wgrad(*a, p)) return -1;
  return (int64_t)p.nsplit * a->KH * a->KW * a->Ca * a->Cb;
}

int icm_conv_wgrad(const icm_wgrad_args* a, void* stream_) {
  using namespace icm;
  if (!a || !a->dw || !a->ws) return ICM_ERR_ARG;
  WgPlan p;
  int rc = plan_wgrad(*a, p);
  if (rc) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  WgDesc d;
  d.gs = a->gs; d.gb = a->gb; d.ws = a->ws; d.gs_bs = a->gs_bs; d.gb_bs = a->gb_bs;
  d.Ca = a->Ca; d.OH = a->OH; d.OW = a->OW; d.Cb = a->Cb; d.H = a->H; d.W = a->W; d.N = a->N;
  d.S = a->stride; d.pad = a->pad; d.ntaps = a->KH * a->KW; d.act_s = a->act_s; d.act_b = a->act_b;
  d.lgTW = p.lgTW; d.lgTH = p.lgTH; d.lgTI = p.lgTI; d.PH = p.PH; d.PW = p.PW; d.PPimg = p.PPimg; d.PPo = p.PPo;
  d.lgPWp2 = ceil_log2(p.PW);
  d.dTIPH = make_fastdiv((uint32_t)((1 << p.lgTI) * p.PH));
  d.dPH = make_fastdiv((uint32_t)p.PH);
  d.tiles_x = p.tiles_x; d.tiles_y = p.tiles_y; d.tiles_n = p.tiles_n; d.ntiles = p.ntiles; d.nsplit = p.nsplit;
  d.natile = p.natile; d.nbtile = p.nbtile; d.ngroups = p.ngroups;
  for (int t = 0; t < WG_MAX_TAPS; ++t) d.tapoff[t] = 0;
  for (int kh = 0; kh < a->KH; ++kh)
    for (int kw = 0; kw < a->KW; ++kw) d.tapoff[kh * a->KW + kw] = (short)(kh * p.PW + kw);
  const long long nblk = (long long)p.natile * p.nbtile * p.ngroups * p.nsplit;
  void (*fn)(const WgDesc) = nullptr;
  if (p.nt == 1 && p.tb == 4) fn = wgrad_kernel<1, 4>;
  else if (p.nt == 9) fn = wgrad_kernel<9, 1>;
  else fn = wgrad_kernel<5, 1>;
  if (p.lds > 64 * 1024)
    hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);
  hipLaunchKernelGGL(fn, dim3((unsigned)nblk), dim3(256), p.lds, stream, d);
  ICM_CHECK_LAUNCH();
  const long long total = (long long)a->Ca * a->Cb * d.ntaps;
  const int rblocks = (int)std::min<long long>((total + 255) / 256, 2048);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rblocks), dim3(256), 0, stream, a->ws, a->dw, a->Ca, a->Cb,
                     d.ntaps, p.nsplit, a->accum);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}

}  // extern "C"
