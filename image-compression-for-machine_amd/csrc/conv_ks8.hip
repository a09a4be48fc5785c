// Latency-bound convolutions (few output tiles, deep K: the 4 096-pixel slice-chain / hyper-path / dim-320 gate
// problems) on f32 MFMA for gfx950: EIGHT MFMA waves share one block of four output tiles and split the K loop.
//
// The staged kernel (conv_igemm.hip) gives such a problem one MFMA wave per tile (or 2-4 with its split-K tilings of
// single 32 x 32 tiles): a 64-co x 64-pixel block of a 224 -> 176 3x3 conv is 4 tiles x 1 008 MFMAs in sequence, and the
// launch has fewer workgroups than the chip has CUs -- the serial slice loop waits on that critical path 150 times per
// step.  Here the eight waves of a workgroup take every eighth (8-channel group, tap) sub-step of the SAME four
// tiles (the critical path is an eighth), stage the activation-free operand by LDS-DMA themselves (no loader waves:
// materialised activations make every chain operand eligible), and at the end park their partial accumulators in LDS
// where wave w sums tile w / 2, row half w % 2 in wave order (deterministic) and runs the fused epilogue of that half
// tile -- reduction and epilogue are spread over all eight waves too.
#include <cstdlib>
#include "conv_common.h"

namespace icm {

template <int EPI>
__device__ __forceinline__ void ks8_store(const ConvDesc& d, const ConvPtrs& P, const f32x16 v, int cot, int hf, int h, int n,
                                          int oy, int ox, bool pvalid) {
  if (hf == 0) store_half_e<EPI, 0>(d, P, v, cot, h, n, oy, ox, pvalid);
  else store_half_e<EPI, 1>(d, P, v, cot, h, n, oy, ox, pvalid);
}

template <int TCO, int TPX>
__global__ __launch_bounds__(512, 2) void conv_ks8_kernel(const ConvDesc d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  static_assert(TCO * TPX == 4, "four output tiles per workgroup: one half tile per wave in the epilogue");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const ConvPtrs P = d.g[blockIdx.y];
  const PatchGeom& pg = d.pg;
  int bid;
  {
    const int nb = gridDim.x, hb = blockIdx.x;
    const int xcd = hb & 7, q = hb >> 3;
    bid = xcd * (nb >> 3) + min(xcd, nb & 7) + q;
  }
  const int cb = bid % d.ncb;
  int pt = bid / d.ncb;
  const int tx_i = pt % d.tiles_x;
  pt /= d.tiles_x;
  const int ty_i = pt % d.tiles_y;
  const int tn_i = pt / d.tiles_y;
  const int ox0 = tx_i << d.lgTW, oy0 = ty_i << d.lgTH, n0 = tn_i << d.lgTI;
  const int iyb = oy0 + d.iy0, ixb = ox0 + d.ix0;   // stride-1 input sampling only (host-checked)
  const int bufsz = d.ckm * 8 * pg.CS;
  const int nchunks = (d.nchunks8 + d.ckm - 1) / d.ckm;
  const long long HWb = (long long)pg.H * pg.W * 4;
  const float* zero = icm_zero_page + lane;
  const int nj = (pg.TIPH * pg.PW + 63) >> 6;

  PlaneMap pm;
  plane_map_init(pm, pg, n0, iyb, ixb, lane);
  const char* xb = reinterpret_cast<const char*>(P.x);
  auto stage = [&](int chunk) {
    const int c0 = chunk * d.ckm * 8;
    const int nch = min(d.ckm, d.nchunks8 - chunk * d.ckm) * 8;
    float* dst = smem + (chunk & 1) * bufsz;
    for (int cl = wave; cl < nch; cl += 8) {
      const int c = c0 + cl;
      const char* base = xb + (long long)c * HWb;
      float* slab = dst + cl * pg.CS;
#pragma unroll
      for (int j = 0; j < ICM_MAXJ; ++j) {
        if (j < nj && pm.loff[j] >= 0) {
          const float* pp = (c < pg.C && pm.goff[j] >= 0) ? reinterpret_cast<const float*>(base + pm.goff[j]) : zero;
          __builtin_amdgcn_global_load_lds(pp, slab + 64 * j, 4, 0, 0);
        }
      }
    }
  };

  const int h = lane >> 5, l31 = lane & 31;
  const int TWm = (1 << d.lgTW) - 1, THm = (1 << d.lgTH) - 1;
  int boff[TPX];
#pragma unroll
  for (int tp = 0; tp < TPX; ++tp) {
    const int p = tp * 32 + l31;
    const int tx = p & TWm, ty = (p >> d.lgTW) & THm, ti = p >> (d.lgTW + d.lgTH);
    boff[tp] = h * pg.CS + ti * pg.PP + ty * pg.PWrow + tx;
  }
  f32x16 acc[TCO][TPX];
#pragma unroll
  for (int a = 0; a < TCO; ++a)
#pragma unroll
    for (int b = 0; b < TPX; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  const char* wbase = reinterpret_cast<const char*>(P.wp);
  const int cot0 = cb * TCO;
  unsigned wl[TCO];
#pragma unroll
  for (int a = 0; a < TCO; ++a) wl[a] = (unsigned)(min(cot0 + a, d.ncot - 1) * 64 + lane) * 16u;
  const long long qstride = (long long)d.ncot * 64 * 16;
  const int Qtot = d.nchunks8 * d.ntaps;
  const int qchunk = d.ckm * d.ntaps;   // sub-steps per staged chunk (<= 64: host-checked)
  int step_lane;
  {
    const int sq = lane / d.ntaps, tq = lane - sq * d.ntaps;
    step_lane = sq * 8 * pg.CS + d.tapoff[tq];
  }
  // this wave's sub-steps: q = wave, wave + 8, ...; weights of step i + 4 are in flight while step i is multiplied
  constexpr int PF = 4;
  f32x4 aq[PF][TCO];
#pragma unroll
  for (int u = 0; u < PF; ++u)
#pragma unroll
    for (int a = 0; a < TCO; ++a)
      aq[u][a] = *reinterpret_cast<const f32x4*>(wbase + (long long)min(wave + 8 * u, Qtot - 1) * qstride + wl[a]);

  stage(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int chunk = 0, qend = min(qchunk, Qtot);
  if (nchunks > 1) stage(1);
  // crossing into the next staged chunk: its DMAs have landed (vmcnt), everybody is done with the buffer it replaces
  auto cross = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ++chunk;
    qend = min((chunk + 1) * qchunk, Qtot);
    if (chunk + 1 < nchunks) stage(chunk + 1);
  };
  for (int q0 = wave; q0 < Qtot; q0 += 8 * PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int q = q0 + 8 * u;
      if (q >= Qtot) break;
      while (q >= qend) cross();
      const float* bp = smem + (chunk & 1) * bufsz + __builtin_amdgcn_readlane(step_lane, q - chunk * qchunk);
      float bv[4][TPX];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int tp = 0; tp < TPX; ++tp) bv[j][tp] = bp[boff[tp] + 2 * j * pg.CS];
      const char* sp = wbase + (long long)min(q + 8 * PF, Qtot - 1) * qstride;
#pragma unroll
      for (int a = 0; a < TCO; ++a) {
#pragma unroll
        for (int tp = 0; tp < TPX; ++tp)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[a][tp] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[u][a][j], bv[j][tp], acc[a][tp], 0, 0, 0);
        aq[u][a] = *reinterpret_cast<const f32x4*>(sp + wl[a]);
      }
    }
  }
  while (chunk + 1 < nchunks) cross();   // waves that ran out of sub-steps early still take part in every barrier
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                        // all sub-steps done: the staging buffers are free for the reduction

  // ---- K-split reduction + epilogue: red[wave][tile][r][lane]; wave w finishes tile w >> 1, rows (w & 1) * 8 .. + 8
  float* red = smem;
#pragma unroll
  for (int a = 0; a < TCO; ++a)
#pragma unroll
    for (int tp = 0; tp < TPX; ++tp)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((wave * 4 + a * TPX + tp) * 16 + r) * 64 + lane] = acc[a][tp][r];
  __syncthreads();
  const int t = wave >> 1, hf = wave & 1;
  f32x16 v;
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = 0.0f;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += red[((k * 4 + t) * 16 + hf * 8 + q) * 64 + lane];
    if (hf == 0) v[q] = s; else v[8 + q] = s;
  }
  const int a_t = t / TPX, tp_t = t % TPX;
  const int cot = cot0 + a_t;
  const int p = tp_t * 32 + l31;
  const int tx = p & TWm, ty = (p >> d.lgTW) & THm, ti = p >> (d.lgTW + d.lgTH);
  const int n = n0 + ti, oyv = oy0 + ty, oxv = ox0 + tx;
  const bool pvalid = (n < pg.N) && (oyv < d.OHv) && (oxv < d.OWv) && cot < d.ncot;
  const int oy = oyv * d.out_sy + d.out_oy, ox = oxv * d.out_sx + d.out_ox;
  switch (d.epi) {
    case ICM_EPI_RES: ks8_store<ICM_EPI_RES>(d, P, v, cot, hf, h, n, oy, ox, pvalid); break;
    case ICM_EPI_RES_GELU: ks8_store<ICM_EPI_RES_GELU>(d, P, v, cot, hf, h, n, oy, ox, pvalid); break;
    case ICM_EPI_MUL_DGELU: ks8_store<ICM_EPI_MUL_DGELU>(d, P, v, cot, hf, h, n, oy, ox, pvalid); break;
    case ICM_EPI_LRP: ks8_store<ICM_EPI_LRP>(d, P, v, cot, hf, h, n, oy, ox, pvalid); break;
    case ICM_EPI_RES_MUL_DGELU: ks8_store<ICM_EPI_RES_MUL_DGELU>(d, P, v, cot, hf, h, n, oy, ox, pvalid); break;
    default: ks8_store<ICM_EPI_NONE>(d, P, v, cot, hf, h, n, oy, ox, pvalid); break;
  }
}

int launch_conv_ks8(const ConvDesc& d, int tco, long long nblk, int ngroups, size_t lds_bytes, hipStream_t stream) {
  void (*fn)(const ConvDesc) = tco == 2 ? conv_ks8_kernel<2, 2> : conv_ks8_kernel<1, 4>;
  if (lds_bytes > 64 * 1024 && !ensure_max_lds(reinterpret_cast<const void*>(fn))) return ICM_ERR_LAUNCH;
  hipLaunchKernelGGL(fn, dim3((unsigned)nblk, ngroups, 1), dim3(512), lds_bytes, stream, d);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}

}  // namespace icm
