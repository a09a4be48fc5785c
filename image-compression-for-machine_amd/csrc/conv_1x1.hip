// Pointwise (1x1, stride 1) convolutions / Linear layers on f32 MFMA for gfx950: the barrier-free direct-operand kernel.
#include <cstdlib>
#include "conv_common.h"

namespace icm {

// ------------------------------------------------------------------------------------------------
// Pointwise (1x1, stride 1) convolutions / Linear layers with many pixels: no LDS, no barriers.
// A 1x1 conv has no tap reuse, so staging the operand through LDS buys nothing: the B fragment of
// v_mfma_f32_32x32x2_f32 (B[k][n = lane & 31], k = lane >> 5) IS a pair of 128-byte row segments of two
// channel planes -- each lane loads its own operand straight from global memory (fully coalesced), DB
// 8-channel chunks ahead of their use; A fragments are the packed weights from L2 as in the kernel above.
// Every wave owns TCO x TPX output tiles of one pixel strip and runs independently (256-thread groups, two
// per CU at <= 256 VGPRs): the epilogue of one wave overlaps the K loop of its SIMD neighbour, which the
// LDS-staged kernel cannot do for K loops this short (K = 96-192: two barrier rounds, prologue and epilogue
// exposed).
template <int TCO, int TPX>
__global__ __launch_bounds__(256, 2) void conv1x1_kernel(const ConvDesc d) {
  constexpr int DB = 4;   // B chunks in flight (HBM latency); A fragments are prefetched two chunks ahead (L2 latency)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const ConvPtrs P = d.g[blockIdx.y];
  int bid;
  {
    const int nb = gridDim.x, hb = blockIdx.x;
    const int xcd = hb & 7, q = hb >> 3;
    bid = xcd * (nb >> 3) + min(xcd, nb & 7) + q;
  }
  const int cb = bid % d.ncb, ptile = bid / d.ncb;
  const int h = lane >> 5, l31 = lane & 31;
  const int HW = d.pg.H * d.pg.W;
  const int NP = d.pg.N * HW;
  const int p_base = (ptile * 4 + wave) * (32 * TPX);
  if (p_base >= NP) return;   // no barriers in this kernel: a wave past the end simply leaves
  unsigned voff[TPX];
  int pn[TPX], phw[TPX];
  bool pv[TPX];
#pragma unroll
  for (int tp = 0; tp < TPX; ++tp) {
    const int p = p_base + tp * 32 + l31;
    pv[tp] = p < NP;
    const int pc = min(p, NP - 1);
    pn[tp] = pc / HW;
    phw[tp] = pc - pn[tp] * HW;
    voff[tp] = (unsigned)(pn[tp] * (int)d.pg.bs + phw[tp] + h * HW) * 4u;
  }
  f32x16 acc[TCO][TPX];
#pragma unroll
  for (int a = 0; a < TCO; ++a)
#pragma unroll
    for (int b = 0; b < TPX; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  const char* xb = reinterpret_cast<const char*>(P.x);       // wave-uniform bases: loads are saddr + 32-bit voffset
  const char* wbase = reinterpret_cast<const char*>(P.wp);
  const long long cstride = (long long)HW * 4;               // bytes per channel plane
  const int cot0 = cb * TCO;
  unsigned wl[TCO];
#pragma unroll
  for (int a = 0; a < TCO; ++a) wl[a] = (unsigned)(min(cot0 + a, d.ncot - 1) * 64 + lane) * 16u;
  const long long qstride = (long long)d.ncot * 64 * 16;
  const int nq = d.nchunks8;
  const int act = d.pg.act;

  float bq[DB][4][TPX];
  f32x4 aq[2][TCO];
  auto loadB = [&](float (&dst)[4][TPX], int q) {
    const char* sb = xb + (long long)min(q, nq - 1) * 8 * cstride;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int tp = 0; tp < TPX; ++tp) dst[j][tp] = *reinterpret_cast<const float*>(sb + 2 * j * cstride + voff[tp]);
  };
#pragma unroll
  for (int a = 0; a < TCO; ++a) {
    aq[0][a] = *reinterpret_cast<const f32x4*>(wbase + wl[a]);
    aq[1][a] = *reinterpret_cast<const f32x4*>(wbase + (long long)min(1, nq - 1) * qstride + wl[a]);
  }
#pragma unroll
  for (int u = 0; u < DB; ++u) loadB(bq[u], u);

  for (int q0 = 0; q0 < nq; q0 += DB) {
#pragma unroll
    for (int u = 0; u < DB; ++u) {
      const int q = q0 + u;
      if (q >= nq) break;   // wave-uniform
      float bv[4][TPX];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int tp = 0; tp < TPX; ++tp) bv[j][tp] = bq[u][j][tp];
      if (act == ICM_ACT_GELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int tp = 0; tp < TPX; ++tp) bv[j][tp] = gelu_f(bv[j][tp]);
      } else if (act == ICM_ACT_SQUARE) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int tp = 0; tp < TPX; ++tp) bv[j][tp] *= bv[j][tp];
      }
      loadB(bq[u], q + DB);
      const char* sp = wbase + (long long)min(q + 2, nq - 1) * qstride;
#pragma unroll
      for (int a = 0; a < TCO; ++a) {
#pragma unroll
        for (int tp = 0; tp < TPX; ++tp)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[a][tp] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[u & 1][a][j], bv[j][tp], acc[a][tp], 0, 0, 0);
        aq[u & 1][a] = *reinterpret_cast<const f32x4*>(sp + wl[a]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  int zero[TPX];
#pragma unroll
  for (int tp = 0; tp < TPX; ++tp) zero[tp] = 0;
  // the pipelined two-operand epilogue holds two sets of operand registers: with 6 accumulator tiles (TCO 3 x TPX 2)
  // it spilled 40 VGPRs (164 B scratch per lane); that instantiation takes the batched half-tile form
  epilogue_dispatch<TCO, TPX, !(TCO == 3 && TPX == 2)>(d, P, acc, cot0, h, pn, zero, phw, pv);
}

struct Cfg1x1 {
  int tco, tpx;
  void (*fn)(const ConvDesc);
  float eff;   // relative throughput of a wave of this tile shape when the chip is full (loads per MFMA; measured,
               // profiles/r03_conv1x1_ablation.txt)
};
static const Cfg1x1 kCfgs1x1[] = {
    {6, 1, conv1x1_kernel<6, 1>, 1.0f},  {5, 1, conv1x1_kernel<5, 1>, 1.0f}, {4, 1, conv1x1_kernel<4, 1>, 0.97f},
    {3, 2, conv1x1_kernel<3, 2>, 1.0f},  {2, 2, conv1x1_kernel<2, 2>, 0.7f}, {1, 2, conv1x1_kernel<1, 2>, 0.6f},
    {3, 1, conv1x1_kernel<3, 1>, 0.95f}, {2, 1, conv1x1_kernel<2, 1>, 0.6f},   // more, smaller wave tiles
};

int run_conv1x1(const icm_conv_args* arr, int ngroups, long long wp_off, int g_force_1x1, hipStream_t stream) {
  const icm_conv_args& a = arr[0];
  if (g_force_1x1 == 0) return -1;
  if (a.KH != 1 || a.KW != 1 || a.stride != 1 || a.pad != 0 || a.pixel_shuffle || (a.Cin % 8) != 0) return -1;
  if (a.OH != a.H || a.OW != a.W) return -1;
  const int ncot = cdiv(a.Cout, 32);
  const long long NP = (long long)a.N * a.H * a.W;
  // Tile shape: no padded co tiles if avoidable; among those, the shape that keeps the chip busiest.  Every wave is
  // independent, so a launch wants >= 2 waves per SIMD (2 048): the stf Linear layers on 32x32 maps (16 384 pixels) give
  // the 192-co x 32-px tile only 512 waves -- half the SIMDs idle, the other half one wave each (768 -> 192: 79 us;
  // with 96-co tiles 59 us).  score = min(waves, 2 048) x the shape's full-chip efficiency; ties: the earlier (larger).
  int best = -1, best_pad = 1 << 30;
  const int ncfg = (int)(sizeof(kCfgs1x1) / sizeof(kCfgs1x1[0]));
  for (int i = 0; i < ncfg; ++i) best_pad = std::min(best_pad, cdiv(ncot, kCfgs1x1[i].tco) * kCfgs1x1[i].tco);
  float best_score = -1.0f;
  for (int i = 0; i < ncfg; ++i) {
    const Cfg1x1& k = kCfgs1x1[i];
    if (cdiv(ncot, k.tco) * k.tco != best_pad) continue;
    const long long waves = ((NP + 32 * k.tpx - 1) / (32 * k.tpx)) * cdiv(ncot, k.tco) * ngroups;
    const float score = (float)std::min<long long>(waves, 2048) * k.eff;
    if (score > best_score) {
      best_score = score;
      best = i;
    }
  }
  {
    static const int force = getenv("ICM_1X1_CFG") ? atoi(getenv("ICM_1X1_CFG")) : -1;   // measurement only
    if (force >= 0 && force < (int)(sizeof(kCfgs1x1) / sizeof(kCfgs1x1[0]))) best = force;
  }
  const Cfg1x1& c = kCfgs1x1[best];
  const int ncb = cdiv(ncot, c.tco);
  const long long strips = (NP + 32 * c.tpx - 1) / (32 * c.tpx);
  // independent waves need >= one wave per SIMD to beat the split-K / co-resident tilings of the staged kernel
  static const long long kMinWaves = getenv("ICM_1X1_MIN_WAVES") ? atoll(getenv("ICM_1X1_MIN_WAVES")) : 1024;
  // (short contractions pay less for a half-empty chip than the staged kernel pays for its barriers: 320 -> 160 at 16x16
  //  x 2 members, 640 waves: 24.9 us here against 29.7 us staged; 1 536 -> 384 on the same map is the other way round)
  static const int kShortK = getenv("ICM_1X1_SHORT_K") ? atoi(getenv("ICM_1X1_SHORT_K")) : 384;
  const long long need = a.Cin <= kShortK ? kMinWaves / 2 : kMinWaves;
  if (g_force_1x1 < 0 && strips * ncb * ngroups < need) return -1;
  ConvDesc d;
  for (int gi = 0; gi < ICM_MAX_GROUPS; ++gi) {
    const icm_conv_args& s = arr[gi < ngroups ? gi : 0];
    d.g[gi].x = s.x;
    d.g[gi].wp = s.wp + wp_off;
    d.g[gi].bias = s.bias;
    d.g[gi].y = s.y;
    d.g[gi].res = s.res;
    d.g[gi].aux = s.aux;
    d.g[gi].aux2 = s.aux2;
    d.g[gi].y2 = s.y2;
  }
  d.y_bs = a.y_bs; d.res_bs = a.res_bs; d.aux_bs = a.aux_bs; d.aux2_bs = a.aux2_bs; d.y2_bs = a.y2_bs;
  d.pg = PatchGeom{};
  d.pg.H = a.H; d.pg.W = a.W; d.pg.N = a.N; d.pg.C = a.Cin; d.pg.act = a.pro_act; d.pg.bs = a.x_bs;
  d.Cout = a.Cout;
  d.ps2 = 0;
  d.OHf = a.H; d.OWf = a.W; d.OHv = a.H; d.OWv = a.W;
  d.out_sy = d.out_sx = 1; d.out_oy = d.out_ox = 0; d.iy0 = d.ix0 = 0;
  d.ntaps = 1;
  d.lgTW = d.lgTH = d.lgTI = 0;
  d.tiles_x = d.tiles_y = d.tiles_n = 1;
  d.ncot = ncot; d.nchunks8 = a.Cin / 8; d.ckm = 1; d.ncb = ncb;
  d.epi = a.epi; d.accum = a.accum;
  for (int t = 0; t < ICM_MAX_TAPS; ++t) d.tapoff[t] = 0;
  const long long nblk = ((strips + 3) / 4) * ncb;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return ICM_ERR_ARG;
  hipLaunchKernelGGL(c.fn, dim3((unsigned)nblk, ngroups, 1), dim3(256), 0, stream, d);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}

}  // namespace icm
