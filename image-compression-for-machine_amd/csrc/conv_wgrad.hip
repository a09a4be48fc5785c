// Weight-gradient kernel of the convolution family (f32 MFMA, gfx950).
//
//   dW[a][b][t] = sum_{n,p} actS(gs[n,a,p]) * actB(gb[n,b,p*stride - pad + t])        (+ optional db[a] = sum gs)
//
// GEMM view: M = a (channels of the small-grid tensor), N = b (channels of the big-grid tensor), K = pixels.
// Workgroup = 512 threads: 4 MFMA waves + 4 loader waves (same split as conv_igemm.hip).  Per pixel tile
// (64 small-grid pixels) the loaders stage the small-grid tile as [a][pixel] (row stride 65: conflict-free A
// fragments) and the big-grid halo patch as [b][patch] (odd plane stride: conflict-free B fragments) into the
// other LDS buffer while the MFMA waves sweep the current one: every (tap, b-tile) pair re-reads the same
// patch at a different offset, so one staging serves up to 16 taps.  Accumulators ([TA a-tiles] x [pairs of
// this wave]) stay in registers across the workgroup's whole share of the pixel tiles; partial sums go to
// [split][tap][a][b] slabs with 128-B coalesced plain stores and are reduced + transposed to the canonical
// [a][b][kh][kw] layout by a second kernel: bitwise reproducible, no float atomics (Guideline 12).
// The bias gradient (row sums of the small-grid tile) is accumulated by the loader waves for free.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include "icm_common.h"

namespace icm {

#define WG_MAX_TAPS 32

// Winograd F(2x2, 3x3) form (wgrad_wino.hip): writes dU slabs [split][16][a][b] (+ bias partials [split][a])
int wgrad_wino_plan(const icm_wgrad_args& a, int nproblems, int* nsplit_out, int* nchunks_out);
int launch_wgrad_wino(const icm_wgrad_args* arr, int n, int nsplit, int nchunks, float* const* ws, float* const* dbias_ws,
                      hipStream_t stream);

#define WG_MAXG 32
struct WgPtrs {
  const float* gs;
  const float* gb;
  float* ws;
  float* dbias_ws;   // [nsplit][Ca] or null
};
struct WgDesc {
  WgPtrs g[WG_MAXG];   // problems of identical geometry: blockIdx.y selects one (deferred, batched wgrads)
  long long gs_bs;
  PatchGeom pg;      // big-grid tensor + patch layout (identity colmap)
  int Ca, OH, OW, act_s;
  int S, pad, ntaps, tpg;   // taps per tap-group
  int lgTW, lgTH, lgTI, lgNPX, gs_vec4;
  int tiles_x, tiles_y, tiles_n, ntiles, nsplit;
  int natile, nbtile, ngroups;
  int xcd_order;               // 1: XCD-aware workgroup order
  int po_h;                    // patch offset of pixel 2kp+1 relative to pixel 2kp
  int tapoff[WG_MAX_TAPS];
  int pe[32];                  // patch offset of pixel 2kp (wave-uniform: scalar loads)
};

// Loader waves (4 of the 8 waves of a workgroup): per pixel tile stage the small-grid tile as [a][pixel] (row stride
// npx + 1) and the big-grid halo patch as [b][patch] into the LDS buffer the MFMA waves are not reading, one tile
// ahead; accumulate the fused bias gradient (row sums of the small-grid tile) on the way.
template <int TA, int TB>
__device__ __forceinline__ void wgrad_loader(const WgDesc& d, const WgPtrs& G, float* smem, int tid, int lane, int wave,
                                             int a0, int b0, int split, int niter, int bt, int tg) {
  const PatchGeom& pg = d.pg;
  const int npx = 1 << d.lgNPX, grow = npx + 1;
  const int gs_sz = TA * 32 * grow, gb_sz = TB * 32 * pg.CS;
  const int bufsz = gs_sz + gb_sz;
  const int TWm = (1 << d.lgTW) - 1, THm = (1 << d.lgTH) - 1;
  __builtin_amdgcn_s_setprio(3);   // loaders are latency-critical and issue little: let them win arbitration
  const int ltid = tid - 256;
  // fused bias gradient = row sums of the staged small-grid tiles: loader thread r adds row r of the PREVIOUS tile
  // (complete since the last barrier; the MFMA waves only read it) -- one register, fixed summation order
  const bool do_bias = G.dbias_ws != nullptr && bt == 0 && tg == 0 && ltid < TA * 32;
  float bacc = 0.0f;
  const int OHW = d.OH * d.OW;
  const int p = ltid & (npx - 1), a_sub = ltid >> d.lgNPX, a_step = 256 >> d.lgNPX;
  const int tx = p & TWm, ty = (p >> d.lgTW) & THm, ti = p >> (d.lgTW + d.lgTH);
  for (int it = 0; it <= niter; ++it) {
    if (do_bias && it > 0) {
      const float* row = smem + ((it - 1) & 1) * bufsz + ltid * grow;
      float s0 = 0.0f, s1 = 0.0f;
      for (int j = 0; j < npx; j += 2) { s0 += row[j]; s1 += row[j + 1]; }
      bacc += s0 + s1;
    }
    if (it < niter) {
      int q = split + it * d.nsplit;
      const int tx_i = q % d.tiles_x; q /= d.tiles_x;
      const int ty_i = q % d.tiles_y;
      const int tn_i = q / d.tiles_y;
      const int ox0 = tx_i << d.lgTW, oy0 = ty_i << d.lgTH, n0 = tn_i << d.lgTI;
      float* gsT = smem + (it & 1) * bufsz;
      float* gbP = gsT + gs_sz;
      const int n = n0 + ti, oy = oy0 + ty, ox = ox0 + tx;
      const bool pv = n < pg.N && oy < d.OH && ox < d.OW;
      PlaneMap pm;
      if (pg.dma) {   // big-grid patch by LDS-DMA first: it flies while the small-grid tile goes through registers
        plane_map_init(pm, pg, n0, oy0 * d.S - d.pad, ox0 * d.S - d.pad, lane);
        stage_planes_dma(G.gb, pm, pg, b0, TB * 32, gbP, __builtin_amdgcn_readfirstlane(wave) - 4, lane);
      }
      if (d.gs_vec4) {
        // 16-byte loads: lane handles 4 consecutive pixels of channel (ltid >> lg4) + rows4 * k; npx / 4 lanes per
        // channel row (16 for 64-pixel tiles, 8 for 32-pixel tiles), rows4 = 256 / (npx / 4) channels per pass
        const int lg4 = d.lgNPX - 2, rows4 = 256 >> lg4;
        const int p4 = (ltid & ((1 << lg4) - 1)) << 2;
        const int tx4 = p4 & TWm, ty4 = (p4 >> d.lgTW) & THm, ti4 = p4 >> (d.lgTW + d.lgTH);
        const int n4 = n0 + ti4, oy4 = oy0 + ty4, ox4 = ox0 + tx4;
        const bool pv4 = n4 < pg.N && oy4 < d.OH && ox4 < d.OW;
        const float* src4 = G.gs + (long long)n4 * d.gs_bs + oy4 * d.OW + ox4;
        const int a_sub4 = ltid >> lg4;
        // batches of 6 loads in flight (a full unroll of the 12 / 48 loads of the 192-row tiles spills registers)
#pragma unroll 1
        for (int k0 = 0; k0 < TA * 2; k0 += 6) {
          f32x4 v[6];
#pragma unroll
          for (int u = 0; u < 6; ++u) {
            const int a = a_sub4 + rows4 * (k0 + u), ca = a0 + a;
            v[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (k0 + u < TA * 2 && a < TA * 32 && pv4 && ca < d.Ca)
              v[u] = *reinterpret_cast<const f32x4*>(src4 + (long long)ca * OHW);
          }
#pragma unroll
          for (int u = 0; u < 6; ++u) {
            const int a = a_sub4 + rows4 * (k0 + u);
            if (k0 + u < TA * 2 && a < TA * 32) {
              float* dst = gsT + a * grow + p4;
#pragma unroll
              for (int e = 0; e < 4; ++e) dst[e] = apply_act(v[u][e], d.act_s);
            }
          }
        }
      } else {
      const float* src = G.gs + (long long)n * d.gs_bs + oy * d.OW + ox;
#pragma unroll 1
      for (int k0 = 0; k0 < TA * 8; k0 += 16) {   // 16 loads in flight per lane
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int a = a_sub + a_step * (k0 + u), ca = a0 + a;
          v[u] = (a < TA * 32 && pv && ca < d.Ca) ? src[(long long)ca * OHW] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int a = a_sub + a_step * (k0 + u);
          if (a < TA * 32) gsT[a * grow + p] = apply_act(v[u], d.act_s);
        }
      }
      }
      if (pg.dma) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the barrier below publishes this stage: DMAs must have landed
        if (pg.act != ICM_ACT_NONE) act_planes_inplace(pg, TB * 32, gbP, __builtin_amdgcn_readfirstlane(wave) - 4, lane);
      } else if (pg.vec4) {
        plane_map_init_v4(pm, pg, n0, oy0 * d.S - d.pad, ox0 * d.S - d.pad, lane);
        if (pg.pipe) stage_planes_v4<8, true>(G.gb, pm, pg, b0, TB * 32, gbP, __builtin_amdgcn_readfirstlane(wave) - 4);
        else stage_planes_v4<8, false>(G.gb, pm, pg, b0, TB * 32, gbP, __builtin_amdgcn_readfirstlane(wave) - 4);
      } else {
        plane_map_init(pm, pg, n0, oy0 * d.S - d.pad, ox0 * d.S - d.pad, lane);
        if (pg.pipe) stage_planes<12, true>(G.gb, pm, pg, b0, TB * 32, gbP, __builtin_amdgcn_readfirstlane(wave) - 4);
        else stage_planes<12, false>(G.gb, pm, pg, b0, TB * 32, gbP, __builtin_amdgcn_readfirstlane(wave) - 4);
      }
    }
    __syncthreads();
  }
  if (do_bias && a0 + ltid < d.Ca) G.dbias_ws[(long long)split * d.Ca + a0 + ltid] = bacc;
}

// ---- single-tap-group kernel with 3 x 3 accumulator tiles per MFMA wave (1x1 problems) ----------------------------
// Workgroup tile = (TA x TB) 32-tiles of (a, b); the four MFMA waves form a WA x WB x WK grid: wave (wa, wb, wk) holds
// the 3 x 3 tiles [3 wa, 3 wa + 3) x [3 wb, 3 wb + 3) and takes the pixel pairs kp = wk (mod WK); the WK partial sums
// are folded through LDS at the end.  96-wide tiles fit the codec's channel counts (96 / 192 / 576; 160 / 320) far
// better than 128 x 128, and the loaders are latency-bound (a fixed number of bytes in flight per CU), so the FLOPs
// per staged byte decide the speed: <6,6> (192 x 192 per workgroup) moves half the bytes per FLOP of <3,3>.
//   <6,6,2,2>  WK = 1   192 x 192   (GDN gamma, proj / conv1x1 of the dim-192 gates, qkv)
//   <3,6,1,2>  WK = 2    96 x 192   <6,3,2,1>  WK = 2   192 x 96   (ResidualUnit 1x1s, thin-channel ends)
//   <3,3,1,1>  WK = 4    96 x  96
//   <3,3,1,1,TAPW>  96 x 96 x 4 taps: the four waves take four TAPS of a multi-tap (3x3) problem instead of four
//                  pixel-pair subsets -- one staging of the patch serves four taps, and 96-wide tiles fit the 96 / 192
//                  channel layers that the 64-wide tiles of the general kernel cover at 56 %
template <int TA, int TB, int WA, int WB, bool TAPW = false>
__global__ __launch_bounds__(512, 2) void wgrad_t33_kernel(const WgDesc d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  static_assert(TA == 3 * WA && TB == 3 * WB && (WA * WB == 1 || WA * WB == 2 || WA * WB == 4), "3 x 3 tiles per wave");
  static_assert(!TAPW || WA * WB == 1, "tap-per-wave mode: every wave holds the whole 96 x 96 block");
  constexpr int WK = TAPW ? 1 : 4 / (WA * WB);
  const PatchGeom& pg = d.pg;
  const int npx = 1 << d.lgNPX, grow = npx + 1;
  const int gs_sz = TA * 32 * grow, gb_sz = TB * 32 * pg.CS;
  const int bufsz = gs_sz + gb_sz;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const WgPtrs G = d.g[blockIdx.y];
  int bid = blockIdx.x;
  if (d.xcd_order) {
    const int nb = gridDim.x, hb = blockIdx.x;
    const int xcd = hb & 7, q = hb >> 3;
    bid = xcd * (nb >> 3) + min(xcd, nb & 7) + q;
  }
  const int at = bid % d.natile; bid /= d.natile;
  const int bt = bid % d.nbtile; bid /= d.nbtile;
  const int tg = bid % d.ngroups;
  const int split = bid / d.ngroups;
  const int a0 = at * TA * 32, b0 = bt * TB * 32, t0 = tg * d.tpg;
  const int niter = (d.ntiles - split + d.nsplit - 1) / d.nsplit;
  if (wave >= 4) {
    wgrad_loader<TA, TB>(d, G, smem, tid, lane, wave, a0, b0, split, niter, bt, tg);
    return;
  }
  const int h = lane >> 5, l31 = lane & 31;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int wk = TAPW ? 0 : wave_u % WK, wsp = TAPW ? 0 : wave_u / WK;
  const int wa = wsp / WB, wb = wsp % WB;
  // tap of this wave: t0 (all waves) or t0 + wave (TAPW; waves past the last tap redo it and do not store)
  const int nt_g = min(d.tpg, d.ntaps - t0);
  const int mytap = TAPW ? t0 + min(wave_u, nt_g - 1) : t0;
  const bool tap_live = !TAPW || wave_u < nt_g;
  int boffs[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) boffs[u] = ((wb * 3 + u) * 32 + l31) * pg.CS + d.tapoff[mytap] + h * d.po_h;
  f32x16 acc[9];
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  const int pe_lane = d.pe[lane & 31];
  __syncthreads();  // tile 0 staged
  const int nkp = npx >> 1;   // 16 / 32 pixel pairs per tile: every wave takes nkp / WK of them (an even number)
  for (int it = 0; it < niter; ++it) {
    const float* gsT = smem + (it & 1) * bufsz;
    const float* gbP = gsT + gs_sz;
    const float* arow = gsT + (wa * 3 * 32 + l31) * grow + h;
    float avA[3], bvA[3], avB[3], bvB[3];
    auto fetch = [&](float (&av)[3], float (&bv)[3], int kn) {
      const int po = __builtin_amdgcn_readlane(pe_lane, kn);
#pragma unroll
      for (int u = 0; u < 3; ++u) av[u] = arow[u * 32 * grow + 2 * kn];
#pragma unroll
      for (int u = 0; u < 3; ++u) bv[u] = gbP[boffs[u] + po];
    };
    auto mma = [&](const float (&av)[3], const float (&bv)[3]) {
#pragma unroll
      for (int ta = 0; ta < 3; ++ta)
#pragma unroll
        for (int tb = 0; tb < 3; ++tb)
          acc[ta * 3 + tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ta], bv[tb], acc[ta * 3 + tb], 0, 0, 0);
    };
    fetch(avA, bvA, wk);
    for (int kp = wk; kp < nkp; kp += 2 * WK) {
      fetch(avB, bvB, min(kp + WK, nkp - 1));
      mma(avA, bvA);
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      fetch(avA, bvA, min(kp + 2 * WK, nkp - 1));
      if (kp + WK < nkp) mma(avB, bvB);
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    }
    __syncthreads();
  }
  if constexpr (WK > 1) {
    // the WK waves of a (wa, wb) group hold partial sums of the same 9 tiles: fold them through LDS (free after the
    // last tile barrier; the loaders have left, ended waves do not take part in s_barrier) so that the workgroup
    // writes ONE slab.  At most two accumulator sets per group are parked at a time (LDS budget).
    constexpr int SET = 9 * 16 * 64;               // floats of one wave's accumulators
    constexpr int SLOTS = (WK - 1) < 2 ? (WK - 1) : 2;
    float* red = smem + wsp * SLOTS * SET;
#pragma unroll
    for (int r0 = 1; r0 < WK; r0 += SLOTS) {
      if (wk >= r0 && wk < r0 + SLOTS) {
        float* dst = red + (wk - r0) * SET;
#pragma unroll
        for (int i = 0; i < 9; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[(i * 16 + r) * 64 + lane] = acc[i][r];
      }
      __syncthreads();
      if (wk == 0) {
        for (int k = 0; k < SLOTS && r0 + k < WK; ++k)
#pragma unroll
          for (int i = 0; i < 9; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] += red[k * SET + (i * 16 + r) * 64 + lane];
      }
      __syncthreads();
    }
    if (wk != 0) return;
  }
  if (!tap_live) return;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int ta = wa * 3 + i / 3, tb = wb * 3 + i % 3;
    const int b = b0 + tb * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int a = a0 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (a < d.Ca && b < pg.C)
        G.ws[(((long long)split * d.ntaps + mytap) * d.Ca + a) * pg.C + b] = acc[i][r];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- 3x3 stride-1 problems with activation-free operands: all NINE taps from one staging, DMA-only loaders ----------
// The tap-group kernels above re-stage every pixel tile once per group of four taps (9 taps = 4 + 4 + 1: three stagings,
// the last with three of four waves idle), and their loader waves are latency-bound.  Here the workgroup is EIGHT
// MFMA waves and no loader wave: both operands are staged by LDS-DMA (global_load_lds_dword: no VGPRs, every element
// of the next tile in flight while this one is multiplied), wave w owns tap w of all TA x TB tiles of the (a, b) block
// and the tiles of the ninth tap are dealt round-robin (tile e -> wave e mod 8) -- 10-11 accumulator tiles per wave
// for the 96 x 96 block, 4-5 for 64 x 64.  One staging serves all 81 (36) tile-taps: a third of the staged bytes per
// FLOP of the tap-group kernels.  The bias gradient (row sums of the small-grid tile) comes from the A fragments wave 0
// reads anyway.  Needs activation-free operands (materialised activations) and the linear patch layout.
// TPW = own taps per wave: 1 for 3x3 (8 + 1 taps), 3 for 5x5 (24 + 1 taps: wave w owns taps w, w + 8, w + 16; the A
// fragments are shared by all of a wave's taps, so a 96 x 32 block costs 3 A + 3-4 B reads per 9-10 MFMAs).
template <int TA, int TB, int TPW = 1>
__global__ __launch_bounds__(512, 2) void wgrad_tap9_kernel(const WgDesc d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NT = TA * TB;        // tiles per tap
  constexpr int NX = (NT + 7) / 8;   // tiles of the last tap per wave (at most)
  constexpr int TLAST = 8 * TPW;     // the tap whose tiles are dealt round-robin
  const PatchGeom& pg = d.pg;
  const int npx = 1 << d.lgNPX, grow = npx + 1;
  const int gs_sz = TA * 32 * grow, gb_sz = TB * 32 * pg.CS;
  const int bufsz = gs_sz + gb_sz;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const WgPtrs G = d.g[blockIdx.y];
  int bid = blockIdx.x;
  if (d.xcd_order) {
    const int nb = gridDim.x, hb = blockIdx.x;
    const int xcd = hb & 7, q = hb >> 3;
    bid = xcd * (nb >> 3) + min(xcd, nb & 7) + q;
  }
  const int at = bid % d.natile; bid /= d.natile;
  const int bt = bid % d.nbtile;
  const int split = bid / d.nbtile;
  const int a0 = at * TA * 32, b0 = bt * TB * 32;
  const int niter = (d.ntiles - split + d.nsplit - 1) / d.nsplit;
  const int TWm = (1 << d.lgTW) - 1, THm = (1 << d.lgTH) - 1;
  const int OHW = d.OH * d.OW;
  const long long HWb = (long long)pg.H * pg.W * 4;
  const float* zero = icm_zero_page + lane;
  const int nj = (pg.TIPH * pg.PW + 63) >> 6;

  // LDS-DMA staging of pixel tile `it` of this workgroup's share into buffer (it & 1)
  auto stage = [&](int it) {
    int q = split + it * d.nsplit;
    const int tx_i = q % d.tiles_x; q /= d.tiles_x;
    const int ty_i = q % d.tiles_y;
    const int tn_i = q / d.tiles_y;
    const int ox0 = tx_i << d.lgTW, oy0 = ty_i << d.lgTH, n0 = tn_i << d.lgTI;
    float* gsT = smem + (it & 1) * bufsz;
    float* gbP = gsT + gs_sz;
    {   // small-grid tile [a][pixel]: one channel row per DMA instruction, lane = pixel of the tile
      const int tx = lane & TWm, ty = (lane >> d.lgTW) & THm, ti = lane >> (d.lgTW + d.lgTH);
      const int n = n0 + ti, oy = oy0 + ty, ox = ox0 + tx;
      const bool pv = n < pg.N && oy < d.OH && ox < d.OW;
      const float* base = G.gs + ((long long)n * d.gs_bs + oy * d.OW + ox);
      if (lane < npx) {
        for (int r = wave; r < TA * 32; r += 8) {
          const int a = a0 + r;
          const float* pp = (pv && a < d.Ca) ? base + (long long)a * OHW : zero;
          __builtin_amdgcn_global_load_lds(pp, gsT + r * grow, 4, 0, 0);
        }
      }
    }
    PlaneMap pm;
    plane_map_init(pm, pg, n0, oy0 * d.S - d.pad, ox0 * d.S - d.pad, lane);
    const char* gbb = reinterpret_cast<const char*>(G.gb);
    for (int cl = wave; cl < TB * 32; cl += 8) {
      const int c = b0 + cl;
      const char* base = gbb + (long long)c * HWb;
      float* slab = gbP + cl * pg.CS;
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
  int boffs[TPW][TB];
#pragma unroll
  for (int j = 0; j < TPW; ++j)
#pragma unroll
    for (int u = 0; u < TB; ++u) boffs[j][u] = (u * 32 + l31) * pg.CS + d.tapoff[wave + 8 * j] + h * d.po_h;
  int xoff[NX], xrow[NX];
  int nx_w = 0;
#pragma unroll
  for (int x = 0; x < NX; ++x) {
    const int e = wave + 8 * x;
    if (e < NT) nx_w = x + 1;
    const int ee = e < NT ? e : 0;
    xoff[x] = ((ee % TB) * 32 + l31) * pg.CS + d.tapoff[TLAST] + h * d.po_h;
    xrow[x] = (ee / TB) * 32 * grow;
  }
  f32x16 acc[TPW * NT + NX];
#pragma unroll
  for (int i = 0; i < TPW * NT + NX; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  const bool do_bias = G.dbias_ws != nullptr && bt == 0 && wave == 0;
  float bsum[TA];
#pragma unroll
  for (int u = 0; u < TA; ++u) bsum[u] = 0.0f;
  const int pe_lane = d.pe[lane & 31];
  const int nkp = npx >> 1;

  stage(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int it = 0; it < niter; ++it) {
    if (it + 1 < niter) stage(it + 1);   // flies while this tile is multiplied
    const float* gsT = smem + (it & 1) * bufsz;
    const float* gbP = gsT + gs_sz;
    const float* arow = gsT + l31 * grow + h;
    float avA[TA], bvA[TPW][TB], axA[NX], bxA[NX], avB[TA], bvB[TPW][TB], axB[NX], bxB[NX];
    auto fetch = [&](float (&av)[TA], float (&bv)[TPW][TB], float (&ax)[NX], float (&bx)[NX], int kn) {
      const int po = __builtin_amdgcn_readlane(pe_lane, kn);
#pragma unroll
      for (int u = 0; u < TA; ++u) av[u] = arow[u * 32 * grow + 2 * kn];
#pragma unroll
      for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int u = 0; u < TB; ++u) bv[j][u] = gbP[boffs[j][u] + po];
#pragma unroll
      for (int x = 0; x < NX; ++x) {
        ax[x] = arow[xrow[x] + 2 * kn];
        bx[x] = gbP[xoff[x] + po];
      }
    };
    auto mma = [&](const float (&av)[TA], const float (&bv)[TPW][TB], const float (&ax)[NX], const float (&bx)[NX]) {
#pragma unroll
      for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int ta = 0; ta < TA; ++ta)
#pragma unroll
          for (int tb = 0; tb < TB; ++tb)
            acc[(j * TA + ta) * TB + tb] =
                __builtin_amdgcn_mfma_f32_32x32x2f32(av[ta], bv[j][tb], acc[(j * TA + ta) * TB + tb], 0, 0, 0);
#pragma unroll
      for (int x = 0; x < NX; ++x)
        if (x < nx_w) acc[TPW * NT + x] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[x], bx[x], acc[TPW * NT + x], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < TA; ++u) bsum[u] += av[u];
    };
    fetch(avA, bvA, axA, bxA, 0);
    for (int kp = 0; kp < nkp; kp += 2) {
      fetch(avB, bvB, axB, bxB, kp + 1);
      mma(avA, bvA, axA, bxA);
      fetch(avA, bvA, axA, bxA, min(kp + 2, nkp - 1));
      mma(avB, bvB, axB, bxB);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next tile's DMAs have landed before the barrier publishes them
    __syncthreads();
  }
  if (do_bias) {
#pragma unroll
    for (int u = 0; u < TA; ++u) {
      const float sfull = bsum[u] + __shfl_xor(bsum[u], 32, 64);   // pixels 2kp (h = 0) + pixels 2kp+1 (h = 1)
      const int a = a0 + u * 32 + l31;
      if (h == 0 && a < d.Ca) G.dbias_ws[(long long)split * d.Ca + a] = sfull;
    }
  }
#pragma unroll
  for (int i = 0; i < TPW * NT + NX; ++i) {
    int tap, ta, tb;
    if (i < TPW * NT) { tap = wave + 8 * (i / NT); ta = (i % NT) / TB; tb = i % TB; }
    else {
      const int e = wave + 8 * (i - TPW * NT);
      if (e >= NT) continue;
      tap = TLAST; ta = e / TB; tb = e % TB;
    }
    const int b = b0 + tb * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int a = a0 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (a < d.Ca && b < pg.C)
        G.ws[(((long long)split * d.ntaps + tap) * d.Ca + a) * pg.C + b] = acc[i][r];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- 1x1 problems with activation-free operands: eight MFMA waves, DMA-only staging ---------------------------------
// Same idea as the tap kernels above for problems without taps: no loader waves; the (TA x TB) block is covered by
// WA x WB groups of 3 x 3 tiles and the 8 / (WA * WB) waves of a group split the pixel pairs (kp = wk mod WK) and fold
// their partial sums through LDS at the end (fixed order).  The bias gradient comes from the A fragments of the wb == 0
// waves, folded over the K split the same way.
// V4 (64-pixel tiles, OW % 4 == 0): 16-byte DMA, FOUR channel rows per instruction -- (TA + TB) * 8 staging instructions
// per tile instead of (TA + TB) * 32.  The address path charges per instruction, not per byte, and at one
// 256-byte row per instruction the 288 DMAs of a 192 x 96 block took as long as its 576 MFMAs per SIMD (matrix pipe 38 %
// busy).  A 16-byte DMA writes LDS linearly (base + 16 * lane), so rows are 64 floats apart with no padding; the bank
// spread comes from rotating row r by 4 * (r mod 16) floats instead, which the per-lane GLOBAL address absorbs for free
// (lane (s, q) of an instruction loads pixels 4 (q - r mod 16) mod 64 ... of row r = 4 i + s) and costs the fragment
// reads one add + and per pixel pair (element (r, p) sits at 64 r + ((p + 4 (r mod 16)) mod 64); two-way conflicts).
template <int TA, int TB, int WA, int WB, bool V4>
__global__ __launch_bounds__(512, 2) void wgrad_dma1_kernel(const WgDesc d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  static_assert(TA == 3 * WA && TB == 3 * WB && (WA * WB == 1 || WA * WB == 2 || WA * WB == 4), "3 x 3 tiles per wave");
  constexpr int WK = 8 / (WA * WB);
  const PatchGeom& pg = d.pg;
  const int npx = 1 << d.lgNPX, grow = V4 ? 64 : npx + 1, brow = V4 ? 64 : pg.CS;
  const int gs_sz = TA * 32 * grow, gb_sz = TB * 32 * brow;
  const int bufsz = gs_sz + gb_sz;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const WgPtrs G = d.g[blockIdx.y];
  int bid = blockIdx.x;
  if (d.xcd_order) {
    const int nb = gridDim.x, hb = blockIdx.x;
    const int xcd = hb & 7, q = hb >> 3;
    bid = xcd * (nb >> 3) + min(xcd, nb & 7) + q;
  }
  const int at = bid % d.natile; bid /= d.natile;
  const int bt = bid % d.nbtile;
  const int split = bid / d.nbtile;
  const int a0 = at * TA * 32, b0 = bt * TB * 32;
  const int niter = (d.ntiles - split + d.nsplit - 1) / d.nsplit;
  const int TWm = (1 << d.lgTW) - 1, THm = (1 << d.lgTH) - 1;
  const int OHW = d.OH * d.OW;
  const long long HW = (long long)pg.H * pg.W;
  const float* zero = icm_zero_page + lane;
  const float* zero4 = icm_zero_page + 4 * lane;

  // both operands of a 1x1 stride-1 problem are [channel][pixel of the tile]: one channel row per DMA instruction
  auto stage = [&](int it) {
    int q = split + it * d.nsplit;
    const int tx_i = q % d.tiles_x; q /= d.tiles_x;
    const int ty_i = q % d.tiles_y;
    const int tn_i = q / d.tiles_y;
    const int ox0 = tx_i << d.lgTW, oy0 = ty_i << d.lgTH, n0 = tn_i << d.lgTI;
    float* gsT = smem + (it & 1) * bufsz;
    float* gbP = gsT + gs_sz;
    const int tx = lane & TWm, ty = (lane >> d.lgTW) & THm, ti = lane >> (d.lgTW + d.lgTH);
    const int n = n0 + ti, oy = oy0 + ty, ox = ox0 + tx;
    const bool pv = n < pg.N && oy < d.OH && ox < d.OW;
    const float* abase = G.gs + ((long long)n * d.gs_bs + oy * d.OW + ox);
    const float* bbase = G.gb + ((long long)n * pg.bs + oy * pg.W + ox);
    if constexpr (V4) {
      // instruction i covers rows 4 i .. 4 i + 3: lane = (row slot s = lane >> 4, LDS piece q = lane & 15)
      const int s = lane >> 4, q = lane & 15;
      for (int i = wave; i < (TA + TB) * 8; i += 8) {
        const bool isa = i < TA * 8;
        const int r = (isa ? i : i - TA * 8) * 4 + s;                 // row of its operand
        const int px = ((q - (r & 15)) & 15) * 4;                     // first of the 4 tile pixels this lane moves
        const int tx4 = px & TWm, ty4 = (px >> d.lgTW) & THm, ti4 = px >> (d.lgTW + d.lgTH);
        const int n4 = n0 + ti4, oy4 = oy0 + ty4, ox4 = ox0 + tx4;
        const bool pv4 = n4 < pg.N && oy4 < d.OH && ox4 < d.OW;
        const float* src;
        if (isa) {
          const int a = a0 + r;
          src = (pv4 && a < d.Ca) ? G.gs + ((long long)n4 * d.gs_bs + (long long)a * OHW + oy4 * d.OW + ox4) : zero4;
        } else {
          const int c = b0 + r;
          src = (pv4 && c < pg.C) ? G.gb + ((long long)n4 * pg.bs + (long long)c * HW + oy4 * pg.W + ox4) : zero4;
        }
        __builtin_amdgcn_global_load_lds(src, (isa ? gsT : gbP) + (isa ? i : i - TA * 8) * 256, 16, 0, 0);
      }
    } else if (lane < npx) {
      for (int r = wave; r < TA * 32; r += 8) {
        const int a = a0 + r;
        __builtin_amdgcn_global_load_lds((pv && a < d.Ca) ? abase + (long long)a * OHW : zero, gsT + r * grow, 4, 0, 0);
      }
      for (int r = wave; r < TB * 32; r += 8) {
        const int c = b0 + r;
        __builtin_amdgcn_global_load_lds((pv && c < pg.C) ? bbase + (long long)c * HW : zero, gbP + r * pg.CS, 4, 0, 0);
      }
    }
  };

  const int h = lane >> 5, l31 = lane & 31;
  const int wk = wave % WK, wsp = wave / WK;
  const int wa = wsp / WB, wb = wsp % WB;
  int boffs[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) boffs[u] = ((wb * 3 + u) * 32 + l31) * brow + (V4 ? 0 : h);
  const int rot = 4 * (l31 & 15) + h;   // V4: column of pixel p in this lane's rows = (p + rot) & 63
  f32x16 acc[9];
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  float bsum[3] = {0.0f, 0.0f, 0.0f};
  const int nkp = npx >> 1;

  stage(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int it = 0; it < niter; ++it) {
    if (it + 1 < niter) stage(it + 1);
    const float* gsT = smem + (it & 1) * bufsz;
    const float* gbP = gsT + gs_sz;
    const float* arow = gsT + (wa * 3 * 32 + l31) * grow + (V4 ? 0 : h);
    float avA[3], bvA[3], avB[3], bvB[3];
    auto fetch = [&](float (&av)[3], float (&bv)[3], int kn) {
      const int col = V4 ? ((2 * kn + rot) & 63) : 2 * kn;
#pragma unroll
      for (int u = 0; u < 3; ++u) av[u] = arow[u * 32 * grow + col];
#pragma unroll
      for (int u = 0; u < 3; ++u) bv[u] = gbP[boffs[u] + col];
    };
    auto mma = [&](const float (&av)[3], const float (&bv)[3]) {
#pragma unroll
      for (int ta = 0; ta < 3; ++ta)
#pragma unroll
        for (int tb = 0; tb < 3; ++tb)
          acc[ta * 3 + tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ta], bv[tb], acc[ta * 3 + tb], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 3; ++u) bsum[u] += av[u];
    };
    fetch(avA, bvA, wk);
    for (int kp = wk; kp < nkp; kp += 2 * WK) {
      fetch(avB, bvB, min(kp + WK, nkp - 1));
      mma(avA, bvA);
      fetch(avA, bvA, min(kp + 2 * WK, nkp - 1));
      if (kp + WK < nkp) mma(avB, bvB);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // ---- fold the WK partial sums of every (wa, wb) group through LDS (the staging buffers are free now)
  constexpr int SET = 9 * 16 * 64;
  constexpr int SLOTS = (WK - 1) < 2 ? (WK - 1) : 2;
  float* red = smem + wsp * SLOTS * SET;
#pragma unroll
  for (int r0 = 1; r0 < WK; r0 += SLOTS) {
    if (wk >= r0 && wk < r0 + SLOTS) {
      float* dst = red + (wk - r0) * SET;
#pragma unroll
      for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(i * 16 + r) * 64 + lane] = acc[i][r];
    }
    __syncthreads();
    if (wk == 0) {
      for (int k = 0; k < SLOTS && r0 + k < WK; ++k)
#pragma unroll
        for (int i = 0; i < 9; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][r] += red[k * SET + (i * 16 + r) * 64 + lane];
    }
    __syncthreads();
  }
  // bias gradient: row sums seen by the wb == 0 waves, folded over h and over the K split in wave order
  if (G.dbias_ws != nullptr && bt == 0) {
    float* bred = smem;   // [WA][WK][96] after the accumulator fold (barrier above)
    if (wb == 0) {
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const float sfull = bsum[u] + __shfl_xor(bsum[u], 32, 64);
        if (h == 0) bred[(wa * WK + wk) * 96 + u * 32 + l31] = sfull;
      }
    }
    __syncthreads();
    if (wb == 0 && wk == 0 && h == 0) {
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        float t = 0.0f;
        for (int k = 0; k < WK; ++k) t += bred[(wa * WK + k) * 96 + u * 32 + l31];
        const int a = a0 + (wa * 3 + u) * 32 + l31;
        if (a < d.Ca) G.dbias_ws[(long long)split * d.Ca + a] = t;
      }
    }
  }
  if (wk != 0) return;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int ta = wa * 3 + i / 3, tb = wb * 3 + i % 3;
    const int b = b0 + tb * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int a = a0 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (a < d.Ca && b < pg.C) G.ws[((long long)split * d.Ca + a) * pg.C + b] = acc[i][r];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- general kernel: accumulators = (tap, a-tile, b-tile) triples dealt to the four MFMA waves ----------------------
template <int TA, int TB, int NACC>
__global__ __launch_bounds__(512, 2) void wgrad_kernel(const WgDesc d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const PatchGeom& pg = d.pg;
  const int npx = 1 << d.lgNPX, grow = npx + 1;   // pixels per tile (16/32/64), gs row stride
  const int gs_sz = TA * 32 * grow, gb_sz = TB * 32 * pg.CS;
  const int bufsz = gs_sz + gb_sz;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool loader = wave >= 4;
  const WgPtrs G = d.g[blockIdx.y];

  // XCD-aware order (see conv_igemm.hip): hardware id b runs on XCD b % 8; every XCD gets a contiguous run of logical
  // workgroups, so the (a-tile, b-tile, tap-group) workgroups of one pixel split read their pixels through one L2
  // (only for problems with many pixels: the 4 096-pixel slice-chain problems measured 3 % slower with it)
  int bid = blockIdx.x;
  if (d.xcd_order) {
    const int nb = gridDim.x, hb = blockIdx.x;
    const int xcd = hb & 7, q = hb >> 3;
    bid = xcd * (nb >> 3) + min(xcd, nb & 7) + q;
  }
  const int at = bid % d.natile; bid /= d.natile;
  const int bt = bid % d.nbtile; bid /= d.nbtile;
  const int tg = bid % d.ngroups;
  const int split = bid / d.ngroups;
  const int a0 = at * TA * 32, b0 = bt * TB * 32, t0 = tg * d.tpg;
  const int nt = min(d.tpg, d.ntaps - t0);
  const int TWm = (1 << d.lgTW) - 1, THm = (1 << d.lgTH) - 1;
  const int niter = (d.ntiles - split + d.nsplit - 1) / d.nsplit;

  if (loader) {
    wgrad_loader<TA, TB>(d, G, smem, tid, lane, wave, a0, b0, split, niter, bt, tg);
    return;
  }

  // ------------------------------------------------------------------ MFMA waves
  const int h = lane >> 5, l31 = lane & 31;
  // accumulator i of this wave: tile q = wave + 4*i -> (tap, ta, tb); all wave-uniform
  const int ntile = nt * TA * TB;
  int boffs[NACC];
  int tsel[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    const int q = wave + 4 * i;
    const int qq = q < ntile ? q : 0;
    const int tap = qq / (TA * TB), rem = qq % (TA * TB);
    const int ta = rem / TB, tb = rem % TB;
    tsel[i] = q < ntile ? ta : -1;
    boffs[i] = (tb * 32 + l31) * pg.CS + d.tapoff[t0 + tap] + h * d.po_h;
  }
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

  const int pe_lane = d.pe[lane & 31];   // pixel-pair offsets live in one VGPR; v_readlane picks entry kp
  // A tile of accumulator i: tile q = wave + 4*i has ta = (q % (TA*TB)) / TB, which is `i` for the 4x4 (1x1-conv)
  // configuration and wave-uniform otherwise (TA*TB divides 4) -> no per-MFMA operand select, one A read per pair
  constexpr bool kPerAcc = (TA * TB == 16);
  static_assert(kPerAcc ? (NACC == 4 && TA == 4) : (4 % (TA * TB) == 0), "tile -> A-row mapping");
  constexpr int NA = kPerAcc ? TA : 1;
  const int ta_w = kPerAcc ? 0 : (wave % (TA * TB)) / TB;
  __syncthreads();  // tile 0 staged
  const int nkp = npx >> 1;   // even (npx is 16 / 32 / 64)
  for (int it = 0; it < niter; ++it) {
    const float* gsT = smem + (it & 1) * bufsz;
    const float* gbP = gsT + gs_sz;
    const float* arow = gsT + (ta_w * 32 + l31) * grow + h;
    // Two explicit register sets (ping-pong), loop unrolled by two pixel pairs: the fragments of pair kp+1 are
    // fetched while the MFMAs of pair kp issue and nothing is copied between iterations, so no s_waitcnt ever
    // waits on an LDS read issued in the same half-iteration.
    float avA[NA], bvA[NACC], avB[NA], bvB[NACC];
    auto fetch = [&](float (&av)[NA], float (&bv)[NACC], int kn) {
      const int po = __builtin_amdgcn_readlane(pe_lane, kn);
#pragma unroll
      for (int u = 0; u < NA; ++u) av[u] = arow[u * 32 * grow + 2 * kn];
#pragma unroll
      for (int i = 0; i < NACC; ++i) bv[i] = gbP[boffs[i] + po];
    };
    auto mma = [&](const float (&av)[NA], const float (&bv)[NACC]) {
#pragma unroll
      for (int i = 0; i < NACC; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kPerAcc ? i : 0], bv[i], acc[i], 0, 0, 0);
    };
    fetch(avA, bvA, 0);
    for (int kp = 0; kp < nkp; kp += 2) {
      fetch(avB, bvB, kp + 1);
      mma(avA, bvA);
#pragma unroll
      for (int i = 0; i < NACC; ++i) {   // one MFMA, then the address VALU + LDS reads of the other set
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
      fetch(avA, bvA, min(kp + 2, nkp - 1));   // the last fetch re-reads the final pair (never used)
      mma(avB, bvB);
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
    }
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    if (tsel[i] < 0) continue;
    const int q = wave + 4 * i;
    const int tap = t0 + q / (TA * TB), rem = q % (TA * TB);
    const int ta = rem / TB, tb = rem % TB;
    const int b = b0 + tb * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int a = a0 + ta * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (a < d.Ca && b < pg.C)
        G.ws[(((long long)split * d.ntaps + tap) * d.Ca + a) * pg.C + b] = acc[i][r];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// dw[(a*Cb + b)*ntaps + t] (+)= sum_s ws[((s*ntaps + t)*Ca + a)*Cb + b]      (+ db[a] (+)= sum_s dbias_ws[s][a])
// A slab is [tap][a][b] with (a, b) contiguous, so a workgroup takes a CONTIGUOUS run of 256 (a, b) positions (1024
// with 16-byte loads for 1x1 problems): every load instruction of a wave reads one 256-B / 1-KB segment per
// (split, tap) -- the earlier (a-row x 32 b) tiling read 128-B pieces scattered over 64 split slabs and ran at
// 0.2 TB/s.  Splits are independent rows: 8 loads in flight per thread.  Taps are transposed through LDS so that the
// canonical [a][b][kh][kw] layout is written in contiguous runs.  blockIdx.y = problem of the group; the last
// blockIdx.x rows of the grid reduce the fused bias gradients.
struct RedPtrs {
  const float* ws;
  float* dw;
  const float* dbias_ws;
  float* dbias;
  int accum, accum_bias;
  int dw_ld;   // row length (in b-columns) of the destination tensor: Cb, or the full input-channel count of a wider weight
};
struct RedDesc {
  RedPtrs g[WG_MAXG];
  int Ca, Cb, ntaps, nsplit, nsplit_bias, nwblocks, vec4;
  int wino;   // the slabs hold Winograd-domain gradients dU[16][a][b] (wgrad_wino.hip): ntaps = 16 in, dW = G^T dU G (9 taps) out
};
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const RedDesc d) {
  __shared__ __attribute__((aligned(16))) float tile[WG_MAX_TAPS][257];
  const RedPtrs G = d.g[blockIdx.y];
  const int Ca = d.Ca, ntaps = d.ntaps, nsplit = d.nsplit;
  if ((int)blockIdx.x >= d.nwblocks) {   // bias part: 16 channels x 16 split lanes per workgroup, combined in lane order
    if (!G.dbias) return;
    float* part = &tile[0][0];
    const int pl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int a = ((int)blockIdx.x - d.nwblocks) * 16 + pl;
    float s = 0.0f;
    if (a < Ca)
      for (int k = sl; k < d.nsplit_bias; k += 16) s += G.dbias_ws[(long long)k * Ca + a];
    part[sl * 16 + pl] = s;
    __syncthreads();
    if (sl == 0 && a < Ca) {
      float t = part[pl];
#pragma unroll
      for (int u = 1; u < 16; ++u) t += part[u * 16 + pl];
      if (G.accum_bias) t += G.dbias[a];
      G.dbias[a] = t;
    }
    return;
  }
  const long long CaCb = (long long)Ca * d.Cb;
  const long long slab = CaCb * ntaps;
  if (d.vec4) {
    // 1x1 problems, 16-byte aligned.  Workgroup = 16 float4 positions x 16 split lanes: lane sl adds the slabs
    // k = sl (mod 16) (a few hundred slabs of a 1x1 problem are summed by ~600 workgroups with 1-2 load batches each,
    // instead of 36 workgroups walking 256 slabs: that walk cost more than the GEMM); the 16 partial sums meet in LDS
    // and are added in lane order -- deterministic.
    f32x4* part = reinterpret_cast<f32x4*>(&tile[0][0]);   // [16 split lanes][16 positions]
    const int pl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const long long ab = ((long long)blockIdx.x * 16 + pl) * 4;
    f32x4 s = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (ab < CaCb) {
      const float* p = G.ws + ab;
      int k = sl;
      for (; k + 7 * 16 < nsplit; k += 8 * 16) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (long long)(k + 16 * u) * slab);
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
      for (; k < nsplit; k += 16) s += *reinterpret_cast<const f32x4*>(p + (long long)k * slab);
    }
    part[sl * 16 + pl] = s;
    __syncthreads();
    if (sl == 0 && ab < CaCb) {
      f32x4 t = part[pl];
#pragma unroll
      for (int u = 1; u < 16; ++u) t += part[u * 16 + pl];
      const long long ar = ab / d.Cb;   // (Cb % 4 == 0 and dw_ld % 4 == 0 when dw_ld != Cb: host-checked)
      f32x4* o = reinterpret_cast<f32x4*>(G.dw + (G.dw_ld == d.Cb ? ab : ar * G.dw_ld + (ab - ar * d.Cb)));
      if (G.accum) t += *o;
      *o = t;
    }
    return;
  }
  // multi-tap problems: workgroup = 64 (a, b) positions x 4 tap lanes (lane tl sums the taps t = tl (mod 4)); two taps
  // x up to 8 splits = 16 independent loads in flight per thread.  (One thread per position walking all taps one after
  // the other -- 25 dependent load batches for a 5x5 -- kept ~100 reductions per step latency-bound: 4 ms per step.)
  const int pl = threadIdx.x & 63, tl = threadIdx.x >> 6;
  const long long ab0 = (long long)blockIdx.x * 64;
  const long long ab = ab0 + pl;
  const bool valid = ab < CaCb;
  for (int t = tl; t < ntaps; t += 8) {
    const int t2 = t + 4;
    float sa = 0.0f, sb = 0.0f;
    if (valid) {
      const float* pa = G.ws + (long long)t * CaCb + ab;
      const float* pb = G.ws + (long long)(t2 < ntaps ? t2 : t) * CaCb + ab;
      for (int k = 0; k < nsplit; k += 8) {
        float va[8], vb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          va[u] = (k + u < nsplit) ? pa[(long long)(k + u) * slab] : 0.0f;
          vb[u] = (k + u < nsplit) ? pb[(long long)(k + u) * slab] : 0.0f;
        }
        sa += ((va[0] + va[1]) + (va[2] + va[3])) + ((va[4] + va[5]) + (va[6] + va[7]));
        sb += ((vb[0] + vb[1]) + (vb[2] + vb[3])) + ((vb[4] + vb[5]) + (vb[6] + vb[7]));
      }
    }
    tile[t][pl] = sa;
    if (t2 < ntaps) tile[t2][pl] = sb;
  }
  __syncthreads();
  const int nv = (int)min((long long)64, CaCb - ab0);
  if (d.wino) {
    // dW[p][q] = sum_ij G[i][p] dU[i][j] G[j][q],  G = [[1,0,0],[1/2,1/2,1/2],[1/2,-1/2,1/2],[0,0,1]]  (linear: applied
    // once, after the pixel splits have been added)
    for (int e = threadIdx.x; e < nv * 9; e += 256) {
      const int abl = e / 9, t = e - abl * 9;
      const int p = t / 3, q = t - p * 3;
      float col[4];   // (dU G)[i][q]
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float u0 = tile[i * 4 + 0][abl], u1 = tile[i * 4 + 1][abl], u2 = tile[i * 4 + 2][abl], u3 = tile[i * 4 + 3][abl];
        col[i] = q == 0 ? u0 + 0.5f * (u1 + u2) : (q == 1 ? 0.5f * (u1 - u2) : 0.5f * (u1 + u2) + u3);
      }
      float v = p == 0 ? col[0] + 0.5f * (col[1] + col[2]) : (p == 1 ? 0.5f * (col[1] - col[2]) : 0.5f * (col[1] + col[2]) + col[3]);
      const long long abg = ab0 + abl, ar = abg / d.Cb;
      float* o = G.dw + (ar * G.dw_ld + (abg - ar * d.Cb)) * 9 + t;
      if (G.accum) v += *o;
      *o = v;
    }
    return;
  }
  if (G.dw_ld == d.Cb) {
    float* o = G.dw + ab0 * ntaps;
    for (int e = threadIdx.x; e < nv * ntaps; e += 256) {
      const int abl = e / ntaps, t = e - abl * ntaps;
      float v = tile[t][abl];
      if (G.accum) v += o[e];
      o[e] = v;
    }
  } else {   // column block of a wider weight: rows are dw_ld * ntaps floats apart
    for (int e = threadIdx.x; e < nv * ntaps; e += 256) {
      const int abl = e / ntaps, t = e - abl * ntaps;
      const long long abg = ab0 + abl, ar = abg / d.Cb;
      float* o = G.dw + (ar * G.dw_ld + (abg - ar * d.Cb)) * ntaps + t;
      float v = tile[t][abl];
      if (G.accum) v += *o;
      *o = v;
    }
  }
}

// test hooks (icm_debug_force_wgrad_cfg): kernel variant 0 = <2,1,7>, 1 = <2,2,9>, 2 = <4,4,4> (general kernel);
// 3 = t33<3,3,1,1>, 4 = t33<6,6,2,2>, 5 = t33<3,6,1,2>, 6 = t33<6,3,2,1> (3 x 3 tiles per wave, one tap per workgroup);
// XCD-aware workgroup order 0 / 1; -1 = automatic choice
#define WG_NVARIANTS 15  /* 7 = t33<3,3,1,1,tap-per-wave>: 96 x 96 x 4 taps per workgroup (3x3 problems);
                          * 8 / 9 = tap9<3,3> / tap9<2,2>: all nine taps from one DMA staging (3x3 s1 p1, no activation) */
static int g_force_variant = -1, g_force_xcd = -1;

struct WgPlan {
  int ta, tb, nacc, tpg, ws;
  int lgTW, lgTH, lgTI, lgNPX, PH, PW, PP, CS;
  int tiles_x, tiles_y, tiles_n, ntiles, nsplit, natile, nbtile, ngroups;
  size_t lds;
};

static int plan_wgrad(const icm_wgrad_args& a, WgPlan& p, int nproblems = 1) {
  if (!a.gs || !a.gb || a.N <= 0 || a.Ca <= 0 || a.Cb <= 0 || a.KH * a.KW > WG_MAX_TAPS) return ICM_ERR_ARG;
  if (a.stride != 1 && a.stride != 2) return ICM_ERR_UNSUPPORTED;
  if (a.OH != (a.H + 2 * a.pad - a.KH) / a.stride + 1 || a.OW != (a.W + 2 * a.pad - a.KW) / a.stride + 1)
    return ICM_ERR_ARG;
  const int ntaps = a.KH * a.KW;
  bool ok = false;
  for (int lg = 6; lg >= 4 && !ok; --lg) {   // pixels per tile: 64, else 32 / 16 for tiny images with big halos
    p.lgNPX = lg;
    // tile width: up to 16 pixels for halo patches (keeps the patch rows short); halo-free (1x1) problems take the
    // widest rows the image offers -- every channel row of a tile is one contiguous run in memory, and 64-byte runs
    // (16 pixels) scattered over channel planes use a fraction of what 256-byte runs get from L2 / the memory side
    const bool halo_free = a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0;
    p.lgTW = std::min(std::min(halo_free ? lg : 4, lg), ceil_log2(a.OW));
    p.lgTH = std::min(lg - p.lgTW, ceil_log2(a.OH));
    p.lgTI = lg - p.lgTW - p.lgTH;
    const int TW = 1 << p.lgTW, TH = 1 << p.lgTH, TI = 1 << p.lgTI;
    p.PW = (TW - 1) * a.stride + a.KW;
    p.PH = (TH - 1) * a.stride + a.KH;
    p.PP = p.PH * p.PW;
    p.CS = TI * p.PP;
    if ((p.CS & 1) == 0) p.CS += 1;
    auto lds_of = [&](int ta, int tb) {
      return (size_t)2 * (ta * 32 * ((1 << lg) + 1) + tb * 32 * p.CS) * 4;
    };
    p.ws = 0;
    auto t33 = [&](int v, int ta, int tb) -> bool {
      // the fold of the K-split waves parks up to two accumulator sets per (wa, wb) group in the staging LDS
      const int wk = 4 / ((ta / 3) * (tb / 3));
      const size_t fold = wk > 1 ? (size_t)(ta / 3) * (tb / 3) * std::min(wk - 1, 2) * 9 * 16 * 64 * 4 : 0;
      const size_t need = std::max(lds_of(ta, tb), fold);
      if (need > 160 * 1024) return false;
      p.ta = ta; p.tb = tb; p.nacc = 9; p.tpg = 1; p.ws = v;
      return true;
    };
    auto dma1 = [&](int v, int ta, int tb) -> bool {   // 1x1, activation-free: 8 MFMA waves, DMA-only staging
      if (!(ntaps == 1 && a.stride == 1 && a.pad == 0 && a.act_s == ICM_ACT_NONE && a.act_b == ICM_ACT_NONE)) return false;
      const int groups = (ta / 3) * (tb / 3), wk = 8 / groups;
      const size_t fold = (size_t)groups * std::min(wk - 1, 2) * 9 * 16 * 64 * 4;
      if (std::max(lds_of(ta, tb), fold) > 160 * 1024) return false;
      p.ta = ta; p.tb = tb; p.nacc = 9; p.tpg = 1; p.ws = v;
      return true;
    };
    auto variant = [&](int v) -> bool {   // kernel variant v for this tile size; false if its LDS does not fit
      switch (v) {
        case 11: return dma1(11, 6, 6);
        case 12: return dma1(12, 3, 6);
        case 13: return dma1(13, 6, 3);
        case 14: return dma1(14, 3, 3);
        case 3: return t33(3, 3, 3);
        case 4: return t33(4, 6, 6);
        case 5: return t33(5, 3, 6);
        case 6: return t33(6, 6, 3);
        case 8:
        case 9: {   // all nine taps of a 3x3 stride-1 problem from one DMA staging: 96 x 96 (8) or 64 x 64 (9) blocks
          const int t = v == 8 ? 3 : 2;
          if (!(a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.act_s == ICM_ACT_NONE &&
                a.act_b == ICM_ACT_NONE) || lds_of(t, t) > 160 * 1024)
            return false;
          p.ta = t; p.tb = t; p.nacc = 9; p.tpg = 9; p.ws = v;
          return true;
        }
        case 10:   // all 25 taps of a 5x5 problem (stride 1 or 2) from one DMA staging: 96 x 32 blocks, three taps per wave
          if (!(a.KH == 5 && a.KW == 5 && a.pad == 2 && a.act_s == ICM_ACT_NONE && a.act_b == ICM_ACT_NONE) ||
              lds_of(3, 1) > 160 * 1024)
            return false;
          p.ta = 3; p.tb = 1; p.nacc = 10; p.tpg = 25; p.ws = v;
          return true;
        case 7:   // 96 x 96 (a, b) block, four taps per workgroup (one per MFMA wave), no K split
          if (lds_of(3, 3) > 160 * 1024) return false;
          p.ta = 3; p.tb = 3; p.nacc = 9; p.tpg = std::min(ntaps, 4); p.ws = 7;
          return true;
        case 2: if (lds_of(4, 4) > 150 * 1024) return false; p.ta = 4; p.tb = 4; p.nacc = 4; p.tpg = 1; return true;
        case 1: if (lds_of(2, 2) > 150 * 1024) return false; p.ta = 2; p.tb = 2; p.nacc = 9; p.tpg = std::min(ntaps, 9); return true;
        default:   // <=14 taps per group x 2 a-tiles = 28 tiles = 7 per MFMA wave
          if (lds_of(2, 1) > 150 * 1024) return false;
          p.ta = 2; p.tb = 1; p.nacc = 7; p.tpg = ntaps <= 14 ? ntaps : (ntaps + 1) / 2;
          return p.tpg <= 14;
      }
    };
    if (g_force_variant >= 0) ok = variant(g_force_variant);
    else if (ntaps == 1) {
      // 1x1 problems: the loaders are latency-bound, so time ~ padded MACs x max(1, kappa x staged bytes per MAC);
      // kappa from the measured 45 TF of the 96 x 96 tile (0.29 of the MFMA peak; profiles/r02_*)
      static const struct { int v, ta, tb; } cand[] = {{4, 6, 6}, {5, 3, 6}, {6, 6, 3}, {3, 3, 3}, {2, 4, 4}};
      double bestc = 1e300;
      int bestv = -1;
      for (const auto& c : cand) {
        const size_t need = c.v == 2 ? lds_of(4, 4) : lds_of(c.ta, c.tb);
        if (need > 160 * 1024) continue;
        const double area = (double)cdiv(a.Ca, 32 * c.ta) * 32 * c.ta * cdiv(a.Cb, 32 * c.tb) * 32 * c.tb;
        const double cost = area * std::max(1.0, 5.2 * (1.0 / c.ta + 1.0 / c.tb));
        if (cost < bestc - 1e-9) { bestc = cost; bestv = c.v; }
      }
      ok = bestv >= 0 && variant(bestv);
      // activation-free operands: the DMA-only kernel (8 MFMA waves).  Staging costs it little, so the block is chosen
      // by padded area alone; among equal areas the four-way K split (96 x 192 / 192 x 96 blocks) measured best:
      // 192 -> 192 @128: <6,3> 75.8, <3,6> 73.4, <6,6> 68.3, <3,3> 63.9 TF against 58.7 for the loader-wave kernel
      // (profiles/r02_tune_wgrad_dma1.txt)
      static const int dma1_on = getenv("ICM_WG_DMA1") ? atoi(getenv("ICM_WG_DMA1")) : 1;
      if (ok && dma1_on && a.act_s == ICM_ACT_NONE && a.act_b == ICM_ACT_NONE && a.stride == 1 && a.pad == 0) {
        static const struct { int v, ta, tb; } dc[] = {{13, 6, 3}, {12, 3, 6}, {11, 6, 6}, {14, 3, 3}};
        double ba = 1e300;
        int bv = -1;
        for (const auto& c : dc) {
          const double area = (double)cdiv(a.Ca, 32 * c.ta) * 32 * c.ta * cdiv(a.Cb, 32 * c.tb) * 32 * c.tb;
          if (area < ba - 1e-9) {
            WgPlan keep = p;
            if (variant(c.v)) { ba = area; bv = c.v; }
            p = keep;
          }
        }
        if (bv >= 0) variant(bv);
      }
      // a smaller pixel tile is only worth it for the big tiles that need it (<6,6> does not fit 64 pixels)
      if (ok && lg == 6 && p.ws < 11) {
        const double area6 = (double)cdiv(a.Ca, 32 * p.ta) * 32 * p.ta * cdiv(a.Cb, 32 * p.tb) * 32 * p.tb *
                             std::max(1.0, 5.2 * (1.0 / p.ta + 1.0 / p.tb));
        const double area66 = (double)cdiv(a.Ca, 192) * 192 * cdiv(a.Cb, 192) * 192 * std::max(1.0, 5.2 / 3.0);
        if (area66 < area6 - 1e-9) ok = false;   // retry at lg = 5, where <6,6> fits
      }
    }
    else if (ntaps <= 9 && lds_of(2, 2) <= 150 * 1024) {
      // 3x3: 64 x 64 x 9-tap tiles, unless 96-wide tiles fit the channel counts so much better (96 / 192 channels:
      // 100 % against 56 %) that idling the waves without a tap in the last tap group (9 taps = 4 + 4 + 1) still wins
      const double pad64 = (double)cdiv(a.Ca, 64) * 64 * cdiv(a.Cb, 64) * 64;
      const double pad96 = (double)cdiv(a.Ca, 96) * 96 * cdiv(a.Cb, 96) * 96 * (4.0 * cdiv(ntaps, 4) / ntaps);
      // activation-free 3x3 stride-1 problems whose channels fit 96-wide blocks, with many pixels: the nine-tap
      // single-staging kernel (96 -> 96 @ 64x64 x6: 92 TF against 65 for the tap-group kernel and 54 for <2,2,9>).  For
      // the 4 096-pixel slice-chain problems <2,2,9> (two workgroups per CU, 64 tiles per problem) stays ahead
      // (77 vs 71 TF at 480 -> 224 x10; profiles/r02_tune_wgrad_tap9.txt), as it does over the 64 x 64 tap9 form.
      static const int tap9_on = getenv("ICM_WG_TAP9") ? atoi(getenv("ICM_WG_TAP9")) : 1;
      const double p96 = (double)cdiv(a.Ca, 96) * 96 * cdiv(a.Cb, 96) * 96;
      ok = tap9_on && ntaps == 9 && p96 <= pad64 && (long long)a.N * a.OH * a.OW >= 16384 && variant(8);
      if (!ok) ok = (ntaps > 1 && pad96 * 1.1 < pad64 && variant(7)) || variant(1);
    }
    else {
      static const int tap25_on = getenv("ICM_WG_TAP25") ? atoi(getenv("ICM_WG_TAP25")) : 1;
      ok = tap25_on && ntaps == 25 && (long long)a.N * a.OH * a.OW >= 16384 && variant(10);
      if (ok && TI * p.PP > ICM_MAXJ * 64) ok = false;
      if (!ok) ok = variant(0);
    }
    if (ok && TI * p.PP > ICM_MAXJ * 64) ok = false;   // PlaneMap capacity
    if (ok) {
      p.lds = lds_of(p.ta, p.tb);
      if (p.ws && p.ws < 7) {
        const int wk = 4 / ((p.ta / 3) * (p.tb / 3));
        if (wk > 1) p.lds = std::max(p.lds, (size_t)(p.ta / 3) * (p.tb / 3) * std::min(wk - 1, 2) * 9 * 16 * 64 * 4);
      }
      if (p.ws >= 11) {
        const int groups = (p.ta / 3) * (p.tb / 3), wk = 8 / groups;
        p.lds = std::max(p.lds, (size_t)groups * std::min(wk - 1, 2) * 9 * 16 * 64 * 4);
      }
    }
  }
  if (!ok) return ICM_ERR_UNSUPPORTED;
  const int TW = 1 << p.lgTW, TH = 1 << p.lgTH, TI = 1 << p.lgTI;
  p.tiles_x = cdiv(a.OW, TW); p.tiles_y = cdiv(a.OH, TH); p.tiles_n = cdiv(a.N, TI);
  p.ntiles = p.tiles_x * p.tiles_y * p.tiles_n;
  p.natile = cdiv(a.Ca, 32 * p.ta); p.nbtile = cdiv(a.Cb, 32 * p.tb); p.ngroups = cdiv(ntaps, p.tpg);
  const int base = p.natile * p.nbtile * p.ngroups * std::max(1, nproblems);
  // pixel splits: minimise (rounds of workgroups over the CUs) x (pixel tiles per workgroup + un-overlapped
  // prologue), with a small charge per split for the extra slab traffic
  const int occ = std::max<int>(1, std::min<int>(2, (int)((160 * 1024) / p.lds)));
  const double slots = 256.0 * occ;
  double best = 1e300;
  p.nsplit = 1;
  for (int sp = 1; sp <= std::min(p.ntiles, 256); ++sp) {
    const double rounds = std::ceil((double)base * sp / slots);
    const double per = (double)cdiv(p.ntiles, sp) + 0.8;
    const double cost = rounds * per * (1.0 + 0.004 * sp);
    if (cost < best - 1e-9) { best = cost; p.nsplit = sp; }
  }
  return ICM_OK;
}

}  // namespace icm

extern "C" {

int64_t icm_wgrad_workspace_floats_grouped(const icm_wgrad_args* a, int n) {
  icm::WgPlan p;
  if (a && n >= 1 && a->algo == ICM_ALGO_WINOGRAD) {
    int nsplit = 0, nchunks = 0;
    if (!a->gs || !a->gb || a->N <= 0 || a->Ca <= 0 || a->Cb <= 0 || icm::wgrad_wino_plan(*a, n, &nsplit, &nchunks)) return -1;
    return (int64_t)nsplit * 16 * a->Ca * a->Cb + (int64_t)nsplit * a->Ca;
  }
  if (!a || n < 1 || icm::plan_wgrad(*a, p, n)) return -1;
  return (int64_t)p.nsplit * a->KH * a->KW * a->Ca * a->Cb + (int64_t)p.nsplit * a->Ca;
}
int64_t icm_wgrad_workspace_floats(const icm_wgrad_args* a) { return icm_wgrad_workspace_floats_grouped(a, 1); }

static int wgrad_grouped(const icm_wgrad_args* arr, int n, hipStream_t stream) {
  using namespace icm;
  if (!arr || n < 1 || n > WG_MAXG) return ICM_ERR_ARG;
  const icm_wgrad_args* a = &arr[0];
  if (a->algo == ICM_ALGO_WINOGRAD) {
    int nsplit = 0, nchunks = 0;
    if (!a->gs || !a->gb || a->N <= 0 || a->Ca <= 0 || a->Cb <= 0) return ICM_ERR_ARG;
    int rcw = wgrad_wino_plan(*a, n, &nsplit, &nchunks);
    if (rcw) return rcw;
    if ((long long)a->N * ((a->W + 1) / 2) * ((a->H + 1) / 2) >= 65536) return ICM_ERR_UNSUPPORTED;   // tile index fits the fast division
    const long long slab_all = (long long)nsplit * 16 * a->Ca * a->Cb;
    float* wsp[WG_MAXG];
    float* dbp[WG_MAXG];
    RedDesc r{};
    for (int i = 0; i < n; ++i) {
      const icm_wgrad_args& b = arr[i];
      if (!b.gs || !b.gb || !b.dw || !b.ws) return ICM_ERR_ARG;
      if (b.Ca != a->Ca || b.Cb != a->Cb || b.OH != a->OH || b.OW != a->OW || b.H != a->H || b.W != a->W || b.N != a->N ||
          b.KH != a->KH || b.KW != a->KW || b.stride != a->stride || b.pad != a->pad || b.act_s != a->act_s ||
          b.act_b != a->act_b || b.gs_bs != a->gs_bs || b.gb_bs != a->gb_bs || b.algo != a->algo)
        return ICM_ERR_ARG;
      if (b.ws_floats > 0 && b.ws_floats < slab_all + (long long)nsplit * a->Ca) return ICM_ERR_ARG;
      wsp[i] = b.ws;
      dbp[i] = b.dbias ? b.ws + slab_all : nullptr;
    }
    rcw = launch_wgrad_wino(arr, n, nsplit, nchunks, wsp, dbp, stream);
    if (rcw) return rcw;
    bool any_bias = false;
    for (int i = 0; i < WG_MAXG; ++i) {
      const icm_wgrad_args& b = arr[i < n ? i : 0];
      r.g[i].ws = b.ws; r.g[i].dw = b.dw; r.g[i].dbias_ws = b.dbias ? b.ws + slab_all : nullptr; r.g[i].dbias = b.dbias;
      r.g[i].accum = b.accum; r.g[i].accum_bias = b.accum_bias;
      r.g[i].dw_ld = b.dw_ld > 0 ? b.dw_ld : a->Cb;
      if (r.g[i].dw_ld < a->Cb) return ICM_ERR_ARG;
      if (i < n) any_bias |= b.dbias != nullptr;
    }
    r.Ca = a->Ca; r.Cb = a->Cb; r.ntaps = 16; r.nsplit = nsplit; r.nsplit_bias = nsplit; r.vec4 = 0; r.wino = 1;
    r.nwblocks = (int)(((long long)a->Ca * a->Cb + 63) / 64);
    const int rblocks = r.nwblocks + (any_bias ? cdiv(a->Ca, 16) : 0);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rblocks, n), dim3(256), 0, stream, r);
    ICM_CHECK_LAUNCH();
    return ICM_OK;
  }
  if (a->algo != ICM_ALGO_DIRECT) return ICM_ERR_ARG;
  WgPlan p;
  int rc = plan_wgrad(*a, p, n);
  if (rc) return rc;
  for (int i = 0; i < n; ++i) {
    const icm_wgrad_args& b = arr[i];
    if (!b.gs || !b.gb || !b.dw || !b.ws) return ICM_ERR_ARG;
    if (b.Ca != a->Ca || b.Cb != a->Cb || b.OH != a->OH || b.OW != a->OW || b.H != a->H || b.W != a->W ||
        b.N != a->N || b.KH != a->KH || b.KW != a->KW || b.stride != a->stride || b.pad != a->pad ||
        b.act_s != a->act_s || b.act_b != a->act_b || b.gs_bs != a->gs_bs || b.gb_bs != a->gb_bs || b.algo != a->algo)
      return ICM_ERR_ARG;
  }
  if (((long long)a->N * a->gb_bs + 8LL * a->H * a->W) * 4 >= (1LL << 31)) return ICM_ERR_UNSUPPORTED;   // PlaneMap byte offsets are int32
  const int ntaps = a->KH * a->KW;
  const int nslab = p.nsplit;   // (wave-split mode folds its four waves through LDS: still one slab per pixel split)
  const long long slab_all = (long long)nslab * ntaps * a->Ca * a->Cb;
  for (int i = 0; i < n; ++i)   // the slabs (+ bias partials) of this launch's split count must fit the caller's workspace
    if (arr[i].ws_floats > 0 && arr[i].ws_floats < slab_all + (long long)p.nsplit * a->Ca) return ICM_ERR_ARG;
  WgDesc d{};
  RedDesc r{};
  for (int i = 0; i < WG_MAXG; ++i) {
    const icm_wgrad_args& b = arr[i < n ? i : 0];
    d.g[i].gs = b.gs; d.g[i].gb = b.gb; d.g[i].ws = b.ws;
    d.g[i].dbias_ws = b.dbias ? b.ws + slab_all : nullptr;
    r.g[i].ws = b.ws; r.g[i].dw = b.dw; r.g[i].dbias_ws = d.g[i].dbias_ws; r.g[i].dbias = b.dbias;
    r.g[i].accum = b.accum; r.g[i].accum_bias = b.accum_bias;
    r.g[i].dw_ld = b.dw_ld > 0 ? b.dw_ld : a->Cb;
    if (r.g[i].dw_ld < a->Cb) return ICM_ERR_ARG;
  }
  d.gs_bs = a->gs_bs;
  PatchGeom& pg = d.pg;
  pg.PW = p.PW; pg.PH = p.PH; pg.PWrow = p.PW; pg.PWh = 0; pg.PP = p.PP; pg.CS = p.CS; pg.S = 1;
  pg.TIPH = (1 << p.lgTI) * p.PH;
  pg.dPW = make_fastdiv((uint32_t)p.PW);
  pg.dTIPH = make_fastdiv((uint32_t)pg.TIPH);
  pg.dPH = make_fastdiv((uint32_t)p.PH);
  pg.H = a->H; pg.W = a->W; pg.N = a->N; pg.C = a->Cb; pg.act = a->act_b; pg.bs = a->gb_bs;
  pg.seg_len = 0; pg.seg_gap = 0; pg.dseg = make_fastdiv(1);   // no blocked channel map on the weight-gradient operands
  {
    const int TW = 1 << p.lgTW;
    bool v4 = ntaps == 1 && a->stride == 1 && a->pad == 0 && p.lgNPX >= 5 && (TW % 4) == 0 && (a->W % 4) == 0 &&
              (a->gb_bs % 4) == 0 && ((long long)a->H * a->W % 4) == 0 && (p.PW % 4) == 0 && (p.PP % 4) == 0;
    bool g4 = p.lgNPX >= 5 && (TW % 4) == 0 && (a->OW % 4) == 0 && (a->gs_bs % 4) == 0 &&
              ((long long)a->OH * a->OW % 4) == 0;
    for (int i = 0; i < n; ++i) {
      v4 = v4 && ((reinterpret_cast<uintptr_t>(arr[i].gb) & 15) == 0);
      g4 = g4 && ((reinterpret_cast<uintptr_t>(arr[i].gs) & 15) == 0);
    }
    pg.vec4 = v4 ? 1 : 0;
    set_v4_pack(pg);
    d.gs_vec4 = g4 ? 1 : 0;
    // big-grid operand by LDS-DMA: no activation to apply and a halo patch (the halo-free 16-byte register path moves
    // 4x the bytes per instruction and is kept); wgrad patches are stored linearly (identity column map, PWrow = PW)
    static const int dma_on = getenv("ICM_WG_DMA") ? atoi(getenv("ICM_WG_DMA")) : 1;
    pg.dma = (dma_on && (a->act_b == ICM_ACT_NONE || dma_on > 1) && !v4) ? 1 : 0;   // ICM_WG_DMA=2: also with an activation (post-pass in LDS)
    static const int pipe_on = getenv("ICM_WG_PIPE") ? atoi(getenv("ICM_WG_PIPE")) : 1;
    pg.pipe = pipe_on;
  }
  d.Ca = a->Ca; d.OH = a->OH; d.OW = a->OW; d.act_s = a->act_s;
  d.xcd_order = ((long long)a->N * a->OH * a->OW >= 16384) ? 1 : 0;
  if (icm::g_force_xcd >= 0) d.xcd_order = icm::g_force_xcd;
  d.S = a->stride; d.pad = a->pad; d.ntaps = ntaps; d.tpg = p.tpg;
  d.lgTW = p.lgTW; d.lgTH = p.lgTH; d.lgTI = p.lgTI; d.lgNPX = p.lgNPX;
  d.tiles_x = p.tiles_x; d.tiles_y = p.tiles_y; d.tiles_n = p.tiles_n; d.ntiles = p.ntiles; d.nsplit = p.nsplit;
  d.natile = p.natile; d.nbtile = p.nbtile; d.ngroups = p.ngroups;
  for (int t = 0; t < WG_MAX_TAPS; ++t) d.tapoff[t] = 0;
  for (int kh = 0; kh < a->KH; ++kh)
    for (int kw = 0; kw < a->KW; ++kw) d.tapoff[kh * a->KW + kw] = kh * p.PW + kw;
  {
    const int TWm = (1 << p.lgTW) - 1, THm = (1 << p.lgTH) - 1;
    auto off = [&](int px) {
      const int tx = px & TWm, ty = (px >> p.lgTW) & THm, ti = px >> (p.lgTW + p.lgTH);
      return ti * p.PP + ty * a->stride * p.PW + tx * a->stride;
    };
    for (int kp = 0; kp < 32; ++kp) d.pe[kp] = (2 * kp < (1 << p.lgNPX)) ? off(2 * kp) : 0;
    d.po_h = off(1) - off(0);
  }
  const long long nblk = (long long)p.natile * p.nbtile * p.ngroups * p.nsplit;
  void (*fn)(const WgDesc) = nullptr;
  static const bool dma1_v4_off = [] { const char* e = getenv("ICM_WG_DMA1_V4"); return e && atoi(e) == 0; }();   // measurement only
  bool v4dma = !dma1_v4_off && p.ws >= 11 && p.ws <= 14 && p.lgNPX == 6 && p.lgTW >= 2 && (a->OW % 4) == 0 &&
               (a->gs_bs % 4) == 0 && (a->gb_bs % 4) == 0;
  for (int i = 0; i < n && v4dma; ++i)
    v4dma = ((reinterpret_cast<uintptr_t>(arr[i].gs) | reinterpret_cast<uintptr_t>(arr[i].gb)) & 15) == 0;
  if (p.ws == 11) fn = v4dma ? wgrad_dma1_kernel<6, 6, 2, 2, true> : wgrad_dma1_kernel<6, 6, 2, 2, false>;
  else if (p.ws == 12) fn = v4dma ? wgrad_dma1_kernel<3, 6, 1, 2, true> : wgrad_dma1_kernel<3, 6, 1, 2, false>;
  else if (p.ws == 13) fn = v4dma ? wgrad_dma1_kernel<6, 3, 2, 1, true> : wgrad_dma1_kernel<6, 3, 2, 1, false>;
  else if (p.ws == 14) fn = v4dma ? wgrad_dma1_kernel<3, 3, 1, 1, true> : wgrad_dma1_kernel<3, 3, 1, 1, false>;
  else if (p.ws == 10) fn = wgrad_tap9_kernel<3, 1, 3>;
  else if (p.ws == 8) fn = wgrad_tap9_kernel<3, 3>;
  else if (p.ws == 9) fn = wgrad_tap9_kernel<2, 2>;
  else if (p.ws == 7) fn = wgrad_t33_kernel<3, 3, 1, 1, true>;
  else if (p.ws == 3) fn = wgrad_t33_kernel<3, 3, 1, 1>;
  else if (p.ws == 4) fn = wgrad_t33_kernel<6, 6, 2, 2>;
  else if (p.ws == 5) fn = wgrad_t33_kernel<3, 6, 1, 2>;
  else if (p.ws == 6) fn = wgrad_t33_kernel<6, 3, 2, 1>;
  else if (p.ta == 4) fn = wgrad_kernel<4, 4, 4>;
  else if (p.tb == 2) fn = wgrad_kernel<2, 2, 9>;
  else fn = wgrad_kernel<2, 1, 7>;
  if (p.lds > 64 * 1024 && !ensure_max_lds(reinterpret_cast<const void*>(fn))) return ICM_ERR_LAUNCH;
  hipLaunchKernelGGL(fn, dim3((unsigned)nblk, n), dim3(512), p.lds, stream, d);
  ICM_CHECK_LAUNCH();
  r.Ca = a->Ca; r.Cb = a->Cb; r.ntaps = ntaps; r.nsplit = nslab; r.nsplit_bias = p.nsplit;
  {
    const long long CaCb = (long long)a->Ca * a->Cb;
    bool v4 = ntaps == 1 && (CaCb % 4) == 0;
    for (int i = 0; i < n; ++i) {
      v4 = v4 && ((reinterpret_cast<uintptr_t>(arr[i].ws) & 15) == 0) && ((reinterpret_cast<uintptr_t>(arr[i].dw) & 15) == 0);
      if (r.g[i].dw_ld != a->Cb) v4 = v4 && (a->Cb % 4) == 0 && (r.g[i].dw_ld % 4) == 0;
    }
    r.vec4 = v4 ? 1 : 0;
    r.nwblocks = (int)((CaCb + 63) / 64);   // 16 float4 positions (1x1) or 64 positions (multi-tap) per workgroup
  }
  bool any_bias = false;
  for (int i = 0; i < n; ++i) any_bias |= arr[i].dbias != nullptr;
  const int rblocks = r.nwblocks + (any_bias ? cdiv(a->Ca, 16) : 0);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rblocks, n), dim3(256), 0, stream, r);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}

void icm_debug_force_wgrad_cfg(int variant, int xcd_order) {
  icm::g_force_variant = (variant >= 0 && variant < WG_NVARIANTS) ? variant : -1;
  icm::g_force_xcd = (xcd_order == 0 || xcd_order == 1) ? xcd_order : -1;
}

int icm_conv_wgrad(const icm_wgrad_args* a, void* stream_) {
  return wgrad_grouped(a, 1, (hipStream_t)stream_);
}
int icm_conv_wgrad_grouped(const icm_wgrad_args* arr, int n, void* stream_) {
  return wgrad_grouped(arr, n, (hipStream_t)stream_);
}

}  // extern "C"
