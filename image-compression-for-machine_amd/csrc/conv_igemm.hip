// Implicit-GEMM convolution family on f32 MFMA (v_mfma_f32_32x32x2_f32) for gfx950.
//
// One kernel serves conv fwd, conv dgrad, convT fwd, convT dgrad, 1x1 convs / Linear layers and the
// GDN channel contraction (reference call sites: include/icm_hip.h).  GEMM view per launch:
//     D[co][pixel] = sum_{tap, ci} Wp[tap][ci][co] * patch[ci][pixel + tap]
//   * M = output channels (MFMA A operand, rows), N = pixels (MFMA B operand, columns) so that the
//     32 lanes of a half-wave hold 32 consecutive pixels of one channel plane: NCHW loads and stores
//     are 128-B coalesced with no transposes anywhere;
//   * the input patch of the block's pixel tile (8 input channels per K-chunk, halo included) is
//     staged once into LDS and re-read for every tap (25x reuse for 5x5) -- activations are never
//     im2col-expanded in HBM or L2; stride-2 patches are stored column-parity-split so the B-fragment
//     ds_read_b32 is bank-conflict free;
//   * weights are pre-packed in MFMA A-fragment order (icm_pack_weights) and stream straight from
//     L2 into registers as one coalesced 16-B load per lane per (tap, 8-channel chunk, 32-co tile);
//   * "scatter" forms (convT fwd / conv dgrad) run as stride^2 output-parity classes, each a dense
//     stride-1 gather with its own tap subset -- no zero-stuffing, no atomics;
//   * pointwise neighbours are fused: operand activation on the LDS staging path (virtual GELU,
//     x^2 for GDN) and the epilogues listed in icm_hip.h (bias, residual, GDN rsqrt, GELU', LRP tanh,
//     PixelShuffle store, gradient accumulation).
#include <cstdlib>
#include <vector>
#include "conv_common.h"

namespace icm {

// 512 threads: waves 0-3 issue MFMAs only (B fragments from LDS, A fragments = packed weights from L2);
// waves 4-7 are loaders that stage the NEXT K-chunk's halo patch into the other LDS buffer meanwhile.
// MFMA and VALU/VMEM are separate pipes per SIMD, so with one MFMA wave and one loader wave per SIMD the
// staging cost disappears behind the 64-cycle v_mfma_f32_32x32x2_f32 issue interval.
// KS > 1: intra-workgroup split-K for small problems (few output tiles, deep K): KS MFMA waves share one output
// tile, each taking every KS-th (8-channel group, tap) sub-step; partial accumulators are summed through LDS.
template <int WCO, int WPX, int TCO, int TPX, int KS = 1>
__global__ __launch_bounds__(512, (TCO * TPX <= 3) ? 4 : 2) void conv_igemm_kernel(const ConvDesc d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  static_assert(WCO * WPX * KS == 4, "4 MFMA waves per workgroup");
  constexpr int BCO_T = WCO * TCO;
  // small tiles: pipeline steps of TS (8-channel group, tap) sub-steps so that a step carries >= 8-12 MFMAs
  constexpr int TS = (TCO * TPX == 1) ? 3 : ((TCO * TPX == 2) ? 2 : 1);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool loader = wave >= 4;
  const ConvPtrs P = d.g[blockIdx.y];
  const PatchGeom& pg = d.pg;

  // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so hardware id b
  // runs on XCD b % 8.  Give every XCD a CONTIGUOUS run of logical tiles: the co-blocks of one pixel tile and the
  // vertically adjacent tiles (which share most of their halo rows) then meet in the same L2.
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
  const int S_in = pg.S == 2 ? 2 : 1;
  const int iyb = oy0 * S_in + d.iy0, ixb = ox0 * S_in + d.ix0;
  const int bufsz = d.ckm * 8 * pg.CS;
  const int nchunks = (d.nchunks8 + d.ckm - 1) / d.ckm;

  if (loader) {
    const int lw = __builtin_amdgcn_readfirstlane(wave) - 4;
    PlaneMap pm;

    if (pg.dma) {
      // activation-free operand with a linear patch layout: LDS-DMA (global_load_lds_dword).  No VGPR holds the data,
      // so a loader wave keeps every element of its channels in flight at once instead of 12-16 loads per lane; the
      // wave drains vmcnt before the barrier that publishes the buffer (barriers do not wait for DMAs)
      plane_map_init(pm, pg, n0, iyb, ixb, lane);
      stage_planes_dma(P.x, pm, pg, 0, min(d.ckm, d.nchunks8) * 8, smem, lw, lane);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      for (int chunk = 0; chunk < nchunks; ++chunk) {
        if (chunk + 1 < nchunks) {
          const int c8 = (chunk + 1) * d.ckm;
          stage_planes_dma(P.x, pm, pg, c8 * 8, min(d.ckm, d.nchunks8 - c8) * 8, smem + ((chunk + 1) & 1) * bufsz, lw, lane);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
      }
      return;
    }
    if (pg.vec4) {
      plane_map_init_v4(pm, pg, n0, iyb, ixb, lane);
      stage_planes_v4<8>(P.x, pm, pg, 0, min(d.ckm, d.nchunks8) * 8, smem, lw);
      __syncthreads();
      for (int chunk = 0; chunk < nchunks; ++chunk) {
        if (chunk + 1 < nchunks) {
          const int c8 = (chunk + 1) * d.ckm;
          stage_planes_v4<8>(P.x, pm, pg, c8 * 8, min(d.ckm, d.nchunks8 - c8) * 8, smem + ((chunk + 1) & 1) * bufsz,
                             lw);
        }
        __syncthreads();
      }
      return;
    }
    plane_map_init(pm, pg, n0, iyb, ixb, lane);
    stage_planes<12>(P.x, pm, pg, 0, min(d.ckm, d.nchunks8) * 8, smem, lw);
    __syncthreads();
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      if (chunk + 1 < nchunks) {
        const int c8 = (chunk + 1) * d.ckm;
        stage_planes<12>(P.x, pm, pg, c8 * 8, min(d.ckm, d.nchunks8 - c8) * 8, smem + ((chunk + 1) & 1) * bufsz, lw);
      }
      __syncthreads();
    }
    return;
  }

  // ------------------------------------------------------------------ MFMA waves
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);   // wave-uniform: step bookkeeping stays on the scalar unit
  const int wk = wave_u % KS, wsp = wave_u / KS;   // K-split index, spatial wave index
  const int wco = wsp / WPX, wpx = wsp % WPX;
  const int h = lane >> 5, l31 = lane & 31;
  const int TWm = (1 << d.lgTW) - 1, THm = (1 << d.lgTH) - 1;
  const int rowmul = S_in * pg.PWrow;
  int boff[TPX];
#pragma unroll
  for (int tp = 0; tp < TPX; ++tp) {
    const int p = (wpx * TPX + tp) * 32 + l31;
    const int tx = p & TWm, ty = (p >> d.lgTW) & THm, ti = p >> (d.lgTW + d.lgTH);
    boff[tp] = h * pg.CS + ti * pg.PP + ty * rowmul + tx;
  }
  f32x16 acc[TCO][TPX];
#pragma unroll
  for (int a = 0; a < TCO; ++a)
#pragma unroll
    for (int b = 0; b < TPX; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

  const char* wbase = reinterpret_cast<const char*>(P.wp);   // wave-uniform: weight loads are saddr + 32-bit voffset
  const int cot0 = cb * BCO_T + wco * TCO;
  unsigned wl[TCO];  // per-lane byte offset of the fragment of co tile a inside one (chunk8, tap) step
#pragma unroll
  for (int a = 0; a < TCO; ++a) wl[a] = (unsigned)(min(cot0 + a, d.ncot - 1) * 64 + lane) * 16u;
  const long long qstride = (long long)d.ncot * 64 * 16;   // bytes between consecutive (chunk8, tap) steps
  const int Qtot = d.nchunks8 * d.ntaps;
  {
    // LDS offset of sub-step q of a chunk (8-channel group q / ntaps, tap q % ntaps) lives in lane q of one VGPR:
    // v_readlane replaces the scalar (group, tap) bookkeeping; the host guarantees ckm * ntaps <= 64
    int step_lane;
    {
      const int sq = lane / d.ntaps, tq = lane - sq * d.ntaps;
      step_lane = sq * 8 * pg.CS + d.tapoff[tq];
    }
    // Pipeline over steps of TS (8-channel group, tap) sub-steps (small tiles take TS > 1 so that a step carries
    // >= 8-12 MFMAs).  A fragments (packed weights) come from L2, whose latency exceeds one step of a small tile:
    // they are prefetched TWO steps ahead into two register sets used alternately (the step loop is unrolled by two,
    // chunks are padded to an even number of steps with zeroed B fragments).  B fragments come from LDS one step ahead.
    auto chunk_nq = [&](int c) { return min(d.ckm, d.nchunks8 - c * d.ckm) * d.ntaps; };
    // scalar pointer to the packed weights of sub-step u of step `st` of chunk `c` (st may run past the chunk:
    // first steps of the next chunk; past the end: clamped, never used)
    // sub-steps of a chunk taken by this wave: q = wk, wk + KS, ...; padded to an even number of steps
    auto steps2 = [&](int nq_c) { return ((((nq_c - wk + KS - 1) / KS) + TS - 1) / TS + 1) & ~1; };
    auto wptr = [&](int c, int st, int u) -> const char* {
      int nq_c = chunk_nq(c);
      const int ns2 = steps2(nq_c);
      if (st >= ns2) {
        st -= ns2;
        c += 1;
        nq_c = (c < nchunks) ? chunk_nq(c) : 1;
      }
      const int qw = min(c * d.ckm * d.ntaps + min((st * TS + u) * KS + wk, nq_c - 1), Qtot - 1);
      return wbase + qw * qstride;
    };
    f32x4 aq0[TS][TCO], aq1[TS][TCO];
#pragma unroll
    for (int u = 0; u < TS; ++u)
#pragma unroll
      for (int a = 0; a < TCO; ++a) {
        aq0[u][a] = *reinterpret_cast<const f32x4*>(wptr(0, 0, u) + wl[a]);
        aq1[u][a] = *reinterpret_cast<const f32x4*>(wptr(0, 1, u) + wl[a]);
      }

    __syncthreads();  // patch of chunk 0 staged
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      const float* cur = smem + (chunk & 1) * bufsz;
      const int nq = chunk_nq(chunk);
      const int nsteps2 = steps2(nq);
      float bv_n[TS][4][TPX];
      float ok_n[TS];   // 1 / 0 (wave-uniform): sub-steps past the chunk contribute zero; applied when bv_n is consumed
      // B fragments of sub-step u of step st of this chunk into bv_n
      auto bload = [&](int st, int u) {
        const int qq = (st * TS + u) * KS + wk;
        const int qc = min(qq, nq - 1);
        ok_n[u] = (qq < nq) ? 1.0f : 0.0f;
        const float* bp = cur + __builtin_amdgcn_readlane(step_lane, qc);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int tp = 0; tp < TPX; ++tp) bv_n[u][j][tp] = bp[boff[tp] + 2 * j * pg.CS];
      };
      // One step with register set `aq`.  Source order IS the issue order (sched_barrier pins each group): the four
      // k-pairs of a weight fragment issue back to back, then -- while the matrix pipe works them off -- the LDS
      // reads of the next step's B fragments and the refill of this fragment with the weights two steps ahead.
      auto half_step = [&](f32x4 (&aq)[TS][TCO], int st) {
        float bv[TS][4][TPX];
#pragma unroll
        for (int u = 0; u < TS; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int tp = 0; tp < TPX; ++tp) bv[u][j][tp] = bv_n[u][j][tp] * ok_n[u];
        const int stn = min(st + 1, nsteps2 - 1);
#pragma unroll
        for (int u = 0; u < TS; ++u) {
          const char* sp = wptr(chunk, st + 2, u);
#pragma unroll
          for (int a = 0; a < TCO; ++a) {
#pragma unroll
            for (int tp = 0; tp < TPX; ++tp)
#pragma unroll
              for (int j = 0; j < 4; ++j)
                acc[a][tp] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[u][a][j], bv[u][j][tp], acc[a][tp], 0, 0, 0);
            if (a == 0) bload(stn, u);
            aq[u][a] = *reinterpret_cast<const f32x4*>(sp + wl[a]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      };
#pragma unroll
      for (int u = 0; u < TS; ++u) bload(0, u);
      for (int st = 0; st < nsteps2; st += 2) {
        half_step(aq0, st);
        half_step(aq1, st + 1);
      }
      __syncthreads();
    }
  }

  if constexpr (KS > 1) {
    // split-K reduction: the staging buffers are free after the last chunk barrier (the loaders have left: ended
    // waves do not take part in s_barrier).  Waves wk > 0 park their accumulators in LDS, wave wk == 0 adds them.
    static_assert(TCO == 1 && TPX == 1, "split-K configurations hold one tile per wave");
    float* red = smem + (wsp * (KS - 1)) * 1024;
    if (wk > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[(wk - 1) * 1024 + r * 64 + lane] = acc[0][0][r];
    }
    __syncthreads();
    if (wk > 0) return;
#pragma unroll
    for (int k = 0; k < KS - 1; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][0][r] += red[k * 1024 + r * 64 + lane];
  }
  // ---- epilogue: D[row = co][col = pixel]; lane holds column l31, rows (r&3)+8*(r>>2)+4*h
  {
    int pn[TPX], poy[TPX], pox[TPX];
    bool pv[TPX];
#pragma unroll
    for (int tp = 0; tp < TPX; ++tp) {
      const int p = (wpx * TPX + tp) * 32 + l31;
      const int tx = p & TWm, ty = (p >> d.lgTW) & THm, ti = p >> (d.lgTW + d.lgTH);
      const int oyv = oy0 + ty, oxv = ox0 + tx;
      pn[tp] = n0 + ti;
      pv[tp] = (pn[tp] < d.pg.N) && (oyv < d.OHv) && (oxv < d.OWv);
      poy[tp] = oyv * d.out_sy + d.out_oy;
      pox[tp] = oxv * d.out_sx + d.out_ox;
    }
    epilogue_dispatch<TCO, TPX>(d, P, acc, cot0, h, pn, poy, pox, pv);
  }
}

// ------------------------------------------------------------------------------------------------
// weight packing: wp[((chunk*ntaps + t)*ncot + cot)*64 + lane][j] = W[co = cot*32 + (lane&31)]
//                                                                    [ci = chunk*8 + 2*j + (lane>>5)][tap t]
struct PackDesc {
  const float* w;
  float* wp;
  int Co, Ci, KHW, src_out_major, ntaps, ncot, nchunks, nonneg;
  float bound, pedestal;
  int src_ld, src_off;          // sub-matrix of a wider weight: row length / first entry of the inner matrix index
  int dst_ncot, dst_cot_off;    // concatenation along GEMM-M: tiles per step of the destination, first tile of this job
  int wino;                     // 0: spatial taps; 1 / 2: Winograd F(2x2,3x3) weights U = G g G^T of the 3x3 kernel (2: of
                                // the 180-degree rotated kernel = input-gradient orientation): 16 "taps" = transform points
  short tapidx[ICM_MAX_TAPS];
};

// One unit = (32-co tile, 8-ci chunk): the 8 x 32 x KHW source block is read with coalesced row segments into LDS
// (row stride padded to an odd number of words: conflict-free transposed reads), then written out in fragment order
// with one 16-byte store per lane per tap.  The canonical layouts differ only in which index is the contiguous row:
// W[co][ci][khw] (src_out_major: 32 rows of 8*KHW) or W[ci][co][khw] (8 rows of 32*KHW).
#define ICM_PACK_LDS (8 * 32 * ICM_MAX_TAPS + 64)
__device__ __forceinline__ void pack_unit(const PackDesc& d, int cot, int chunk, float* lds) {
  const int KHW = d.KHW, tid = threadIdx.x;
  const int co0 = cot * 32, ci0 = chunk * 8;
  int rows, rlen, rstride;
  if (d.src_out_major) { rows = 32; rlen = 8 * KHW; } else { rows = 8; rlen = 32 * KHW; }
  rstride = rlen | 1;
  // batches of 8 independent loads per thread (a plain load -> LDS-store loop waits for every load before the next
  // one is issued: 25 dependent round trips per 5x5 unit made the packing latency-bound at a fifth of its traffic rate)
  const int total = rows * rlen;
  for (int e0 = tid; e0 < total; e0 += 8 * (int)blockDim.x) {
    float v[8];
    int lo[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * (int)blockDim.x;
      v[u] = 0.0f;
      lo[u] = -1;
      if (e < total) {
        const int r = e / rlen, c = e - r * rlen;
        lo[u] = r * rstride + c;
        if (d.src_out_major) {
          const int co = co0 + r, ci = ci0 + c / KHW;
          if (co < d.Co && ci < d.Ci) v[u] = d.w[((long long)co * d.src_ld + d.src_off + ci0) * KHW + c];
        } else {
          const int ci = ci0 + r, co = co0 + c / KHW;
          if (ci < d.Ci && co < d.Co) v[u] = d.w[((long long)ci * d.src_ld + d.src_off + co0) * KHW + c];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (lo[u] >= 0) {
        float x = v[u];
        if (d.nonneg) {
          x = fmaxf(x, d.bound);
          x = x * x - d.pedestal;
        }
        lds[lo[u]] = x;
      }
    }
  }
  __syncthreads();
  f32x4* out = reinterpret_cast<f32x4*>(d.wp);
  for (int e = tid; e < d.ntaps * 64; e += blockDim.x) {
    const int t = e >> 6, lane = e & 63;
    const int col = lane & 31, hh = lane >> 5, k = d.tapidx[t];
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int cil = 2 * j + hh;
      const float* gsrc = d.src_out_major ? lds + col * rstride + cil * KHW : lds + cil * rstride + col * KHW;
      float v;
      if (d.wino) {
        // U[a][b] = sum_pq G[a][p] g[p][q] G[b][q],  G = [[1,0,0],[1/2,1/2,1/2],[1/2,-1/2,1/2],[0,0,1]]   (t = 4 a + b)
        const int wa = t >> 2, wb = t & 3;
        float row[3];   // (g G^T)[p][wb]
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const float g0 = gsrc[d.wino == 2 ? (2 - p) * 3 + 2 : p * 3 + 0];
          const float g1 = gsrc[d.wino == 2 ? (2 - p) * 3 + 1 : p * 3 + 1];
          const float g2 = gsrc[d.wino == 2 ? (2 - p) * 3 + 0 : p * 3 + 2];
          row[p] = wb == 0 ? g0 : (wb == 3 ? g2 : (wb == 1 ? 0.5f * ((g0 + g2) + g1) : 0.5f * ((g0 + g2) - g1)));
        }
        v = wa == 0 ? row[0] : (wa == 3 ? row[2] : (wa == 1 ? 0.5f * ((row[0] + row[2]) + row[1]) : 0.5f * ((row[0] + row[2]) - row[1])));
      } else {
        v = gsrc[k];
      }
      const bool valid = (co0 + col < d.Co) && (ci0 + cil < d.Ci);
      o[j] = valid ? v : 0.0f;   // padded rows / channels are exact zeros (also under the nonneg transform)
    }
    out[((long long)(chunk * d.ntaps + t) * d.dst_ncot + d.dst_cot_off + cot) * 64 + lane] = o;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void pack_weights_kernel(const PackDesc d) {
  __shared__ float lds[ICM_PACK_LDS];
  const int units = d.ncot * d.nchunks;
  for (int u = blockIdx.x; u < units; u += gridDim.x) pack_unit(d, u % d.ncot, u / d.ncot, lds);
}

#define ICM_PACK_NB 24
struct PackMulti {
  PackDesc g[ICM_PACK_NB];
};
__global__ __launch_bounds__(256) void pack_weights_multi_kernel(const PackMulti m) {
  __shared__ float lds[ICM_PACK_LDS];
  const PackDesc& d = m.g[blockIdx.y];
  const int units = d.ncot * d.nchunks;
  for (int u = blockIdx.x; u < units; u += gridDim.x) pack_unit(d, u % d.ncot, u / d.ncot, lds);
}

// ---------------------------------------------------------------------------------- host planning
struct Tap {
  int kidx;    // kh*KW + kw in the canonical weight
  int dy, dx;  // patch-relative offsets
};
struct ConvClass {
  int cy, cx;      // output parity (scatter) -- 0,0 for gather
  int iy0, ix0;    // input coordinate of patch row/col 0 for virtual output 0
  int ey, ex;      // patch extents contributed by taps (max dy + 1)
  std::vector<Tap> taps;
};

// Taps of one axis for output parity c of a scatter conv: k with (c + pad - k) % s == 0, input offset (c+pad-k)/s
static void axis_taps(int K, int s, int pad, int c, std::vector<int>& ks, std::vector<int>& dis) {
  for (int k = 0; k < K; ++k) {
    const int v = c + pad - k;
    if (((v % s) + s) % s != 0) continue;
    ks.push_back(k);
    dis.push_back(v >= 0 ? v / s : -((-v) / s));
  }
}

static std::vector<ConvClass> build_classes(int KH, int KW, int stride, int pad, int transposed) {
  std::vector<ConvClass> out;
  if (!transposed) {
    ConvClass c;
    c.cy = c.cx = 0;
    c.iy0 = c.ix0 = -pad;
    c.ey = KH;
    c.ex = KW;
    for (int kh = 0; kh < KH; ++kh)
      for (int kw = 0; kw < KW; ++kw) c.taps.push_back({kh * KW + kw, kh, kw});
    out.push_back(c);
    return out;
  }
  for (int cy = 0; cy < stride; ++cy)
    for (int cx = 0; cx < stride; ++cx) {
      std::vector<int> khs, dys, kws, dxs;
      axis_taps(KH, stride, pad, cy, khs, dys);
      axis_taps(KW, stride, pad, cx, kws, dxs);
      ConvClass c;
      c.cy = cy;
      c.cx = cx;
      int miny = 0, minx = 0, maxy = 0, maxx = 0;
      for (size_t i = 0; i < dys.size(); ++i) {
        miny = i ? std::min(miny, dys[i]) : dys[i];
        maxy = i ? std::max(maxy, dys[i]) : dys[i];
      }
      for (size_t i = 0; i < dxs.size(); ++i) {
        minx = i ? std::min(minx, dxs[i]) : dxs[i];
        maxx = i ? std::max(maxx, dxs[i]) : dxs[i];
      }
      c.iy0 = miny;
      c.ix0 = minx;
      c.ey = maxy - miny + 1;
      c.ex = maxx - minx + 1;
      for (size_t a = 0; a < khs.size(); ++a)
        for (size_t b = 0; b < kws.size(); ++b)
          c.taps.push_back({khs[a] * KW + kws[b], dys[a] - miny, dxs[b] - minx});
      out.push_back(c);
    }
  return out;
}

struct KernelCfg {
  int wco, wpx, tco, tpx;
  void (*fn)(const ConvDesc);
  int ks = 1;
};
static const KernelCfg kCfgs[] = {
    {1, 4, 3, 2, conv_igemm_kernel<1, 4, 3, 2>},  // 96 co x 256 px
    {1, 4, 6, 1, conv_igemm_kernel<1, 4, 6, 1>},  // 192 x 128
    {1, 4, 2, 2, conv_igemm_kernel<1, 4, 2, 2>},  // 64 x 256
    {1, 4, 1, 2, conv_igemm_kernel<1, 4, 1, 2>},  // 32 x 256
    {1, 4, 3, 1, conv_igemm_kernel<1, 4, 3, 1>},  // 96 x 128
    {1, 4, 5, 1, conv_igemm_kernel<1, 4, 5, 1>},  // 160 x 128
    {2, 2, 2, 1, conv_igemm_kernel<2, 2, 2, 1>},  // 128 x 64
    {2, 2, 1, 1, conv_igemm_kernel<2, 2, 1, 1>},  // 64 x 64
    {4, 1, 1, 1, conv_igemm_kernel<4, 1, 1, 1>},  // 128 x 32
    {1, 4, 1, 1, conv_igemm_kernel<1, 4, 1, 1>},  // 32 x 128
    {2, 1, 1, 1, conv_igemm_kernel<2, 1, 1, 1, 2>, 2},  // 64 x 32, K split in 2
    {1, 2, 1, 1, conv_igemm_kernel<1, 2, 1, 1, 2>, 2},  // 32 x 64, K split in 2
    {1, 1, 1, 1, conv_igemm_kernel<1, 1, 1, 1, 4>, 4},  // 32 x 32, K split in 4
};
static int g_force_cfg = -1;

static int g_force_1x1 = getenv("ICM_CONV_1X1") ? atoi(getenv("ICM_CONV_1X1")) : -1;   // -1 auto, 0 never, 1 whenever eligible

struct Geometry {
  int lgTW, lgTH, lgTI, PH, PW, PWh, PWrow, PP, CS, tiles_x, tiles_y, tiles_n, ckm;
  size_t lds_bytes;
};

static Geometry make_geometry(int bpx, int OHv, int OWv, int N, int S, int ey, int ex, int nchunks8, int ntaps,
                              int tiles_per_wave) {
  Geometry g;
  const int lgB = ceil_log2(bpx);
  g.lgTW = std::min(std::min(5, lgB), ceil_log2(OWv));
  g.lgTH = std::min(lgB - g.lgTW, ceil_log2(OHv));
  g.lgTI = lgB - g.lgTW - g.lgTH;
  const int TW = 1 << g.lgTW, TH = 1 << g.lgTH, TI = 1 << g.lgTI;
  g.PW = (TW - 1) * S + ex;
  g.PH = (TH - 1) * S + ey;
  g.PWh = (g.PW + 1) / 2;
  g.PWrow = (S == 2) ? 2 * g.PWh : g.PW;
  g.PP = g.PH * g.PWrow;
  g.CS = TI * g.PP;
  g.tiles_x = cdiv(OWv, TW);
  g.tiles_y = cdiv(OHv, TH);
  g.tiles_n = cdiv(N, TI);
  // channels per barrier: >= ~256 MFMAs per MFMA wave between barriers (16k cycles) so that the loaders' L2/HBM
  // round trips and the barrier itself stay hidden, within 64 KB of LDS for the two buffers
  int ckm = cdiv(256, ntaps * 4 * tiles_per_wave);
  ckm = std::max(1, std::min(ckm, nchunks8));
  ckm = std::max(1, std::min(ckm, 64 / std::max(1, ntaps)));   // step table = one VGPR (64 lanes)
  while (ckm > 1 && (size_t)2 * ckm * 8 * g.CS * sizeof(float) > 64 * 1024) --ckm;
  g.ckm = ckm;
  g.lds_bytes = (size_t)2 * ckm * 8 * g.CS * sizeof(float);
  return g;
}

static int run_class(const icm_conv_args* arr, int ngroups, const ConvClass& cls, long long wp_off, int S_in,
                     int out_s, hipStream_t stream) {
  const icm_conv_args& a = arr[0];
  const int OHv = a.transposed ? cdiv(a.OH - cls.cy, out_s) : a.OH;
  const int OWv = a.transposed ? cdiv(a.OW - cls.cx, out_s) : a.OW;
  if (OHv <= 0 || OWv <= 0) return ICM_OK;
  const int ncot = cdiv(a.Cout, 32), nchunks8 = cdiv(a.Cin, 8);
  const int ntaps = (int)cls.taps.size();
  if (ntaps > ICM_MAX_TAPS) return ICM_ERR_UNSUPPORTED;
  if (ntaps == 1 && S_in == 1 && out_s == 1 && cls.iy0 == 0 && cls.ix0 == 0 && a.x_seg_len == 0) {
    const int rc = (g_force_cfg >= 0) ? -1 : run_conv1x1(arr, ngroups, wp_off, g_force_1x1, stream);
    if (rc >= 0) return rc;
  }

  // pick the tile configuration: time ~ rounds x (co-resident workgroups share the MFMA pipes: occ x MFMAs per wave /
  // efficiency + one fixed prologue/epilogue overhead per round), rounds = ceil(workgroups / (256 CUs x occ)); occ = 2
  // for the configurations whose kernels fit 128 VGPRs (launch bounds above), efficiencies measured on MI355X
  // (tools/tune_conv.py; profiles/r01_tune_conv_v7.txt)
  static const double kEff[] = {0.75, 1.00, 0.80, 0.60, 1.10, 0.65, 0.80, 0.80, 0.75, 0.70, 0.70, 0.70, 0.65};
  static const int kOcc[] = {1, 1, 1, 2, 2, 1, 2, 2, 2, 2, 2, 2, 2};
  static const double kCoResBoost = getenv("ICM_CONV_BOOST") ? atof(getenv("ICM_CONV_BOOST")) : 1.25;
  int best = -1, best1 = -1;
  double best_cost = 1e300, best1_cost = 1e300;
  Geometry bg{}, bg1{};
  const int ncfg = (int)(sizeof(kCfgs) / sizeof(kCfgs[0]));
  for (int i = 0; i < ncfg; ++i) {
    if (g_force_cfg >= 0 && g_force_cfg < 100 && i != g_force_cfg) continue;
    const KernelCfg& c = kCfgs[i];
    const int bpx = c.wpx * c.tpx * 32, bco_t = c.wco * c.tco;
    Geometry g = make_geometry(bpx, OHv, OWv, a.N, S_in, cls.ey, cls.ex, nchunks8, ntaps, c.tco * c.tpx);
    if (c.ks > 1) g.lds_bytes = std::max<size_t>(g.lds_bytes, (size_t)(c.ks - 1) * c.wco * c.wpx * 4096);
    if (g.lds_bytes > 160 * 1024) continue;
    if ((1 << g.lgTI) * g.PH * g.PW > ICM_MAXJ * 64) continue;   // PlaneMap capacity
    const long long blocks = (long long)cdiv(ncot, bco_t) * g.tiles_x * g.tiles_y * g.tiles_n * ngroups;
    const int occ_max = (g.lds_bytes * 2 <= 160 * 1024) ? kOcc[i] : 1;
    const int occ = (int)std::min<long long>(occ_max, (blocks + 255) / 256);   // workgroups actually co-resident
    const double rounds = (double)((blocks + 256 * occ - 1) / (256 * occ));
    const double mfma = (double)c.tco * c.tpx * nchunks8 * ntaps * 4 / c.ks + (c.ks > 1 ? 24.0 : 0.0);   // per MFMA wave
    // two co-resident workgroups interleave their MFMA streams: the issue gaps of one wave per SIMD are filled
    // (the interleaving gain fades for long K loops, whose steady state already keeps the matrix pipe busy)
    const double boost = occ > 1 ? 1.0 + (kCoResBoost - 1.0) * std::min(1.0, 1500.0 / mfma) : 1.0;
    double cost = rounds * (occ * mfma / (kEff[i] * boost) + 200.0);
    // a configuration whose co-tile covers all output channels stages every halo patch exactly once
    if (ntaps >= 9 && (long long)OHv * OWv * a.N >= 16384 && cdiv(ncot, bco_t) == 1 && cost < best1_cost) {
      best1_cost = cost;
      best1 = i;
      bg1 = g;
    }
    if (cost < best_cost) {
      best_cost = cost;
      best = i;
      bg = g;
    }
  }
  if (best < 0) return ICM_ERR_UNSUPPORTED;
  // halo convolutions with many pixels: within 20 % of the cheapest estimate, take the single-pass-over-activations
  // tiling (every extra co-block re-stages the whole halo patch from L2 / HBM: 1.85x FETCH_SIZE measured on g_a.2
  // for 3 % of MFMA time)
  if (best1 >= 0 && best1_cost <= 1.2 * best_cost) {
    best = best1;
    bg = bg1;
  }
  const KernelCfg& c = kCfgs[best];
  // Latency-bound launches: the 8-wave K-split kernel (conv_ks8.hip).  Needs stride-1 sampling, no operand activation,
  // the GDN / AXPY2 epilogues excluded (they belong to pointwise launches), and the step table in one VGPR
  // (ckm * ntaps <= 64, as here).  It runs one workgroup per CU, so it is taken only when its grid is ONE nearly full
  // round of the chip (160..256 workgroups): there it cuts the critical path of the serial slice-chain layers by ~15 %
  // (224 -> 176 @16x16: 60 -> 51 us, 480 -> 224: 116 -> 100 us); applied to every small launch it lost 1.3 % on the
  // step (narrow outputs 64 -> 32: 17 -> 29 us, half-empty grids, second-round tails; same-box A/B).
  int ks8_tco = 0;
  {
    static const int ks8_on = getenv("ICM_CONV_KS8") ? atoi(getenv("ICM_CONV_KS8")) : 1;
    static const long long ks8_max = getenv("ICM_CONV_KS8_MAXWG") ? atoll(getenv("ICM_CONV_KS8_MAXWG")) : 256;
    static const long long ks8_min = getenv("ICM_CONV_KS8_MINWG") ? atoll(getenv("ICM_CONV_KS8_MINWG")) : 160;
    const bool epi_ok = a.epi == ICM_EPI_NONE || a.epi == ICM_EPI_RES || a.epi == ICM_EPI_RES_GELU ||
                        a.epi == ICM_EPI_MUL_DGELU || a.epi == ICM_EPI_LRP || a.epi == ICM_EPI_RES_MUL_DGELU;
    if ((ks8_on && g_force_cfg < 0 || g_force_cfg == 100 || g_force_cfg == 101) && S_in == 1 && ntaps > 1 &&
        a.pro_act == ICM_ACT_NONE && epi_ok && a.x_seg_len == 0) {
      const int tco = g_force_cfg == 101 ? 1 : (g_force_cfg == 100 ? 2 : (ncot >= 2 ? 2 : 1));
      const int bpx = tco == 2 ? 64 : 128;
      Geometry g = make_geometry(bpx, OHv, OWv, a.N, S_in, cls.ey, cls.ex, nchunks8, ntaps, 4);
      // more K per barrier than the staged kernel needs: every wave should find >= 2 of its sub-steps in a chunk
      int ckm = std::max(1, std::min(nchunks8, 64 / ntaps));
      while (ckm > 1 && (size_t)2 * ckm * 8 * g.CS * sizeof(float) > 96 * 1024) --ckm;
      g.ckm = ckm;
      g.lds_bytes = std::max((size_t)2 * ckm * 8 * g.CS * sizeof(float), (size_t)8 * 4 * 16 * 64 * sizeof(float));
      const long long blocks = (long long)cdiv(ncot, tco) * g.tiles_x * g.tiles_y * g.tiles_n * ngroups;
      const bool fits = g.lds_bytes <= 160 * 1024 && (1 << g.lgTI) * g.PH * g.PW <= ICM_MAXJ * 64 && g.PWrow == g.PW &&
                        g.PP == g.PH * g.PW;
      if (fits && (g_force_cfg >= 100 || (blocks <= ks8_max && blocks >= ks8_min && tco == 2))) {
        ks8_tco = tco;
        bg = g;
      }
    }
  }

  ConvDesc d{};
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
  PatchGeom& pg = d.pg;
  pg.PW = bg.PW; pg.PH = bg.PH; pg.PWrow = bg.PWrow; pg.PWh = bg.PWh; pg.PP = bg.PP; pg.CS = bg.CS; pg.S = S_in;
  pg.TIPH = (1 << bg.lgTI) * bg.PH;
  pg.dPW = make_fastdiv((uint32_t)bg.PW);
  pg.dTIPH = make_fastdiv((uint32_t)pg.TIPH);
  pg.dPH = make_fastdiv((uint32_t)bg.PH);
  pg.H = a.H; pg.W = a.W; pg.N = a.N; pg.C = a.Cin; pg.act = a.pro_act; pg.bs = a.x_bs;
  pg.seg_len = a.x_seg_len; pg.seg_gap = a.x_seg_len ? a.x_seg_gap : 0;
  pg.dseg = make_fastdiv((uint32_t)std::max(1, a.x_seg_len));
  {
    const int TW = 1 << bg.lgTW;
    const int plane4 = (1 << bg.lgTI) * bg.PH * (bg.PW / 4);
    bool v4 = S_in == 1 && cls.ex == 1 && cls.ey == 1 && cls.ix0 == 0 && (bg.PW % 4) == 0 && (TW % 4) == 0 &&
              (a.W % 4) == 0 && (a.x_bs % 4) == 0 && ((long long)a.H * a.W % 4) == 0 && plane4 <= 4 * 64 &&
              (bg.CS % 4) == 0 && (bg.PWrow % 4) == 0 && (bg.PP % 4) == 0;
    for (int gi = 0; gi < ngroups; ++gi) v4 = v4 && ((reinterpret_cast<uintptr_t>(arr[gi].x) & 15) == 0);
    if (a.x_seg_len) v4 = false;   // (the packed vec4 form moves several channels per instruction: keep segments simple)
    pg.vec4 = v4 ? 1 : 0;
    // LDS-DMA staging: no activation to apply, linear patch layout (stride-1 input sampling: no column-parity split),
    // not the 16-byte halo-free path (4x the bytes per instruction)
    static const int dma_on = getenv("ICM_CONV_DMA") ? atoi(getenv("ICM_CONV_DMA")) : 1;
    pg.dma = (dma_on && !v4 && S_in == 1 && a.pro_act == ICM_ACT_NONE && bg.PWrow == bg.PW && bg.PP == bg.PH * bg.PW) ? 1 : 0;
    pg.pipe = 0;
    set_v4_pack(pg);
  }
  d.Cout = a.Cout;
  d.ps2 = a.pixel_shuffle == 2;
  d.OHf = d.ps2 ? a.OH * 2 : a.OH;
  d.OWf = d.ps2 ? a.OW * 2 : a.OW;
  d.OHv = OHv; d.OWv = OWv;
  d.out_sy = d.out_sx = out_s;
  d.out_oy = cls.cy; d.out_ox = cls.cx;
  d.iy0 = cls.iy0; d.ix0 = cls.ix0;
  d.ntaps = ntaps;
  d.lgTW = bg.lgTW; d.lgTH = bg.lgTH; d.lgTI = bg.lgTI;
  d.tiles_x = bg.tiles_x; d.tiles_y = bg.tiles_y; d.tiles_n = bg.tiles_n;
  d.ncot = ncot; d.nchunks8 = nchunks8; d.ckm = bg.ckm; d.ncb = cdiv(ncot, ks8_tco ? ks8_tco : c.wco * c.tco);
  d.epi = a.epi; d.accum = a.accum;
  for (int t = 0; t < ICM_MAX_TAPS; ++t) d.tapoff[t] = 0;
  for (int t = 0; t < ntaps; ++t) {
    const Tap& tp = cls.taps[t];
    const int col = (S_in == 2) ? ((tp.dx & 1) * bg.PWh + (tp.dx >> 1)) : tp.dx;
    d.tapoff[t] = tp.dy * bg.PWrow + col;
  }
  const long long nblk = (long long)d.ncb * d.tiles_x * d.tiles_y * d.tiles_n;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return ICM_ERR_ARG;
  if (ks8_tco) {
    d.pg.vec4 = 0;
    d.pg.dma = 1;
    return launch_conv_ks8(d, ks8_tco, nblk, ngroups, bg.lds_bytes, stream);
  }
  if (bg.lds_bytes > 64 * 1024 && !ensure_max_lds(reinterpret_cast<const void*>(c.fn))) return ICM_ERR_LAUNCH;
  hipLaunchKernelGGL(c.fn, dim3((unsigned)nblk, ngroups, 1), dim3(512), bg.lds_bytes, stream, d);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}

static int validate(const icm_conv_args& a) {
  if (!a.x || !a.wp || !a.y) return ICM_ERR_ARG;
  if (a.N <= 0 || a.Cin <= 0 || a.Cout <= 0 || a.H <= 0 || a.W <= 0 || a.OH <= 0 || a.OW <= 0) return ICM_ERR_ARG;
  if (a.KH <= 0 || a.KW <= 0 || a.KH * a.KW > ICM_MAX_TAPS) return ICM_ERR_UNSUPPORTED;
  if (a.stride != 1 && a.stride != 2) return ICM_ERR_UNSUPPORTED;
  if (!a.transposed) {
    if (a.OH != (a.H + 2 * a.pad - a.KH) / a.stride + 1 || a.OW != (a.W + 2 * a.pad - a.KW) / a.stride + 1)
      return ICM_ERR_ARG;
  }
  if ((a.epi == ICM_EPI_RES || a.epi == ICM_EPI_RES_GELU || a.epi == ICM_EPI_RES_MUL_DGELU) && !a.res) return ICM_ERR_ARG;
  if ((a.epi == ICM_EPI_GDN || a.epi == ICM_EPI_IGDN || a.epi == ICM_EPI_MUL_DGELU || a.epi == ICM_EPI_LRP ||
       a.epi == ICM_EPI_AXPY2 || a.epi == ICM_EPI_RES_MUL_DGELU) && !a.aux)
    return ICM_ERR_ARG;
  if (a.epi == ICM_EPI_AXPY2 && !a.aux2) return ICM_ERR_ARG;
  if (((long long)a.N * a.x_bs + 8LL * a.H * a.W) * 4 >= (1LL << 31)) return ICM_ERR_UNSUPPORTED;   // PlaneMap byte offsets are int32
  if (a.pixel_shuffle != 0 && a.pixel_shuffle != 2) return ICM_ERR_UNSUPPORTED;
  if (a.pixel_shuffle == 2 && (a.Cout % 4 != 0 || a.transposed)) return ICM_ERR_ARG;
  if (a.x_seg_len < 0 || (a.x_seg_len > 0 && (a.x_seg_gap < 0 || a.x_seg_len >= 65536))) return ICM_ERR_ARG;
  return ICM_OK;
}

static int conv_run_grouped(const icm_conv_args* arr, int ngroups, hipStream_t stream) {
  if (!arr || ngroups < 1 || ngroups > ICM_MAX_GROUPS) return ICM_ERR_ARG;
  for (int i = 0; i < ngroups; ++i) {
    int rc = validate(arr[i]);
    if (rc) return rc;
  }
  const icm_conv_args& a = arr[0];
  // one launch, one geometry: every member must agree with arr[0] in everything but its pointers (the kernel takes
  // shapes, strides, epilogue kind and flags from arr[0]; a mismatching member would read / write out of bounds)
  for (int i = 1; i < ngroups; ++i) {
    const icm_conv_args& b = arr[i];
    if (b.N != a.N || b.Cin != a.Cin || b.H != a.H || b.W != a.W || b.Cout != a.Cout || b.OH != a.OH || b.OW != a.OW ||
        b.KH != a.KH || b.KW != a.KW || b.stride != a.stride || b.pad != a.pad || b.transposed != a.transposed ||
        b.pro_act != a.pro_act || b.epi != a.epi || b.accum != a.accum || b.pixel_shuffle != a.pixel_shuffle ||
        b.x_bs != a.x_bs || b.y_bs != a.y_bs || b.x_seg_len != a.x_seg_len || b.x_seg_gap != a.x_seg_gap || b.algo != a.algo ||
        (b.res && b.res_bs != a.res_bs) || (b.aux && b.aux_bs != a.aux_bs) || (b.aux2 && b.aux2_bs != a.aux2_bs) ||
        (b.y2 && b.y2_bs != a.y2_bs) || (b.res != nullptr) != (a.res != nullptr) || (b.aux != nullptr) != (a.aux != nullptr) ||
        (b.aux2 != nullptr) != (a.aux2 != nullptr) || (b.y2 != nullptr) != (a.y2 != nullptr) ||
        (b.bias != nullptr) != (a.bias != nullptr) || (b.xv != nullptr) != (a.xv != nullptr))
      return ICM_ERR_ARG;
  }
  if (a.algo == ICM_ALGO_WINOGRAD) return run_conv_wino(arr, ngroups, stream);
  if (a.algo != ICM_ALGO_DIRECT) return ICM_ERR_ARG;
  std::vector<ConvClass> classes = build_classes(a.KH, a.KW, a.stride, a.pad, a.transposed);
  const int ncot = cdiv(a.Cout, 32), nchunks = cdiv(a.Cin, 8);
  long long off = 0;
  for (const ConvClass& cls : classes) {
    int rc = run_class(arr, ngroups, cls, off, a.transposed ? 1 : a.stride, a.transposed ? a.stride : 1, stream);
    if (rc) return rc;
    off += (long long)nchunks * (long long)cls.taps.size() * ncot * 256;
  }
  return ICM_OK;
}

}  // namespace icm

extern "C" {

int icm_conv_run(const icm_conv_args* a, void* stream) {
  return icm::conv_run_grouped(a, 1, (hipStream_t)stream);
}
int icm_conv_run_grouped(const icm_conv_args* a, int ngroups, void* stream) {
  return icm::conv_run_grouped(a, ngroups, (hipStream_t)stream);
}
int icm_conv2d_fwd(const icm_conv_args* a, void* stream) {
  if (!a || a->transposed) return ICM_ERR_ARG;
  return icm_conv_run(a, stream);
}
int icm_convT2d_dgrad(const icm_conv_args* a, void* stream) {
  if (!a || a->transposed) return ICM_ERR_ARG;
  return icm_conv_run(a, stream);
}
int icm_conv2d_dgrad(const icm_conv_args* a, void* stream) {
  if (!a || !a->transposed) return ICM_ERR_ARG;
  return icm_conv_run(a, stream);
}
int icm_convT2d_fwd(const icm_conv_args* a, void* stream) {
  if (!a || !a->transposed) return ICM_ERR_ARG;
  return icm_conv_run(a, stream);
}

int icm_conv_winograd_ok(const icm_conv_args* a) { return (a && icm::wino_supported(*a)) ? 1 : 0; }
int64_t icm_wino_transform_floats(const icm_conv_args* a) {
  return (a && icm::wino_supported(*a) && a->N > 0 && a->Cin > 0) ? (int64_t)icm::wino_transform_floats(*a) : -1;
}
int icm_wino_transform(const icm_conv_args* arr, int ngroups, void* stream) {
  if (!arr || ngroups < 1 || ngroups > ICM_MAX_GROUPS) return ICM_ERR_ARG;
  for (int i = 1; i < ngroups; ++i)
    if (arr[i].N != arr[0].N || arr[i].Cin != arr[0].Cin || arr[i].H != arr[0].H || arr[i].W != arr[0].W ||
        arr[i].x_bs != arr[0].x_bs || arr[i].pro_act != arr[0].pro_act || arr[i].x_seg_len != arr[0].x_seg_len ||
        arr[i].x_seg_gap != arr[0].x_seg_gap)
      return ICM_ERR_ARG;
  return icm::run_wino_transform(arr, ngroups, (hipStream_t)stream);
}

void icm_debug_force_conv_cfg(int idx) { icm::g_force_cfg = idx; }
int icm_debug_forced_conv_cfg(void) { return icm::g_force_cfg; }
void icm_debug_force_conv1x1(int mode) { icm::g_force_1x1 = mode; }

int64_t icm_packed_weight_floats(int Cout, int Cin, int KH, int KW) {
  return (int64_t)icm::cdiv(Cin, 8) * KH * KW * icm::cdiv(Cout, 32) * 256;
}

int icm_pack_weights(const float* w, float* wp, int Cout, int Cin, int KH, int KW, int src_out_major,
                     int transposed, int stride, int pad, int nonneg, float bound, float pedestal, void* stream) {
  using namespace icm;
  if (!w || !wp || Cout <= 0 || Cin <= 0 || KH * KW > ICM_MAX_TAPS || (stride != 1 && stride != 2)) return ICM_ERR_ARG;
  std::vector<ConvClass> classes = build_classes(KH, KW, stride, pad, transposed);
  const int ncot = cdiv(Cout, 32), nchunks = cdiv(Cin, 8);
  long long off = 0;
  for (const ConvClass& cls : classes) {
    const int ntaps = (int)cls.taps.size();
    if (ntaps == 0) continue;
    PackDesc d{};
    d.w = w;
    d.wp = wp + off;
    d.Co = Cout; d.Ci = Cin; d.KHW = KH * KW; d.src_out_major = src_out_major;
    d.ntaps = ntaps; d.ncot = ncot; d.nchunks = nchunks; d.nonneg = nonneg;
    d.bound = bound; d.pedestal = pedestal;
    d.src_ld = src_out_major ? Cin : Cout; d.src_off = 0; d.dst_ncot = ncot; d.dst_cot_off = 0; d.wino = 0;
    for (int t = 0; t < ICM_MAX_TAPS; ++t) d.tapidx[t] = 0;
    for (int t = 0; t < ntaps; ++t) d.tapidx[t] = (short)cls.taps[t].kidx;
    const long long total = (long long)nchunks * ntaps * ncot * 256;
    const int blocks = std::min(nchunks * ncot, 2048);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d);
    ICM_CHECK_LAUNCH();
    off += total;
  }
  return ICM_OK;
}


int icm_pack_weights_batch(const icm_pack_job* jobs, int n, void* stream) {
  using namespace icm;
  if (!jobs || n < 1) return ICM_ERR_ARG;
  PackMulti m{};
  int nb = 0;
  auto flush = [&]() -> int {
    if (nb == 0) return ICM_OK;
    for (int i = nb; i < ICM_PACK_NB; ++i) m.g[i] = m.g[0];
    hipLaunchKernelGGL(pack_weights_multi_kernel, dim3(128, nb), dim3(256), 0, (hipStream_t)stream, m);
    ICM_CHECK_LAUNCH();
    nb = 0;
    return ICM_OK;
  };
  for (int k = 0; k < n; ++k) {
    const icm_pack_job& J = jobs[k];
    if (!J.w || !J.wp || J.Cout <= 0 || J.Cin <= 0 || J.KH * J.KW > ICM_MAX_TAPS || (J.stride != 1 && J.stride != 2))
      return ICM_ERR_ARG;
    std::vector<ConvClass> classes = build_classes(J.KH, J.KW, J.stride, J.pad, J.transposed);
    if (J.wino) {   // Winograd-domain weights of a 3x3 stride-1 kernel: one "class" of 16 transform points
      if (J.wino < 0 || J.wino > 2 || J.KH != 3 || J.KW != 3 || J.stride != 1 || J.pad != 1 || J.nonneg) return ICM_ERR_ARG;
      ConvClass wc;
      wc.cy = wc.cx = wc.iy0 = wc.ix0 = 0;
      wc.ey = wc.ex = 1;
      for (int t = 0; t < 16; ++t) wc.taps.push_back({0, 0, 0});
      classes.assign(1, wc);
    }
    const int ncot = cdiv(J.Cout, 32), nchunks = cdiv(J.Cin, 8);
    const int inner = J.src_out_major ? J.Cin : J.Cout;
    const int src_ld = J.src_ld > 0 ? J.src_ld : inner;
    const int dst_ncot = J.dst_ncot > 0 ? J.dst_ncot : ncot;
    if (J.src_off < 0 || J.src_off + inner > src_ld || J.dst_cot_off < 0 || J.dst_cot_off + ncot > dst_ncot) return ICM_ERR_ARG;
    long long off = 0;
    for (const ConvClass& cls : classes) {
      const int ntaps = (int)cls.taps.size();
      if (ntaps == 0) continue;
      PackDesc& d = m.g[nb];
      d.w = J.w;
      d.wp = J.wp + off;
      d.Co = J.Cout; d.Ci = J.Cin; d.KHW = J.KH * J.KW; d.src_out_major = J.src_out_major;
      d.ntaps = ntaps; d.ncot = ncot; d.nchunks = nchunks; d.nonneg = J.nonneg;
      d.bound = J.bound; d.pedestal = J.pedestal;
      d.src_ld = src_ld; d.src_off = J.src_off; d.dst_ncot = dst_ncot; d.dst_cot_off = J.dst_cot_off; d.wino = J.wino;
      for (int t = 0; t < ICM_MAX_TAPS; ++t) d.tapidx[t] = 0;
      for (int t = 0; t < ntaps; ++t) d.tapidx[t] = (short)cls.taps[t].kidx;
      off += (long long)nchunks * ntaps * dst_ncot * 256;   // class stride of the (possibly concatenated) destination
      if (++nb == ICM_PACK_NB) {
        int rc = flush();
        if (rc) return rc;
      }
    }
  }
  return flush();
}

}  // extern "C"
