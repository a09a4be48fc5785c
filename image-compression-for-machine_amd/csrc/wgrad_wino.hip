// Weight gradients of 3x3 stride-1 pad-1 convolutions as Winograd F(2x2, 3x3) on f32 MFMA for gfx950.
//
//   Y = A^T [ U (.) V ] A,  U = G g G^T,  V = B^T d B      =>      dU = (A dY A^T) (.) V,      dg = G^T dU G
//
// so the weight gradient is 16 independent GEMMs over the TILE index:  dU[xi][a][b] = sum_tiles dM[xi][a][tile] V[xi][b][tile]
// (a = output channel of the convolution, b = input channel) -- 16 multiply-adds per (a, b) and 2x2 output tile
// instead of the 36 of the direct form (conv_wgrad.hip), i.e. 4/9 of the matrix-pipe work; the 4x4 -> 3x3 transform
// G^T dU G is linear and is applied AFTER the pixel-split partial sums have been added (wgrad_reduce_kernel,
// conv_wgrad.hip), so this kernel only writes dU slabs [split][xi][a][b].
//
// Workgroup = 8 waves, no dedicated loaders: block = 64 a x 64 b x all 16 transform points; wave w owns xi = 2w, 2w+1
// for the 2 x 2 tiles of the block (8 accumulator tiles).  K runs over chunks of 8 tiles (= 32 output pixels): every
// thread gathers one (a, tile) 2x2 patch of dY and one (b, tile) 4x4 patch of x two chunks ahead, transforms them in
// registers (A dY A^T: 2x2 -> 4x4, B^T d B) and writes the 16 + 16 values to LDS as dM[xi][tile][a] / V[xi][tile][b]:
// the tile index is the MFMA K dimension, so fragments are ds_read_b32 with lanes along the channel (row stride 68:
// conflict-free for the transposing writes and for the reads).  The bias gradient is the sum of the dY patches the
// A-side loaders see anyway.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include "icm_common.h"

namespace icm {

#define WW_MAXG 32
#define WW_ROW 68                      /* floats per (xi, tile) row: 64 channels + 4 (bank spreading) */
#define WW_SIDE (16 * 8 * WW_ROW)      /* one operand of one chunk */
#define WW_BUF (2 * WW_SIDE)           /* dM + V of one chunk */

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

template <int I, int N, class F>
__device__ __forceinline__ void ww_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    ww_static_for<I + 1, N>(f);
  }
}

struct WwPtrs {
  const float* gs;   // dY [N][Ca][H][W]
  const float* gb;   // x  [N][Cb][H][W]
  float* ws;         // [nsplit][16][Ca][Cb]
  float* dbias_ws;   // [nsplit][Ca] or null
};
struct WwDesc {
  WwPtrs g[WW_MAXG];
  long long gs_bs, gb_bs;
  int N, Ca, Cb, H, W, act_s, act_b;
  int TW, TH;                 // tiles per image row / column
  FastDiv dTW, dTH;
  int ntiles, nchunks, nsplit, natile, nbtile;
};

template <bool VEC>
__global__ __launch_bounds__(512, 2) void wgrad_wino_kernel(const WwDesc d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const WwPtrs G = d.g[blockIdx.y];
  int bid = blockIdx.x;
  const int at = bid % d.natile; bid /= d.natile;
  const int bt = bid % d.nbtile;
  const int split = bid / d.nbtile;
  const int a0 = at * 64, b0 = bt * 64;
  // this workgroup's chunks: split, split + nsplit, ...
  const int niter = (d.nchunks - split + d.nsplit - 1) / d.nsplit;
  const int HW = d.H * d.W;

  // ---- loader role of this thread: channel (a and b alike) wave * 8 + (lane >> 3), tile lane & 7 of the chunk
  const int ch_l = wave * 8 + (lane >> 3), tl = lane & 7;
  const bool aok = a0 + ch_l < d.Ca, bok = b0 + ch_l < d.Cb;
  const int offa = (a0 + ch_l) * HW, offb = (b0 + ch_l) * HW;   // element offsets from the (uniform) tensor bases: < 2^29 (host-checked)
  float bsum = 0.0f;

  // ---- gather state.  ONE register set (4 + 16 floats) holds the raw dY / x values of one chunk, plus a validity mask.
  // The gathers are straight-line code: an invalid piece (image border, channel / tile past the end) reads element 0 of
  // its tensor instead and is zeroed by the mask at transform time -- no exec-masked regions, so the loads go straight
  // to their destination registers and the only vmcnt wait is at their first use, a whole multiply later.
  // VEC (even W >= 4, host-checked): a tile's two dY rows are 8-byte pairs and its four x rows 16-byte runs, one load
  // instruction each: 6 gather instructions per thread and chunk instead of 20 (the address path of the CU, not the
  // matrix pipe, bounded the dword version).  At the left / right image border the run is shifted by one column to
  // stay inside the row and the patch is picked out of it at transform time (mask bits 8 / 9).
  f32x2 ya[2];
  f32x4 xb[4];
  unsigned vmask = 0;           // VEC: bit r = dY row r, bit 2 + r = x row r, 8 = left border, 9 = right border
  //                               else: bit 2 r + c = dY element, bit 4 + 4 r + c = x element
  unsigned oa_cur = 0, ob_cur = 0;
  auto prep = [&](int it) {     // addresses + mask of chunk `it` (past the end: mask 0) for this thread's (channel, tile)
    const int tau = (split + it * d.nsplit) * 8 + tl;
    const uint32_t q = fdiv((uint32_t)tau, d.dTW);
    const int tx = tau - (int)q * d.TW;
    const uint32_t n = fdiv(q, d.dTH);
    const int ty = (int)(q - n * (uint32_t)d.TH);
    const bool tok = tau < d.ntiles;
    const int oy = ty * 2, ox = tx * 2;
    oa_cur = (unsigned)(offa + (int)n * (int)d.gs_bs + oy * d.W + ox);
    unsigned m = 0;
    if constexpr (VEC) {
#pragma unroll
      for (int r = 0; r < 2; ++r) m |= (unsigned)(tok && aok && oy + r < d.H) << r;
#pragma unroll
      for (int r = 0; r < 4; ++r) m |= (unsigned)(tok && bok && (unsigned)(oy - 1 + r) < (unsigned)d.H) << (2 + r);
      const bool left = ox == 0, right = ox == d.W - 2;
      m |= (unsigned)left << 8 | (unsigned)right << 9;
      ob_cur = (unsigned)(offb + (int)n * (int)d.gb_bs + (oy - 1) * d.W + (ox - 1) + (left ? 1 : (right ? -1 : 0)));
    } else {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) m |= (unsigned)(tok && aok && oy + r < d.H && ox + c < d.W) << (2 * r + c);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          m |= (unsigned)(tok && bok && (unsigned)(oy - 1 + r) < (unsigned)d.H && (unsigned)(ox - 1 + c) < (unsigned)d.W) << (4 + 4 * r + c);
      ob_cur = (unsigned)(offb + (int)n * (int)d.gb_bs + (oy - 1) * d.W + (ox - 1));
    }
    vmask = m;
  };
  auto piece = [&](auto kc) {   // k = 0, 1: dY rows; 2..5: x rows
    constexpr int k = decltype(kc)::value;
    if constexpr (k < 2) {
      constexpr int r = k;
      if constexpr (VEC) {
        const unsigned off = (vmask >> r & 1u) ? oa_cur + r * d.W : 0u;
        const f32x2u v = *reinterpret_cast<const f32x2u*>(G.gs + off);
        ya[r] = f32x2{v[0], v[1]};
      } else {
#pragma unroll
        for (int c = 0; c < 2; ++c) ya[r][c] = G.gs[(vmask >> (2 * r + c) & 1u) ? oa_cur + r * d.W + c : 0u];
      }
    } else {
      constexpr int r = k - 2;
      if constexpr (VEC) {
        const unsigned off = (vmask >> (2 + r) & 1u) ? ob_cur + r * d.W : 0u;
        const f32x4u v = *reinterpret_cast<const f32x4u*>(G.gb + off);
        xb[r] = f32x4{v[0], v[1], v[2], v[3]};
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) xb[r][c] = G.gb[(vmask >> (4 + 4 * r + c) & 1u) ? ob_cur + r * d.W + c : 0u];
      }
    }
  };
  // ---- transform, in two steps: rowpass() consumes the register set (so the next gathers can be issued right away) and
  // leaves A dY (8 values) and B^T d (16 values); store_part<g>() finishes one row of either (second factor + 4 LDS
  // stores) -- eight parts, one per MFMA group of the multiply below.
  float tY[4][2], uX[16];
  auto rowpass = [&]() {
    {   // dM = A dY A^T,  A = [[1,0],[1,1],[1,-1],[0,-1]]
      float y[4];   // (activation-free operands only: host-checked)
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) y[2 * r + c] = (vmask >> (VEC ? r : 2 * r + c) & 1u) ? ya[r][c] : 0.0f;
      bsum += (y[0] + y[1]) + (y[2] + y[3]);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        tY[0][c] = y[0 * 2 + c];
        tY[1][c] = y[0 * 2 + c] + y[1 * 2 + c];
        tY[2][c] = y[0 * 2 + c] - y[1 * 2 + c];
        tY[3][c] = -y[1 * 2 + c];
      }
    }
    {   // V = B^T d B
      float dd[16];
      if constexpr (VEC) {
        const bool left = vmask >> 8 & 1u, right = vmask >> 9 & 1u;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool rok = vmask >> (2 + r) & 1u;
          const f32x4 v = xb[r];
          const float e0 = left ? 0.0f : (right ? v[1] : v[0]);
          const float e1 = left ? v[0] : (right ? v[2] : v[1]);
          const float e2 = left ? v[1] : (right ? v[3] : v[2]);
          const float e3 = left ? v[2] : (right ? 0.0f : v[3]);
          dd[r * 4 + 0] = rok ? e0 : 0.0f;
          dd[r * 4 + 1] = rok ? e1 : 0.0f;
          dd[r * 4 + 2] = rok ? e2 : 0.0f;
          dd[r * 4 + 3] = rok ? e3 : 0.0f;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) dd[r * 4 + c] = (vmask >> (4 + 4 * r + c) & 1u) ? xb[r][c] : 0.0f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uX[0 * 4 + j] = dd[0 * 4 + j] - dd[2 * 4 + j];
        uX[1 * 4 + j] = dd[1 * 4 + j] + dd[2 * 4 + j];
        uX[2 * 4 + j] = dd[2 * 4 + j] - dd[1 * 4 + j];
        uX[3 * 4 + j] = dd[1 * 4 + j] - dd[3 * 4 + j];
      }
    }
  };
  auto store_part = [&](auto gc, int buf) {
    constexpr int g = decltype(gc)::value;
    float* da = smem + buf * WW_BUF + tl * WW_ROW + ch_l;
    if constexpr (g < 4) {
      constexpr int i = g;
      da[(i * 4 + 0) * 8 * WW_ROW] = tY[i][0];
      da[(i * 4 + 1) * 8 * WW_ROW] = tY[i][0] + tY[i][1];
      da[(i * 4 + 2) * 8 * WW_ROW] = tY[i][0] - tY[i][1];
      da[(i * 4 + 3) * 8 * WW_ROW] = -tY[i][1];
    } else {
      constexpr int i = g - 4;
      float* db = da + WW_SIDE;
      db[(i * 4 + 0) * 8 * WW_ROW] = uX[i * 4 + 0] - uX[i * 4 + 2];
      db[(i * 4 + 1) * 8 * WW_ROW] = uX[i * 4 + 1] + uX[i * 4 + 2];
      db[(i * 4 + 2) * 8 * WW_ROW] = uX[i * 4 + 2] - uX[i * 4 + 1];
      db[(i * 4 + 3) * 8 * WW_ROW] = uX[i * 4 + 1] - uX[i * 4 + 3];
    }
  };

  // ---- MFMA role: transform points xi = 2 wave, 2 wave + 1; 2 x 2 tiles of the 64 x 64 block
  const int h = lane >> 5, l31 = lane & 31;
  f32x16 acc[2][2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[x][i][j][e] = 0.0f;
  const int foff = (2 * wave) * 8 * WW_ROW + h * WW_ROW + l31;   // fragment base: xi = 2 wave, tile = h (+ 2 j), channel l31
  // One chunk: 8 groups of 4 MFMAs on buffer `rd`.  Riding on them, in program order so that the address path, the LDS
  // and the VALU work in the shadow of the matrix pipe instead of in a phase of their own: the six gather instructions of
  // chunk it + 2 (groups 0-2, as early as possible: their data is needed at the top of the next iteration), the eight
  // transform parts of chunk it + 1 (stores into buffer `wr`), and the fragment reads of the NEXT group.
  auto multiply = [&](int rd, int wr) {
    const float* pa = smem + rd * WW_BUF + foff;
    const float* pb = pa + WW_SIDE;
    float a0v = pa[0], a1v = pa[32], b0v = pb[0], b1v = pb[32];
    ww_static_for<0, 8>([&](auto gc) {
      constexpr int g = decltype(gc)::value, x = g >> 2;
      float na0 = 0.0f, na1 = 0.0f, nb0 = 0.0f, nb1 = 0.0f;
      if constexpr (g < 7) {
        constexpr int o = (((g + 1) >> 2) * 8 + 2 * ((g + 1) & 3)) * WW_ROW;
        na0 = pa[o]; na1 = pa[o + 32]; nb0 = pb[o]; nb1 = pb[o + 32];
      }
      if constexpr (g < 3) {
        piece(std::integral_constant<int, 2 * g>{});
        piece(std::integral_constant<int, 2 * g + 1>{});
      }
      store_part(gc, wr);
      __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise sinks the gathers and stores below the last MFMA)
      acc[x][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0v, b0v, acc[x][0][0], 0, 0, 0);
      acc[x][0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0v, b1v, acc[x][0][1], 0, 0, 0);
      acc[x][1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v, b0v, acc[x][1][0], 0, 0, 0);
      acc[x][1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v, b1v, acc[x][1][1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      a0v = na0; a1v = na1; b0v = nb0; b1v = nb1;
    });
  };

  // Chunks past the end have an all-zero mask (they read element 0 and store zeros into the buffer nobody reads any
  // more), so the loop body has no conditional code.  Measured on the 320 -> 224 x 30 launch
  // (profiles/r03_tune_wgrad_wino.txt): multiplies alone 1.8 us per chunk and CU, gather issue alone 1.1 us, transform
  // 0.4 us -- as separate phases between barriers the three simply added up (3.3 us).
  prep(0);
  ww_static_for<0, 6>(piece);
  rowpass();
  ww_static_for<0, 8>([&](auto gc) { store_part(gc, 0); });
  prep(1);
  ww_static_for<0, 6>(piece);
  __syncthreads();
  for (int it = 0; it < niter; ++it) {
    rowpass();          // chunk it + 1 (gathered during the previous multiply)
    prep(it + 2);
    multiply(it & 1, (it + 1) & 1);
    __syncthreads();
  }

  // ---- bias gradient: sum over the 8 tile lanes of a channel (fixed order), one writer per channel
  if (G.dbias_ws != nullptr && bt == 0) {
    float s = bsum;
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (tl == 0 && aok) G.dbias_ws[(long long)split * d.Ca + a0 + ch_l] = s;
  }
  // ---- dU slabs [split][xi][a][b]
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    const int xi = 2 * wave + x;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int b = b0 + j * 32 + l31;
        float* row0 = G.ws + (((long long)split * 16 + xi) * d.Ca + a0 + i * 32 + 4 * h) * d.Cb + b;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int ar = (e & 3) + 8 * (e >> 2);
          if (a0 + i * 32 + 4 * h + ar < d.Ca && b < d.Cb) row0[(long long)ar * d.Cb] = acc[x][i][j][e];
        }
        __builtin_amdgcn_sched_barrier(0);   // one tile's stores at a time: bounds the live address registers
      }
  }
}

// pixel splits of the Winograd weight gradient: fill the chip (one 512-thread workgroup per CU) with whole rounds
int wgrad_wino_plan(const icm_wgrad_args& a, int nproblems, int* nsplit_out, int* nchunks_out) {
  if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.OH != a.H || a.OW != a.W) return ICM_ERR_UNSUPPORTED;
  if (a.act_s != ICM_ACT_NONE || a.act_b != ICM_ACT_NONE) return ICM_ERR_UNSUPPORTED;   // materialised operands only
  const int TW = cdiv(a.W, 2), TH = cdiv(a.H, 2);
  const long long ntiles = (long long)a.N * TW * TH;
  if (ntiles >= 65536) return ICM_ERR_UNSUPPORTED;   // tile index -> (image, row, column) by the 16-bit fast division
  // element offsets inside the kernel are 32-bit
  if ((long long)a.N * a.gs_bs + (long long)(a.Ca + 64) * a.H * a.W >= (1LL << 30) ||
      (long long)a.N * a.gb_bs + (long long)(a.Cb + 64) * a.H * a.W >= (1LL << 30))
    return ICM_ERR_UNSUPPORTED;
  const int nchunks = (int)((ntiles + 7) / 8);
  const int base = cdiv(a.Ca, 64) * cdiv(a.Cb, 64) * std::max(1, nproblems);
  double best = 1e300;
  int nsplit = 1;
  for (int sp = 1; sp <= std::min(nchunks, 256); ++sp) {
    const double rounds = std::ceil((double)base * sp / 256.0);
    const double per = (double)cdiv(nchunks, sp) + 2.0;   // + prologue / slab store
    const double cost = rounds * per * (1.0 + 0.004 * sp);
    if (cost < best - 1e-9) { best = cost; nsplit = sp; }
  }
  *nsplit_out = nsplit;
  *nchunks_out = nchunks;
  return ICM_OK;
}

int launch_wgrad_wino(const icm_wgrad_args* arr, int n, int nsplit, int nchunks, float* const* ws, float* const* dbias_ws,
                      hipStream_t stream) {
  const icm_wgrad_args& a = arr[0];
  WwDesc d{};
  for (int i = 0; i < WW_MAXG; ++i) {
    const int k = i < n ? i : 0;
    d.g[i].gs = arr[k].gs; d.g[i].gb = arr[k].gb; d.g[i].ws = ws[k]; d.g[i].dbias_ws = dbias_ws[k];
  }
  d.gs_bs = a.gs_bs; d.gb_bs = a.gb_bs;
  d.N = a.N; d.Ca = a.Ca; d.Cb = a.Cb; d.H = a.H; d.W = a.W; d.act_s = a.act_s; d.act_b = a.act_b;
  d.TW = cdiv(a.W, 2); d.TH = cdiv(a.H, 2);
  d.dTW = make_fastdiv((uint32_t)d.TW); d.dTH = make_fastdiv((uint32_t)d.TH);
  d.ntiles = a.N * d.TW * d.TH; d.nchunks = nchunks; d.nsplit = nsplit;
  d.natile = cdiv(a.Ca, 64); d.nbtile = cdiv(a.Cb, 64);
  const long long nblk = (long long)d.natile * d.nbtile * nsplit;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return ICM_ERR_ARG;
  const size_t lds = (size_t)2 * WW_BUF * sizeof(float);
  static const bool novec = [] { const char* e = getenv("ICM_WW_NOVEC"); return e && atoi(e) != 0; }();   // measurement only
  auto fn = (!novec && a.W % 2 == 0 && a.W >= 4) ? wgrad_wino_kernel<true> : wgrad_wino_kernel<false>;
  if (!ensure_max_lds(reinterpret_cast<const void*>(fn))) return ICM_ERR_LAUNCH;
  hipLaunchKernelGGL(fn, dim3((unsigned)nblk, n), dim3(512), lds, stream, d);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}

}  // namespace icm
