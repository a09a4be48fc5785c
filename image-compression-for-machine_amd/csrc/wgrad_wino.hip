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
#include "icm_common.h"

namespace icm {

#define WW_MAXG 32
#define WW_ROW 68                      /* floats per (xi, tile) row: 64 channels + 4 (bank spreading) */
#define WW_SIDE (16 * 8 * WW_ROW)      /* one operand of one chunk */
#define WW_BUF (2 * WW_SIDE)           /* dM + V of one chunk */

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
  auto load = [&](float (&ra)[4], float (&rb)[16], int it) {
    const int tau = (split + it * d.nsplit) * 8 + tl;
    const uint32_t q = fdiv((uint32_t)tau, d.dTW);
    const int tx = tau - (int)q * d.TW;
    const uint32_t n = fdiv(q, d.dTH);
    const int ty = (int)(q - n * (uint32_t)d.TH);
    const bool tok = tau < d.ntiles;
    const int oy = ty * 2, ox = tx * 2;
    const float* pa = G.gs + (unsigned)(offa + (int)n * (int)d.gs_bs + oy * d.W + ox);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int c = 0; c < 2; ++c)
        ra[r * 2 + c] = (tok && aok && oy + r < d.H && ox + c < d.W) ? pa[r * d.W + c] : 0.0f;
    const int ob = offb + (int)n * (int)d.gb_bs + (oy - 1) * d.W + (ox - 1);   // may be "negative" at the image border: masked
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int iy = oy - 1 + r, ix = ox - 1 + c;
        rb[r * 4 + c] = (tok && bok && (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W) ? G.gb[(unsigned)(ob + r * d.W + c)] : 0.0f;
      }
  };
  auto transform_store = [&](const float (&ra)[4], const float (&rb)[16], int buf) {
    float* da = smem + buf * WW_BUF + tl * WW_ROW + ch_l;
    float* db = da + WW_SIDE;
    {   // dM = A dY A^T,  A = [[1,0],[1,1],[1,-1],[0,-1]]
      float y[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = ra[e];   // (activation-free operands only: host-checked)
      bsum += (y[0] + y[1]) + (y[2] + y[3]);
      float t[4][2];   // A dY
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        t[0][c] = y[0 * 2 + c];
        t[1][c] = y[0 * 2 + c] + y[1 * 2 + c];
        t[2][c] = y[0 * 2 + c] - y[1 * 2 + c];
        t[3][c] = -y[1 * 2 + c];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        da[(i * 4 + 0) * 8 * WW_ROW] = t[i][0];
        da[(i * 4 + 1) * 8 * WW_ROW] = t[i][0] + t[i][1];
        da[(i * 4 + 2) * 8 * WW_ROW] = t[i][0] - t[i][1];
        da[(i * 4 + 3) * 8 * WW_ROW] = -t[i][1];
      }
    }
    {   // V = B^T d B
      float dd[16], u[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) dd[e] = rb[e];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        u[0 * 4 + j] = dd[0 * 4 + j] - dd[2 * 4 + j];
        u[1 * 4 + j] = dd[1 * 4 + j] + dd[2 * 4 + j];
        u[2 * 4 + j] = dd[2 * 4 + j] - dd[1 * 4 + j];
        u[3 * 4 + j] = dd[1 * 4 + j] - dd[3 * 4 + j];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        db[(i * 4 + 0) * 8 * WW_ROW] = u[i * 4 + 0] - u[i * 4 + 2];
        db[(i * 4 + 1) * 8 * WW_ROW] = u[i * 4 + 1] + u[i * 4 + 2];
        db[(i * 4 + 2) * 8 * WW_ROW] = u[i * 4 + 2] - u[i * 4 + 1];
        db[(i * 4 + 3) * 8 * WW_ROW] = u[i * 4 + 1] - u[i * 4 + 3];
      }
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
  auto multiply = [&](int buf) {
    const float* pa = smem + buf * WW_BUF + foff;
    const float* pb = pa + WW_SIDE;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int o = (x * 8 + 2 * j) * WW_ROW;
        const float a0v = pa[o], a1v = pa[o + 32], b0v = pb[o], b1v = pb[o + 32];
        acc[x][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0v, b0v, acc[x][0][0], 0, 0, 0);
        acc[x][0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0v, b1v, acc[x][0][1], 0, 0, 0);
        acc[x][1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v, b0v, acc[x][1][0], 0, 0, 0);
        acc[x][1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1v, b1v, acc[x][1][1], 0, 0, 0);
      }
  };

  // one register set: the patches of chunk it + 2 are requested right after chunk it + 1 has been transformed and are
  // consumed after the next multiply -- a whole chunk (~4 000 cycles) of latency budget without a second set of 20
  // registers next to the 128 accumulator registers
  float ra[4], rb[16];
  load(ra, rb, 0);
  transform_store(ra, rb, 0);
  if (niter > 1) load(ra, rb, 1);
  __syncthreads();
  for (int it = 0; it < niter; ++it) {
    multiply(it & 1);
    if (it + 1 < niter) transform_store(ra, rb, (it + 1) & 1);
    if (it + 2 < niter) load(ra, rb, it + 2);
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
  if (!ensure_max_lds(reinterpret_cast<const void*>(wgrad_wino_kernel))) return ICM_ERR_LAUNCH;
  hipLaunchKernelGGL(wgrad_wino_kernel, dim3((unsigned)nblk, n), dim3(512), lds, stream, d);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}

}  // namespace icm
