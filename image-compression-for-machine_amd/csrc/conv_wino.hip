// 3x3 stride-1 pad-1 convolutions (forward and input gradient) as Winograd F(2x2, 3x3) on f32 MFMA for gfx950.
//
// Half of the model's contraction FLOP are 3x3 stride-1 convolutions (the 150 slice-chain layers, the ResidualUnit
// 3x3s of the four attention gates, the hyper synthesis; cnn.py:54-127, layers.py:52-72).  For a 2x2 output tile
//      Y = A^T [ (G g G^T) (.) (B^T d B) ] A        d: 4x4 input patch, g: 3x3 kernel,
// the 36 multiply-adds per (ci, co) of the direct form become 16: the contraction over input channels runs in the
// transform domain as 16 independent GEMMs  M[xi][co][tile] = sum_ci U[xi][co][ci] V[xi][ci][tile]  -- 4/9 of the
// matrix-pipe work.  Everything stays f32; the transform matrices hold 0, +-1, +-1/2 only (measured error against the
// direct kernel: a few 1e-7 of the tensor maximum, the same order as a change of summation order).
//
// Kernel structure (MI355X-first; same 4 MFMA + 4 loader wave split as conv_igemm.hip, one workgroup per CU):
//   * workgroup = (TCO x 32 output channels) x (32 tiles = 128 output pixels), all 16 transform points;
//   * loader waves: lane = (input channel of the 8-channel chunk, tile); it gathers its 4x4 patch straight from global
//     memory (zero padding by mask, optional virtual activation), applies B^T d B in registers (32 add/sub) and writes
//     the 16 transformed values to LDS as V[xi][ci][tile] -- exactly the B-fragment order of v_mfma_f32_32x32x2_f32
//     (lane = (tile, k parity)), so MFMA waves read it with conflict-free, linear ds_read_b32.  Two chunks per
//     barrier, double buffered, loads issued one full step (~4 000 cycles) before their transform;
//   * MFMA wave r owns transform row r (xi = 4r .. 4r+3): 4 x TCO accumulator tiles; A fragments are the Winograd-domain
//     weights U = G g G^T, pre-packed per step in fragment order (icm_pack_weights with wino != 0) and streamed from L2
//     three units ahead;
//   * epilogue: the column half of A^T M A happens in registers (each wave holds a whole row), the row half through
//     LDS; every wave then finishes a quarter of the (channel, tile) elements: 2x2 pixels per lane with the fused
//     neighbours of conv_common.h (bias, residual, GELU materialisation, GELU', LRP tanh, accumulation).
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include "conv_common.h"

namespace icm {

struct WinoDesc {
  ConvPtrs g[ICM_MAX_GROUPS];
  long long x_bs, y_bs, res_bs, aux_bs, y2_bs;
  int N, Cin, Cout, H, W;
  int lgTX, lgTY, lgTI, tiles_x, tiles_y, tiles_n;   // a block = (1 << lgTX) x (1 << lgTY) tiles of (1 << lgTI) images = 32 tiles
  int ncot, nchunks8, ncb, nsteps;                    // nsteps = ceil(nchunks8 / 2): two 8-channel chunks per barrier
  int epi, accum, act;
  int seg_len, seg_gap;
  FastDiv dseg;
  int dbg;            // measurement only (ICM_WINO_DEBUG): 1 = no patch loads, 2 = no weight loads in the loop, 4 = no stores,
                      // 8 = no input transform / LDS stores, 16 = no B-fragment LDS reads
  int vpre;           // the operand pointer holds PRE-TRANSFORMED input (wino_input_transform_kernel): [px block][chunk][xi][ci][tile]
  int px_fast, npx;   // workgroup order: pixel blocks fastest (weights stationary per XCD) when the weights outweigh the activations
};

#define WINO_STEP_FLOATS (2 * 16 * 8 * 32)   /* V of one step: [chunk 2][xi 16][ci 8][tile 32] */

template <int EPI>
__device__ __forceinline__ void wino_finish(const WinoDesc& d, const ConvPtrs& P, float v, int co, long long off, bool ok) {
  // one output element: bias / fused neighbour / accumulate / materialise / store (conv_common.h semantics)
  if (!ok) return;
  if (P.bias) v += P.bias[co];
  if constexpr (EPI == ICM_EPI_RES) v += P.res[off];
  if constexpr (EPI == ICM_EPI_RES_GELU) v += gelu_f(P.res[off]);
  if constexpr (EPI == ICM_EPI_MUL_DGELU) v *= dgelu_f(P.aux[off]);
  if constexpr (EPI == ICM_EPI_RES_MUL_DGELU) v = (v + P.res[off]) * dgelu_f(P.aux[off]);
  if constexpr (EPI == ICM_EPI_LRP) {
    const float t = tanhf(v);
    if (P.y2) P.y2[off] = t;
    v = P.aux[off] + 0.5f * t;
  }
  if (d.accum) v += P.y[off];
  if constexpr (EPI == ICM_EPI_NONE || EPI == ICM_EPI_RES || EPI == ICM_EPI_RES_GELU) {
    if (P.y2) {
      const float gv = gelu_f(v);
      if (P.y2 == P.y) v = gv;   // y2 == y: only the activated value is stored (conv_common.h)
      else P.y2[off] = gv;
    }
  }
  P.y[off] = v;
}

template <int TCO>
__global__ __launch_bounds__(512, 2) void conv_wino_kernel(const WinoDesc d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  ConvPtrs P = d.g[blockIdx.y];
  int bid;
  {
    const int nb = gridDim.x, hb = blockIdx.x;
    const int xcd = hb & 7, q = hb >> 3;
    bid = xcd * (nb >> 3) + min(xcd, nb & 7) + q;
  }
  // Which operand should stay in the XCD's L2?  Consecutive logical blocks run together on one XCD.  Small weights
  // (chain layers: 2.5 MB): co-blocks fastest -- the co-blocks of a pixel block share its activations.  Wide first-layer
  // launches (87 MB of Winograd weights against 5 MB of activations): pixel blocks fastest, so each XCD streams only
  // its own 1/8 of the weights instead of all of them once per pixel block.
  const int cb = d.px_fast ? bid / d.npx : bid % d.ncb;
  int pt = d.px_fast ? bid % d.npx : bid / d.ncb;
  const int pblk = pt;
  const int bx = pt % d.tiles_x;
  pt /= d.tiles_x;
  const int by = pt % d.tiles_y;
  const int bn = pt / d.tiles_y;
  const int TXm = (1 << d.lgTX) - 1, TYm = (1 << d.lgTY) - 1;
  const int HW = d.H * d.W;

  if (wave >= 4 && d.vpre) {
    // ------------------------------------------------------------------ loader waves, pre-transformed operand: LDS-DMA only.
    // The transform ran once per input tensor in its own (HBM-bound, microsecond) launch instead of once per co-block
    // here: no VALU / LDS-store work next to the MFMA waves (which cost them 15-25 %: f32 MFMA and VALU contend), and
    // 16-byte coalesced DMA instead of 32 scattered dword gathers per lane and step.
    const int lw = wave - 4;
    const float* vsrc = P.x + (long long)pblk * (d.nsteps * 2) * 4096;
    auto dma = [&](int step, int buf) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int rnd = 0; rnd < 4; ++rnd) {
          const int e = (rnd * 256 + lw * 64) * 4;   // float offset of this wave's 1 KB piece inside the 16 KB chunk
          __builtin_amdgcn_global_load_lds(vsrc + ((long long)(step * 2 + k) * 4096 + e + lane * 4),
                                           smem + buf * WINO_STEP_FLOATS + k * 4096 + e, 16, 0, 0);
        }
    };
    dma(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int s = 0; s < d.nsteps; ++s) {
      if (s + 1 < d.nsteps) dma(s + 1, (s + 1) & 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // DMAs must have landed before the barrier publishes them
      __syncthreads();
    }
  } else if (wave >= 4) {
    // ------------------------------------------------------------------ loader / input-transform waves
    const int q = (wave - 4) * 64 + lane;
    const int ci_l = q >> 5, t = q & 31;
    const int tx = t & TXm, ty = (t >> d.lgTX) & TYm, ti = t >> (d.lgTX + d.lgTY);
    const int n = (bn << d.lgTI) + ti;
    const int oy = (((by << d.lgTY) + ty) << 1), ox = (((bx << d.lgTX) + tx) << 1);
    unsigned mask = 0;
#pragma unroll
    for (int dy = 0; dy < 4; ++dy)
#pragma unroll
      for (int dx = 0; dx < 4; ++dx) {
        const int iy = oy - 1 + dy, ix = ox - 1 + dx;
        if (n < d.N && (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W) mask |= 1u << (dy * 4 + dx);
      }
    const float* xbase = P.x + ((long long)n * d.x_bs + (long long)(oy - 1) * d.W + (ox - 1));
    const int act = d.act;
    auto load = [&](float (&r)[2][16], int step) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int c = (step * 2 + k) * 8 + ci_l;
        const bool cok = c < d.Cin;
        const int cp = d.seg_len ? c + (int)fdiv((uint32_t)c, d.dseg) * d.seg_gap : c;
        const float* pc = xbase + (long long)cp * HW;
#pragma unroll
        for (int dy = 0; dy < 4; ++dy)
#pragma unroll
          for (int dx = 0; dx < 4; ++dx)
            r[k][dy * 4 + dx] = (cok && ((mask >> (dy * 4 + dx)) & 1u) && !(d.dbg & 1)) ? pc[dy * d.W + dx] : 0.0f;
      }
    };
    auto transform_store = [&](const float (&r)[2][16], int buf) {
      if (d.dbg & 8) return;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        float dd[16], u[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) dd[e] = act ? apply_act(r[k][e], act) : r[k][e];
#pragma unroll
        for (int j = 0; j < 4; ++j) {   // rows of B^T d
          u[0 * 4 + j] = dd[0 * 4 + j] - dd[2 * 4 + j];
          u[1 * 4 + j] = dd[1 * 4 + j] + dd[2 * 4 + j];
          u[2 * 4 + j] = dd[2 * 4 + j] - dd[1 * 4 + j];
          u[3 * 4 + j] = dd[1 * 4 + j] - dd[3 * 4 + j];
        }
        float* dst = smem + buf * WINO_STEP_FLOATS + k * (16 * 256) + ci_l * 32 + t;
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // columns: (B^T d) B
          dst[(i * 4 + 0) * 256] = u[i * 4 + 0] - u[i * 4 + 2];
          dst[(i * 4 + 1) * 256] = u[i * 4 + 1] + u[i * 4 + 2];
          dst[(i * 4 + 2) * 256] = u[i * 4 + 2] - u[i * 4 + 1];
          dst[(i * 4 + 3) * 256] = u[i * 4 + 1] - u[i * 4 + 3];
        }
      }
    };
    // three register sets: the gather loads of step s + 3 are issued while step s is multiplied and step s + 1 is
    // transformed -- two full steps (~8 000 cycles) of latency budget (HBM misses under load take 3-5 us; with one
    // step of slack the MFMA waves waited at the barrier: measured -15 % on the wide first-layer launches)
    float ra[2][16], rb[2][16], rc[2][16];
    load(ra, 0);
    if (d.nsteps > 1) load(rb, 1);
    if (d.nsteps > 2) load(rc, 2);
    transform_store(ra, 0);
    __syncthreads();   // step 0 published
    for (int s = 0; s < d.nsteps; s += 3) {
      // step s is being multiplied (buffer s & 1); produce step s + 1 into the other buffer
      if (s + 3 < d.nsteps) load(ra, s + 3);
      if (s + 1 < d.nsteps) transform_store(rb, (s + 1) & 1);
      __syncthreads();
      if (s + 1 >= d.nsteps) break;
      if (s + 4 < d.nsteps) load(rb, s + 4);
      if (s + 2 < d.nsteps) transform_store(rc, (s + 2) & 1);
      __syncthreads();
      if (s + 2 >= d.nsteps) break;
      if (s + 5 < d.nsteps) load(rc, s + 5);
      if (s + 3 < d.nsteps) transform_store(ra, (s + 3) & 1);
      __syncthreads();
    }
  } else {
  // ------------------------------------------------------------------ MFMA waves: wave r = transform row r
  const int r = wave;
  f32x16 acc[4][TCO];
  const int cot0 = cb * TCO;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int a = 0; a < TCO; ++a)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[c][a][e] = 0.0f;
  const char* wbase = reinterpret_cast<const char*>(P.wp);
  unsigned wl[TCO];
#pragma unroll
  for (int a = 0; a < TCO; ++a) wl[a] = (unsigned)(min(cot0 + a, d.ncot - 1) * 64 + lane) * 16u;
  const long long qstride = (long long)d.ncot * 64 * 16;   // bytes between consecutive (chunk, xi) entries
  const int nunits = d.nsteps * 8;                         // (chunk, column) units of this wave, two chunks per step
  const int lastq = d.nchunks8 * 16 - 1;
  auto wptr = [&](int g) -> const char* {                  // packed weights of unit g = chunk * 4 + column
    const int qq = min((g >> 2) * 16 + 4 * r + (g & 3), lastq);
    return wbase + qq * qstride;
  };
  f32x4 aq[4][TCO];
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int a = 0; a < TCO; ++a) aq[u][a] = *reinterpret_cast<const f32x4*>(wptr(u) + wl[a]);
  // B fragment of (chunk k of the step, column c): V[k][4r + c][ci = 2j + h][tile l31] -- linear in the lane id
  const int boff = (4 * r) * 256 + lane;
  __syncthreads();   // step 0 staged
  for (int s = 0; s < d.nsteps; ++s) {
    const float* vb = smem + (s & 1) * WINO_STEP_FLOATS + boff;
    float bv[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[0][j] = vb[j * 64];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int g = s * 8 + u;
      if (u + 1 < 8) {   // next unit's B fragment (same step: same buffer)
        const float* nb = vb + ((u + 1) >> 2) * (16 * 256) + ((u + 1) & 3) * 256;
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[(u + 1) & 1][j] = (d.dbg & 16) ? 1.0f : nb[j * 64];
      }
      const char* sp = wptr(min(g + 3, nunits - 1));
#pragma unroll
      for (int a = 0; a < TCO; ++a) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[u & 3][a] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[u & 3][a][j], bv[u & 1][j], acc[u & 3][a], 0, 0, 0);
        if (!(d.dbg & 2)) aq[(u + 3) & 3][a] = *reinterpret_cast<const f32x4*>(sp + wl[a]);
      }
    }
    __syncthreads();
  }

  // ---- output transform Y = A^T M A.  Column half in registers: Z[.][0] = M0 + M1 + M2, Z[.][1] = M1 - M2 - M3
  float* zb = smem;   // [row r 4][cp 2][TCO][16 regs][64 lanes]   (the staging buffers are free: last barrier passed)
#pragma unroll
  for (int a = 0; a < TCO; ++a)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float m0 = acc[0][a][e], m1 = acc[1][a][e], m2 = acc[2][a][e], m3 = acc[3][a][e];
      zb[(((r * 2 + 0) * TCO + a) * 16 + e) * 64 + lane] = (m0 + m1) + m2;
      zb[(((r * 2 + 1) * TCO + a) * 16 + e) * 64 + lane] = (m1 - m2) - m3;
    }
  }
  __syncthreads();   // Z complete (loader waves join here: the finishing work is spread over all eight waves)
  // row half + epilogue: wave w finishes accumulator rows e = w and w + 8 of every co tile: 2x2 pixels per lane
  const float* zb = smem;
  const int cot0 = cb * TCO;
  const int h = lane >> 5, l31 = lane & 31;
  const int tx = l31 & TXm, ty = (l31 >> d.lgTX) & TYm, ti = l31 >> (d.lgTX + d.lgTY);
  const int n = (bn << d.lgTI) + ti;
  const int oy = (((by << d.lgTY) + ty) << 1), ox = (((bx << d.lgTX) + tx) << 1);
  const bool nok = n < d.N;
  P.y += (long long)n * d.y_bs;
  if (P.y2) P.y2 += (long long)n * d.y2_bs;
  if (P.res) P.res += (long long)n * d.res_bs;
  if (P.aux) P.aux += (long long)n * d.aux_bs;
  auto finish_all = [&](auto epi_tag) {
    constexpr int EPI = decltype(epi_tag)::value;
#pragma unroll
    for (int a = 0; a < TCO; ++a) {
      const int cot = cot0 + a;
      if (cot >= d.ncot) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int e = wave + 8 * i;
        const int co = cot * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        float z[4][2];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
          for (int cp = 0; cp < 2; ++cp) z[rr][cp] = zb[(((rr * 2 + cp) * TCO + a) * 16 + e) * 64 + lane];
        const bool cok = nok && co < d.Cout;
        const long long base = (long long)co * HW + (long long)oy * d.W + ox;
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
          const float y0 = (z[0][cp] + z[1][cp]) + z[2][cp];
          const float y1 = (z[1][cp] - z[2][cp]) - z[3][cp];
          const bool xok = cok && ox + cp < d.W;
          wino_finish<EPI>(d, P, y0, co, base + cp, xok && oy < d.H);
          wino_finish<EPI>(d, P, y1, co, base + d.W + cp, xok && oy + 1 < d.H);
        }
      }
    }
  };
  if (d.dbg & 4) return;
  switch (d.epi) {
    case ICM_EPI_RES: finish_all(std::integral_constant<int, ICM_EPI_RES>{}); break;
    case ICM_EPI_RES_GELU: finish_all(std::integral_constant<int, ICM_EPI_RES_GELU>{}); break;
    case ICM_EPI_MUL_DGELU: finish_all(std::integral_constant<int, ICM_EPI_MUL_DGELU>{}); break;
    case ICM_EPI_RES_MUL_DGELU: finish_all(std::integral_constant<int, ICM_EPI_RES_MUL_DGELU>{}); break;
    case ICM_EPI_LRP: finish_all(std::integral_constant<int, ICM_EPI_LRP>{}); break;
    default: finish_all(std::integral_constant<int, ICM_EPI_NONE>{}); break;
  }
}


// ---- eight MFMA waves on pre-transformed operands (the default path): no loader waves at all.
// With the input transform in its own launch the loaders only issue LDS-DMA, and what bounds the 4 + 4 kernel above is the
// transformed operand itself: 32 KB of V per step feed just 2 co tiles (rocprofv3: matrix pipe 42 % busy, waves parked
// 65 % of their cycles, 5.6x the algorithmic bytes from L2 / fabric).  Here all eight waves multiply -- wave w owns
// transform row w >> 1, columns 2 (w & 1) and 2 (w & 1) + 1 -- so the same 128 accumulator registers hold TCO = 4 co
// tiles: every staged V byte feeds twice the MFMAs, a step is twice as long (half the barriers), two MFMA waves per
// SIMD cover each other's waits, and each wave issues 4 of the step's 32 DMA instructions itself.  The column half of
// the output transform is split between the two waves of a row (partial sums through LDS, fixed order).
template <int TCO>
__global__ __launch_bounds__(512, 2) void conv_wino8_kernel(const WinoDesc d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  ConvPtrs P = d.g[blockIdx.y];
  int bid;
  {
    const int nb = gridDim.x, hb = blockIdx.x;
    const int xcd = hb & 7, q = hb >> 3;
    bid = xcd * (nb >> 3) + min(xcd, nb & 7) + q;
  }
  const int cb = d.px_fast ? bid / d.npx : bid % d.ncb;
  int pt = d.px_fast ? bid % d.npx : bid / d.ncb;
  const int pblk = pt;
  const int bx = pt % d.tiles_x;
  pt /= d.tiles_x;
  const int by = pt % d.tiles_y;
  const int bn = pt / d.tiles_y;
  const int TXm = (1 << d.lgTX) - 1, TYm = (1 << d.lgTY) - 1;
  const int HW = d.H * d.W;
  const int r = wave >> 1, ch = wave & 1;
  const int cot0 = cb * TCO;

  const float* vsrc = P.x + (long long)pblk * (d.nsteps * 2) * 4096;
  auto dma = [&](int step, int buf) {   // this wave's 4 of the 32 one-KB pieces of the step
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int piece = wave + 8 * i;                 // 0..31: chunk piece >> 4, 1 KB piece & 15 inside the 16 KB chunk
      const int e = (piece >> 4) * 4096 + (piece & 15) * 256;
      __builtin_amdgcn_global_load_lds(vsrc + (long long)step * 8192 + e + lane * 4, smem + buf * WINO_STEP_FLOATS + e, 16, 0, 0);
    }
  };
  f32x16 acc[2][TCO];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int a = 0; a < TCO; ++a)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[c][a][e] = 0.0f;
  const char* wbase = reinterpret_cast<const char*>(P.wp);
  unsigned wl[TCO];
#pragma unroll
  for (int a = 0; a < TCO; ++a) wl[a] = (unsigned)(min(cot0 + a, d.ncot - 1) * 64 + lane) * 16u;
  const long long qstride = (long long)d.ncot * 64 * 16;
  const int nunits = d.nsteps * 4;                         // (chunk, column) units of this wave: two chunks x two columns per step
  const int lastq = d.nchunks8 * 16 - 1;
  const int xi0 = 4 * r + 2 * ch;
  auto wptr = [&](int g) -> const char* {                  // unit g = chunk * 2 + column
    const int qq = min((g >> 1) * 16 + xi0 + (g & 1), lastq);
    return wbase + qq * qstride;
  };
  dma(0, 0);
  f32x4 aq[4][TCO];
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int a = 0; a < TCO; ++a) aq[u][a] = *reinterpret_cast<const f32x4*>(wptr(min(u, nunits - 1)) + wl[a]);
  const int boff = xi0 * 256 + lane;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();   // step 0 landed
  for (int s = 0; s < d.nsteps; ++s) {
    if (s + 1 < d.nsteps) dma(s + 1, (s + 1) & 1);   // issued first: older than every weight load of this step
    const float* vb = smem + (s & 1) * WINO_STEP_FLOATS + boff;
    float bv[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[0][j] = vb[j * 64];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int g = s * 4 + u;
      if (u + 1 < 4) {
        const float* nb = vb + ((u + 1) >> 1) * (16 * 256) + ((u + 1) & 1) * 256;
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[(u + 1) & 1][j] = nb[j * 64];
      }
      const char* sp = wptr(min(g + 3, nunits - 1));
#pragma unroll
      for (int a = 0; a < TCO; ++a) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[u & 1][a] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[u][a][j], bv[u & 1][j], acc[u & 1][a], 0, 0, 0);
        aq[(u + 3) & 3][a] = *reinterpret_cast<const f32x4*>(sp + wl[a]);
      }
    }
    // the DMAs of the next step are older than the 3 * TCO weight loads still in flight: wait for them only
    if constexpr (TCO == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (TCO == 3) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS reads of the buffer are complete
    __builtin_amdgcn_s_barrier();
  }

  // ---- output transform.  Column half per wave pair of a row: ch 0 holds M0, M1; ch 1 holds M2, M3
  //      Z[.][0] = (M0 + M1) + M2,   Z[.][1] = M1 + (-M2 - M3):  each wave contributes one partial per Z
  float* zb = smem;   // per pass of two co tiles: [a 2][row 4][ch 2][cp 2][e 16][lane 64] = 128 KB
  const int h = lane >> 5, l31 = lane & 31;
  const int tx = l31 & TXm, ty = (l31 >> d.lgTX) & TYm, ti = l31 >> (d.lgTX + d.lgTY);
  const int n = (bn << d.lgTI) + ti;
  const int oy = (((by << d.lgTY) + ty) << 1), ox = (((bx << d.lgTX) + tx) << 1);
  const bool nok = n < d.N;
  P.y += (long long)n * d.y_bs;
  if (P.y2) P.y2 += (long long)n * d.y2_bs;
  if (P.res) P.res += (long long)n * d.res_bs;
  if (P.aux) P.aux += (long long)n * d.aux_bs;
#pragma unroll
  for (int a0 = 0; a0 < TCO; a0 += 2) {
    __syncthreads();   // staging buffers / the previous pass are free
#pragma unroll
    for (int al = 0; al < 2; ++al) {
      if (a0 + al >= TCO) break;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float m0 = acc[0][a0 + al][e], m1 = acc[1][a0 + al][e];
        const float p0 = ch == 0 ? m0 + m1 : m0;
        const float p1 = ch == 0 ? m1 : -m0 - m1;
        zb[((((al * 4 + r) * 2 + ch) * 2 + 0) * 16 + e) * 64 + lane] = p0;
        zb[((((al * 4 + r) * 2 + ch) * 2 + 1) * 16 + e) * 64 + lane] = p1;
      }
    }
    __syncthreads();
    auto finish_all = [&](auto epi_tag) {
      constexpr int EPI = decltype(epi_tag)::value;
#pragma unroll
      for (int al = 0; al < 2; ++al) {
        const int cot = cot0 + a0 + al;
        if (a0 + al >= TCO || cot >= d.ncot) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int e = wave + 8 * i;
          const int co = cot * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          float z[4][2];
#pragma unroll
          for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int cp = 0; cp < 2; ++cp)
              z[rr][cp] = zb[((((al * 4 + rr) * 2 + 0) * 2 + cp) * 16 + e) * 64 + lane] +
                          zb[((((al * 4 + rr) * 2 + 1) * 2 + cp) * 16 + e) * 64 + lane];
          const bool cok = nok && co < d.Cout;
          const long long base = (long long)co * HW + (long long)oy * d.W + ox;
#pragma unroll
          for (int cp = 0; cp < 2; ++cp) {
            const float y0 = (z[0][cp] + z[1][cp]) + z[2][cp];
            const float y1 = (z[1][cp] - z[2][cp]) - z[3][cp];
            const bool xok = cok && ox + cp < d.W;
            wino_finish<EPI>(d, P, y0, co, base + cp, xok && oy < d.H);
            wino_finish<EPI>(d, P, y1, co, base + d.W + cp, xok && oy + 1 < d.H);
          }
        }
      }
    };
    switch (d.epi) {
      case ICM_EPI_RES: finish_all(std::integral_constant<int, ICM_EPI_RES>{}); break;
      case ICM_EPI_RES_GELU: finish_all(std::integral_constant<int, ICM_EPI_RES_GELU>{}); break;
      case ICM_EPI_MUL_DGELU: finish_all(std::integral_constant<int, ICM_EPI_MUL_DGELU>{}); break;
      case ICM_EPI_RES_MUL_DGELU: finish_all(std::integral_constant<int, ICM_EPI_RES_MUL_DGELU>{}); break;
      case ICM_EPI_LRP: finish_all(std::integral_constant<int, ICM_EPI_LRP>{}); break;
      default: finish_all(std::integral_constant<int, ICM_EPI_NONE>{}); break;
    }
  }
}

bool wino_supported(const icm_conv_args& a);

// ---- input transform as its own launch: V[px block][chunk][xi][ci][tile] = B^T d B of every 4x4 patch (zero padding, the
// operand activation and the blocked channel map applied here).  One workgroup = one (px block, 8-channel chunk).
struct WinoXfDesc {
  const float* x[ICM_MAX_GROUPS];
  float* v[ICM_MAX_GROUPS];
  long long x_bs;
  int N, Cin, H, W, lgTX, lgTY, lgTI, tiles_x, tiles_y, nchunks, act, seg_len, seg_gap;
  FastDiv dseg;
};
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
// VEC (even W >= 4): a patch row is ONE 16-byte load (shifted by a column at the left / right image border so that it
// stays inside the row, the patch picked out of it) instead of four dword gathers; an invalid row reads element 0 of the
// tensor and is zeroed -- straight-line loads, a quarter of the gather instructions (the address path bounds this kernel).
template <bool VEC>
__global__ __launch_bounds__(256) void wino_input_transform_kernel(const WinoXfDesc d) {
  const float* x = d.x[blockIdx.y];
  float* v = d.v[blockIdx.y];
  const int chunk = blockIdx.x % d.nchunks;
  int pt = blockIdx.x / d.nchunks;
  const int pblk = pt;
  const int bx = pt % d.tiles_x;
  pt /= d.tiles_x;
  const int by = pt % d.tiles_y;
  const int bn = pt / d.tiles_y;
  const int q = threadIdx.x, ci_l = q >> 5, t = q & 31;
  const int TXm = (1 << d.lgTX) - 1, TYm = (1 << d.lgTY) - 1;
  const int tx = t & TXm, ty = (t >> d.lgTX) & TYm, ti = t >> (d.lgTX + d.lgTY);
  const int n = (bn << d.lgTI) + ti;
  const int oy = (((by << d.lgTY) + ty) << 1), ox = (((bx << d.lgTX) + tx) << 1);
  const int c = chunk * 8 + ci_l;
  const bool cok = c < d.Cin && n < d.N;
  const int cp = d.seg_len ? c + (int)fdiv((uint32_t)c, d.dseg) * d.seg_gap : c;
  const long long pbase = (long long)n * d.x_bs + (long long)cp * d.H * d.W + (long long)(oy - 1) * d.W + (ox - 1);
  float dd[16], u[16];
  if constexpr (VEC) {
    const bool tvalid = cok && ox < d.W;
    const bool left = ox == 0, right = ox == d.W - 2;
    f32x4u rv[4];
    bool rok[4];
#pragma unroll
    for (int dy = 0; dy < 4; ++dy) {
      rok[dy] = tvalid && (unsigned)(oy - 1 + dy) < (unsigned)d.H;
      const long long off = rok[dy] ? pbase + dy * d.W + (left ? 1 : (right ? -1 : 0)) : 0;
      rv[dy] = *reinterpret_cast<const f32x4u*>(x + off);
    }
#pragma unroll
    for (int dy = 0; dy < 4; ++dy) {
      const f32x4u w4 = rv[dy];
      float e[4];
      e[0] = left ? 0.0f : (right ? w4[1] : w4[0]);
      e[1] = left ? w4[0] : (right ? w4[2] : w4[1]);
      e[2] = left ? w4[1] : (right ? w4[3] : w4[2]);
      e[3] = left ? w4[2] : (right ? 0.0f : w4[3]);
#pragma unroll
      for (int dx = 0; dx < 4; ++dx) {
        const float val = rok[dy] ? e[dx] : 0.0f;
        dd[dy * 4 + dx] = d.act ? apply_act(val, d.act) : val;
      }
    }
  } else {
    const float* pc = x + pbase;
#pragma unroll
    for (int dy = 0; dy < 4; ++dy)
#pragma unroll
      for (int dx = 0; dx < 4; ++dx) {
        const int iy = oy - 1 + dy, ix = ox - 1 + dx;
        const float val = (cok && (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W) ? pc[dy * d.W + dx] : 0.0f;
        dd[dy * 4 + dx] = d.act ? apply_act(val, d.act) : val;
      }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    u[0 * 4 + j] = dd[0 * 4 + j] - dd[2 * 4 + j];
    u[1 * 4 + j] = dd[1 * 4 + j] + dd[2 * 4 + j];
    u[2 * 4 + j] = dd[2 * 4 + j] - dd[1 * 4 + j];
    u[3 * 4 + j] = dd[1 * 4 + j] - dd[3 * 4 + j];
  }
  float* dst = v + ((long long)pblk * d.nchunks + chunk) * 4096 + ci_l * 32 + t;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    dst[(i * 4 + 0) * 256] = u[i * 4 + 0] - u[i * 4 + 2];
    dst[(i * 4 + 1) * 256] = u[i * 4 + 1] + u[i * 4 + 2];
    dst[(i * 4 + 2) * 256] = u[i * 4 + 2] - u[i * 4 + 1];
    dst[(i * 4 + 3) * 256] = u[i * 4 + 1] - u[i * 4 + 3];
  }
}

struct WinoGeom {
  int lgTX, lgTY, lgTI, tiles_x, tiles_y, tiles_n, nchunks8, nsteps;
};
static WinoGeom wino_geometry(const icm_conv_args& a) {
  WinoGeom g;
  const int tw = cdiv(a.W, 2), th = cdiv(a.H, 2);   // tiles per image row / column
  g.lgTX = std::min(3, ceil_log2(tw));
  g.lgTY = std::min(5 - g.lgTX, ceil_log2(th));
  g.lgTI = 5 - g.lgTX - g.lgTY;
  g.tiles_x = cdiv(tw, 1 << g.lgTX); g.tiles_y = cdiv(th, 1 << g.lgTY); g.tiles_n = cdiv(a.N, 1 << g.lgTI);
  g.nchunks8 = cdiv(a.Cin, 8); g.nsteps = cdiv(g.nchunks8, 2);
  return g;
}

long long wino_transform_floats(const icm_conv_args& a) {
  const WinoGeom g = wino_geometry(a);
  return (long long)g.tiles_x * g.tiles_y * g.tiles_n * (g.nsteps * 2) * 4096;
}

int run_wino_transform(const icm_conv_args* arr, int ngroups, hipStream_t stream) {
  const icm_conv_args& a = arr[0];
  if (!wino_supported(a)) return ICM_ERR_UNSUPPORTED;
  const WinoGeom g = wino_geometry(a);
  WinoXfDesc d{};
  for (int gi = 0; gi < ICM_MAX_GROUPS; ++gi) {
    const icm_conv_args& s = arr[gi < ngroups ? gi : 0];
    if (!s.x || !s.xv) return ICM_ERR_ARG;
    d.x[gi] = s.x; d.v[gi] = s.xv;
  }
  d.x_bs = a.x_bs; d.N = a.N; d.Cin = a.Cin; d.H = a.H; d.W = a.W;
  d.lgTX = g.lgTX; d.lgTY = g.lgTY; d.lgTI = g.lgTI; d.tiles_x = g.tiles_x; d.tiles_y = g.tiles_y;
  d.nchunks = g.nsteps * 2; d.act = a.pro_act;
  d.seg_len = a.x_seg_len; d.seg_gap = a.x_seg_len ? a.x_seg_gap : 0;
  d.dseg = make_fastdiv((uint32_t)std::max(1, a.x_seg_len));
  const long long nblk = (long long)g.tiles_x * g.tiles_y * g.tiles_n * d.nchunks;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return ICM_ERR_ARG;
  static const bool novec = [] { const char* e = getenv("ICM_WINO_XF_NOVEC"); return e && atoi(e) != 0; }();   // measurement only
  if (!novec && a.W % 2 == 0 && a.W >= 4)
    hipLaunchKernelGGL(wino_input_transform_kernel<true>, dim3((unsigned)nblk, ngroups), dim3(256), 0, stream, d);
  else
    hipLaunchKernelGGL(wino_input_transform_kernel<false>, dim3((unsigned)nblk, ngroups), dim3(256), 0, stream, d);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}

bool wino_supported(const icm_conv_args& a) {
  if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.pixel_shuffle) return false;
  if (a.OH != a.H || a.OW != a.W) return false;
  switch (a.epi) {
    case ICM_EPI_NONE: case ICM_EPI_RES: case ICM_EPI_RES_GELU: case ICM_EPI_MUL_DGELU: case ICM_EPI_RES_MUL_DGELU:
    case ICM_EPI_LRP: break;
    default: return false;
  }
  if (a.pro_act != ICM_ACT_NONE && a.pro_act != ICM_ACT_GELU && a.pro_act != ICM_ACT_SQUARE) return false;
  return true;
}

int run_conv_wino(const icm_conv_args* arr, int ngroups, hipStream_t stream) {
  const icm_conv_args& a = arr[0];
  if (!wino_supported(a)) return ICM_ERR_UNSUPPORTED;
  WinoDesc d{};
  for (int gi = 0; gi < ICM_MAX_GROUPS; ++gi) {
    const icm_conv_args& s = arr[gi < ngroups ? gi : 0];
    d.g[gi].x = s.x; d.g[gi].wp = s.wp; d.g[gi].bias = s.bias; d.g[gi].y = s.y;
    d.g[gi].res = s.res; d.g[gi].aux = s.aux; d.g[gi].aux2 = nullptr; d.g[gi].y2 = s.y2;
  }
  d.x_bs = a.x_bs; d.y_bs = a.y_bs; d.res_bs = a.res_bs; d.aux_bs = a.aux_bs; d.y2_bs = a.y2_bs;
  d.N = a.N; d.Cin = a.Cin; d.Cout = a.Cout; d.H = a.H; d.W = a.W;
  {
    const WinoGeom g = wino_geometry(a);
    d.lgTX = g.lgTX; d.lgTY = g.lgTY; d.lgTI = g.lgTI; d.tiles_x = g.tiles_x; d.tiles_y = g.tiles_y; d.tiles_n = g.tiles_n;
    d.nchunks8 = g.nchunks8; d.nsteps = g.nsteps;
  }
  d.ncot = cdiv(a.Cout, 32);
  d.vpre = a.xv != nullptr ? 1 : 0;
  if (d.vpre)   // the operand of every member is its pre-transformed buffer
    for (int gi = 0; gi < ICM_MAX_GROUPS; ++gi) {
      const icm_conv_args& s = arr[gi < ngroups ? gi : 0];
      if (!s.xv) return ICM_ERR_ARG;
      d.g[gi].x = s.xv;
    }
  d.epi = a.epi; d.accum = a.accum; d.act = a.pro_act;
  d.seg_len = a.x_seg_len; d.seg_gap = a.x_seg_len ? a.x_seg_gap : 0;
  d.dseg = make_fastdiv((uint32_t)std::max(1, a.x_seg_len));
  const long long pblocks = (long long)d.tiles_x * d.tiles_y * d.tiles_n;
  // co tiles per workgroup: 2 halves the activation staging per output; 1 gives twice the workgroups (small launches)
  static const int force_tco = getenv("ICM_WINO_TCO") ? atoi(getenv("ICM_WINO_TCO")) : 0;
  int tco = 2;
  {
    const long long b2 = pblocks * cdiv(d.ncot, 2) * ngroups, b1 = pblocks * d.ncot * ngroups;
    const double t2 = std::ceil(b2 / 256.0) * 2.0, t1 = std::ceil(b1 / 256.0) * 1.0 * 1.08;
    if (d.ncot == 1 || t1 < t2) tco = 1;
    if (force_tco == 1 || force_tco == 2) tco = force_tco;
    (void)b1;
  }
  // pre-transformed operand: the eight-MFMA-wave kernel, TCO in {2, 3, 4}: whole rounds of the chip, then wide co blocks
  static const int w8_on = getenv("ICM_WINO8") ? atoi(getenv("ICM_WINO8")) : 1;
  bool w8 = d.vpre && w8_on;
  if (w8) {
    int t8 = 2;
    double bestc = 1e300;
    for (int t = 4; t >= 2; --t) {
      const long long b = pblocks * cdiv(d.ncot, t) * ngroups;
      const double c = std::ceil(b / 256.0) * (t + 0.6);
      if (c < bestc - 1e-9) { bestc = c; t8 = t; }
    }
    if (force_tco >= 2 && force_tco <= 4) t8 = force_tco;
    // launches that cannot give the wide workgroups a (nearly) full round of the chip keep the 4 + 4 kernel with its
    // narrower co blocks (measured: 224 -> 176 single 46.6 vs 54.4 us, 176 -> 128 x2 41.9 vs 49.9 us, 3360 -> 160 607 vs 746 us)
    static const long long w8_min = getenv("ICM_WINO8_MINWG") ? atoll(getenv("ICM_WINO8_MINWG")) : 150;
    if (pblocks * cdiv(d.ncot, t8) * ngroups >= w8_min || force_tco >= 2) tco = t8;
    else w8 = false;
  }
  d.ncb = cdiv(d.ncot, tco);
  d.npx = (int)pblocks;
  static const int dbg = getenv("ICM_WINO_DEBUG") ? atoi(getenv("ICM_WINO_DEBUG")) : 0;
  d.dbg = dbg;
  {
    const double wbytes = 64.0 * a.Cin * a.Cout, abytes = 4.0 * a.Cin * a.N * a.H * a.W;
    static const int force_order = getenv("ICM_WINO_PXFAST") ? atoi(getenv("ICM_WINO_PXFAST")) : -1;
    d.px_fast = (wbytes > 3.0e6 && wbytes > abytes) ? 1 : 0;
    if (force_order == 0 || force_order == 1) d.px_fast = force_order;
  }
  const long long nblk = pblocks * d.ncb;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return ICM_ERR_ARG;
  void (*fn)(const WinoDesc) = tco == 2 ? conv_wino_kernel<2> : conv_wino_kernel<1>;
  size_t lds = (size_t)2 * WINO_STEP_FLOATS * sizeof(float);   // 64 KB: two staging steps; the output transform reuses it
  if (w8) {
    fn = tco == 4 ? conv_wino8_kernel<4> : (tco == 3 ? conv_wino8_kernel<3> : conv_wino8_kernel<2>);
    lds = (size_t)2 * 4 * 2 * 2 * 16 * 64 * sizeof(float);      // 128 KB: one output-transform pass of two co tiles
    if (!ensure_max_lds(reinterpret_cast<const void*>(fn))) return ICM_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(fn, dim3((unsigned)nblk, ngroups, 1), dim3(512), lds, stream, d);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}

}  // namespace icm
