// Window multi-head self-attention core on the matrix cores (8x8 windows: 64 tokens per window), gfx950.
//
// One wave = one (window, head): Q^T, K^T, V^T (and dO^T in the backward pass) of the head sit in LDS as [d][token]
// slabs (row stride 65: the token-major reads of an MFMA A operand are conflict-free, the channel-major reads of a
// B operand are consecutive), straight from the NCHW qkv tensor -- cyclic shift, window partition and head split are
// address arithmetic as in winattn.hip.  All contractions are v_mfma_f32_32x32x2_f32 (exact f32):
//
//   S^T = K Q^T   (64 x 64, K = hd)      column = query i, so a query's 64 scores live in 32 registers of lanes i and
//                                        i + 32: the softmax is an in-lane reduction plus one cross-half exchange
//   O^T = V^T P^T (hd x 64, K = 64)      P^T is used as the MFMA B operand IN PLACE: register r of the accumulator tile
//                                        holds rows j_r (lanes 0-31) and j_r + 4 (lanes 32-63), exactly the (k = 0, k = 1)
//                                        pair of a K = 2 step, so the A operand reads V^T[d][j_r + 4 (lane >> 5)] and no
//                                        probability ever moves between lanes or through memory
//
// and the backward pass adds dP^T = V dO^T, dQ^T = K^T dS^T, and -- with the (i, j) roles of the tiles swapped, which
// the MFMA gives for free by swapping its operands -- S = Q K^T, dP = dO V^T, dV^T = dO^T P, dK^T = Q^T dS.
// Relative-position-bias gradients: each lane adds its dS entries into the head's LDS table, lanes 0-31 first, then
// lanes 32-63 (within a half all table entries of one register are distinct), so the sum order is fixed; the
// per-(window, head) tables go to the same workspace slabs winattn.hip reduces in window order.
#include "icm_common.h"

namespace icm {

struct WaDesc;   // winattn.hip

struct WmDesc {
  const float* qkv;
  const float* table;
  float* out;          // fwd
  const float* dout;   // bwd
  float* dqkv;         // bwd
  float* dtable_ws;    // bwd
  int N, C, H, W, heads, shift, nwx, nwy;
  float scale;
};

#define WM_WS 8
#define WM_T 64
#define WM_TS 65
#define WM_NTAB 225   // (2 * 8 - 1)^2

__device__ __forceinline__ int wm_region(int s, int L, int shift) { return s < L - WM_WS ? 0 : (s < L - shift ? 1 : 2); }

// pixel offset (oy * W + ox) and shift-mask label of token t of window (wy, wx)
__device__ __forceinline__ void wm_token(const WmDesc& d, int wy, int wx, int t, int& pix, int& lab) {
  const int r = t >> 3, c = t & 7;
  const int sy = wy * WM_WS + r, sx = wx * WM_WS + c;
  int oy = sy + d.shift, ox = sx + d.shift;
  if (oy >= d.H) oy -= d.H;
  if (ox >= d.W) ox -= d.W;
  pix = oy * d.W + ox;
  lab = d.shift > 0 ? wm_region(sy, d.H, d.shift) * 3 + wm_region(sx, d.W, d.shift) : 0;
}

// rows of a 32x32 accumulator tile held by register r of a lane in half h: (r & 3) + 8 (r >> 2) + 4 h
__device__ __forceinline__ int wm_row(int r) { return (r & 3) + 8 * (r >> 2); }
// relative-position index of (query i, key j), both 0..63 (win_attention.py:64-74)
__device__ __forceinline__ int wm_relidx(int i, int j) {
  return ((i >> 3) - (j >> 3) + WM_WS - 1) * (2 * WM_WS - 1) + ((i & 7) - (j & 7) + WM_WS - 1);
}

// acc[x][y] (+)= sum_d A^T[d][x-tile rows] * B^T[d][y-tile cols]: rows = tokens of slab `rowT`, cols = tokens of slab
// `colT` (both [HD][WM_TS]); 2 x 2 tiles, HD / 2 MFMA steps each
template <int HD>
__device__ __forceinline__ void wm_tok_tok(const float* __restrict__ rowT, const float* __restrict__ colT, int lane,
                                           f32x16 (&acc)[2][2]) {
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.0f;
#pragma unroll
  for (int ks = 0; ks < HD / 2; ++ks) {
    const int off = (2 * ks + h) * WM_TS + l31;
    const float a0 = rowT[off], a1 = rowT[off + 32], b0 = colT[off], b1 = colT[off + 32];
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
  }
}

// out^T[d][y] = sum_x chT[d][x] * M[x][y] with M = acc (rows x in registers, cols y in lanes) used in place as the B
// operand: ND d-tiles of 32 rows (rows >= HD are computed on clamped addresses and never stored)
template <int HD, int ND>
__device__ __forceinline__ void wm_ch_tok(const float* __restrict__ chT, const f32x16 (&M)[2][2], int lane,
                                          f32x16 (&o)[ND][2]) {
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int dt = 0; dt < ND; ++dt)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][y][r] = 0.0f;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int tok = x * 32 + wm_row(r) + 4 * h;
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) {
        const int dd = min(dt * 32 + l31, HD - 1);
        const float a = chT[dd * WM_TS + tok];
        o[dt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, M[x][0][r], o[dt][0], 0, 0, 0);
        o[dt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, M[x][1][r], o[dt][1], 0, 0, 0);
      }
    }
}

// store o^T tiles (rows = head channels d, cols = tokens) to a [*, H*W] channel-plane tensor at channel base `cb`
template <int HD, int ND>
__device__ __forceinline__ void wm_store(float* __restrict__ base, long long HW, const int (&pix)[2], int lane,
                                         const f32x16 (&o)[ND][2], float mul) {
  const int h = lane >> 5;
#pragma unroll
  for (int dt = 0; dt < ND; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dd = dt * 32 + wm_row(r) + 4 * h;
      if (dd < HD) {
        base[(long long)dd * HW + pix[0]] = o[dt][0][r] * mul;
        base[(long long)dd * HW + pix[1]] = o[dt][1][r] * mul;
      }
    }
}

// stage NS [HD][64] slabs (lane = token): ALL loads are issued before the first LDS store -- the wave is alone on its
// SIMD while it loads, so the NS * HD loads overlap each other instead of completing in small dependent batches
template <int HD, int NS>
__device__ __forceinline__ void wm_load_all(const float* const (&src)[NS], long long HW, int pix, int lane, float mul0,
                                            float* __restrict__ dst0) {
  float v[NS][HD];
#pragma unroll
  for (int k = 0; k < NS; ++k)
#pragma unroll
    for (int dd = 0; dd < HD; ++dd) v[k][dd] = src[k][(long long)dd * HW + pix];
#pragma unroll
  for (int k = 0; k < NS; ++k)
#pragma unroll
    for (int dd = 0; dd < HD; ++dd) dst0[(k * HD + dd) * WM_TS + lane] = k == 0 ? v[k][dd] * mul0 : v[k][dd];
}

__device__ __forceinline__ float wm_xhalf(float v) { return __shfl_xor(v, 32, 64); }

// scores -> probabilities in place, S^T orientation: acc[jt][it][r] = score of (key j = jt*32 + row(r) + 4h, query
// i = it*32 + l31).  Adds bias + mask; returns per-query max / sum through m, l (two queries per lane: it = 0, 1).
__device__ __forceinline__ void wm_softmax_T(f32x16 (&acc)[2][2], const float* __restrict__ bias, const int (&labq)[2],
                                             const unsigned char* __restrict__ labs, bool masked, int lane,
                                             float (&m)[2], float (&linv)[2]) {
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int i = it * 32 + l31;
    float mx = -3.0e38f;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = jt * 32 + wm_row(r) + 4 * h;
        float s = acc[jt][it][r] + bias[wm_relidx(i, j)];
        if (masked && labs[j] != labq[it]) s += -100.0f;
        acc[jt][it][r] = s;
        mx = fmaxf(mx, s);
      }
    mx = fmaxf(mx, wm_xhalf(mx));
    float sum = 0.0f;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = expf(acc[jt][it][r] - mx);
        acc[jt][it][r] = p;
        sum += p;
      }
    sum += wm_xhalf(sum);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[jt][it][r] *= inv;
    m[it] = mx;
    linv[it] = inv;
  }
}

// per-wave LDS: NS slabs [HD][65] + bias table (225, padded to 228) + stats
template <int HD, int NS>
struct WmLds {
  static constexpr int kSlab = HD * WM_TS;
  static constexpr int kFloats = NS * kSlab + 228 + (NS > 3 ? 228 + 3 * WM_T : 0);
};

template <int HD>
__global__ __launch_bounds__(256, 2) void winattn_mfma_fwd_kernel(const WmDesc d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int ND = (HD + 31) / 32;
  using LDS = WmLds<HD, 3>;
  int* pixs = reinterpret_cast<int*>(smem + 4 * LDS::kFloats);                     // [64] pixel offset of every token
  unsigned char* labs = reinterpret_cast<unsigned char*>(pixs + WM_T);             // [64] shift-mask region labels
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* Qt = smem + wave * LDS::kFloats;
  float* Kt = Qt + LDS::kSlab;
  float* Vt = Kt + LDS::kSlab;
  float* bias = Vt + LDS::kSlab;
  int bid;
  {   // XCD-aware window order (see winattn.hip)
    const int nb = gridDim.x, hb = blockIdx.x;
    const int xcd = hb & 7, q = hb >> 3;
    bid = xcd * (nb >> 3) + min(xcd, nb & 7) + q;
  }
  const int wx = bid % d.nwx; bid /= d.nwx;
  const int wy = bid % d.nwy;
  const int n = bid / d.nwy;
  const long long HW = (long long)d.H * d.W;
  if (tid < WM_T) {
    int px, lb;
    wm_token(d, wy, wx, tid, px, lb);
    pixs[tid] = px;
    labs[tid] = (unsigned char)lb;
  }
  __syncthreads();
  const int l31 = lane & 31;
  const int mypix = pixs[lane];
  const int pix2[2] = {pixs[l31], pixs[32 + l31]};
  const int labq[2] = {labs[l31], labs[32 + l31]};
  const float* base = d.qkv + (long long)n * 3 * d.C * HW;
  const bool masked = d.shift > 0;
  for (int head = wave; head < d.heads; head += 4) {
    {
      const float* const srcs[3] = {base + (long long)(head * HD) * HW, base + (long long)(d.C + head * HD) * HW,
                                    base + (long long)(2 * d.C + head * HD) * HW};
      wm_load_all<HD, 3>(srcs, HW, mypix, lane, d.scale, Qt);   // Qt | Kt | Vt are consecutive slabs
    }
    for (int e = lane; e < WM_NTAB; e += 64) bias[e] = d.table[e * d.heads + head];
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    f32x16 st[2][2];
    wm_tok_tok<HD>(Kt, Qt, lane, st);           // S^T: rows = keys, cols = queries
    float m[2], linv[2];
    wm_softmax_T(st, bias, labq, labs, masked, lane, m, linv);
    f32x16 o[ND][2];
    wm_ch_tok<HD, ND>(Vt, st, lane, o);         // O^T = V^T P^T
    wm_store<HD, ND>(d.out + ((long long)n * d.C + head * HD) * HW, HW, pix2, lane, o, 1.0f);
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();            // slabs are rewritten by the next head
  }
}

template <int HD>
__global__ __launch_bounds__(256, 1) void winattn_mfma_bwd_kernel(const WmDesc d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int ND = (HD + 31) / 32;
  using LDS = WmLds<HD, 4>;
  int* pixs = reinterpret_cast<int*>(smem + 4 * LDS::kFloats);
  unsigned char* labs = reinterpret_cast<unsigned char*>(pixs + WM_T);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* Qt = smem + wave * LDS::kFloats;
  float* Kt = Qt + LDS::kSlab;
  float* Vt = Kt + LDS::kSlab;
  float* Gt = Vt + LDS::kSlab;          // dO^T
  float* bias = Gt + LDS::kSlab;
  float* dbias = bias + 228;            // table gradient of this (window, head)
  float* stat_m = dbias + 228;          // per query: max, 1 / sum, delta = sum_j P dP
  float* stat_l = stat_m + WM_T;
  float* stat_d = stat_l + WM_T;
  int bid;
  {
    const int nb = gridDim.x, hb = blockIdx.x;
    const int xcd = hb & 7, q = hb >> 3;
    bid = xcd * (nb >> 3) + min(xcd, nb & 7) + q;
  }
  const int win = bid;
  const int wx = bid % d.nwx; bid /= d.nwx;
  const int wy = bid % d.nwy;
  const int n = bid / d.nwy;
  const long long HW = (long long)d.H * d.W;
  if (tid < WM_T) {
    int px, lb;
    wm_token(d, wy, wx, tid, px, lb);
    pixs[tid] = px;
    labs[tid] = (unsigned char)lb;
  }
  __syncthreads();
  const int l31 = lane & 31, h = lane >> 5;
  const int mypix = pixs[lane];
  const int pix2[2] = {pixs[l31], pixs[32 + l31]};
  const int labq[2] = {labs[l31], labs[32 + l31]};
  const float* base = d.qkv + (long long)n * 3 * d.C * HW;
  float* dbase = d.dqkv + (long long)n * 3 * d.C * HW;
  const bool masked = d.shift > 0;
  for (int head = wave; head < d.heads; head += 4) {
    {
      const float* const srcs[4] = {base + (long long)(head * HD) * HW, base + (long long)(d.C + head * HD) * HW,
                                    base + (long long)(2 * d.C + head * HD) * HW,
                                    d.dout + ((long long)n * d.C + head * HD) * HW};
      wm_load_all<HD, 4>(srcs, HW, mypix, lane, d.scale, Qt);   // Qt | Kt | Vt | dO^T are consecutive slabs
    }
    for (int e = lane; e < WM_NTAB; e += 64) {
      bias[e] = d.table[e * d.heads + head];
      dbias[e] = 0.0f;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    // ---- column = query orientation: P^T, dP^T -> delta, dS^T -> table gradient, dQ^T
    {
      f32x16 pt[2][2];
      wm_tok_tok<HD>(Kt, Qt, lane, pt);
      float m[2], linv[2];
      wm_softmax_T(pt, bias, labq, labs, masked, lane, m, linv);
      if (h == 0) {
        stat_m[l31] = m[0]; stat_m[32 + l31] = m[1];
        stat_l[l31] = linv[0]; stat_l[32 + l31] = linv[1];
      }
      f32x16 dpt[2][2];
      wm_tok_tok<HD>(Vt, Gt, lane, dpt);          // dP^T[j][i] = sum_d V[j][d] dO[i][d]
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        float del = 0.0f;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
          for (int r = 0; r < 16; ++r) del += pt[jt][it][r] * dpt[jt][it][r];
        del += wm_xhalf(del);
        if (h == 0) stat_d[it * 32 + l31] = del;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
          for (int r = 0; r < 16; ++r) pt[jt][it][r] *= dpt[jt][it][r] - del;   // dS^T
      }
      // table gradient into the wave's own LDS table.  Different lanes hit the same entry in different
      // instructions, so the adds are LDS atomics (plain read-modify-writes may legally be reordered per thread); the
      // table is private to this wave and, within one instruction, the 32 lanes of a half hit 32 distinct entries
      // (halves take turns), so the order of every entry's additions is the program order: bitwise reproducible.
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        if (h == hh) {
#pragma unroll
          for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int idx = wm_relidx(it * 32 + l31, jt * 32 + wm_row(r) + 4 * h);
                __hip_atomic_fetch_add(dbias + idx, pt[jt][it][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
              }
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
      }
      f32x16 o[ND][2];
      wm_ch_tok<HD, ND>(Kt, pt, lane, o);         // dQ^T[d][i] = sum_j K[j][d] dS[i][j]
      wm_store<HD, ND>(dbase + (long long)(head * HD) * HW, HW, pix2, lane, o, d.scale);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();              // stats visible to the whole wave
    // ---- column = key orientation: P, dV^T, dP -> dS, dK^T
    {
      f32x16 p[2][2];                             // p[it][jt][r]: query i = it*32 + row(r) + 4h, key j = jt*32 + l31
      wm_tok_tok<HD>(Qt, Kt, lane, p);
#pragma unroll
      for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = it * 32 + wm_row(r) + 4 * h;
          const float mi = stat_m[i], li = stat_l[i];
          const int labi = labs[i];
#pragma unroll
          for (int jt = 0; jt < 2; ++jt) {
            const int j = jt * 32 + l31;
            float s = p[it][jt][r] + bias[wm_relidx(i, j)];
            if (masked && labq[jt] != labi) s += -100.0f;
            p[it][jt][r] = expf(s - mi) * li;
          }
        }
      {
        f32x16 o[ND][2];
        wm_ch_tok<HD, ND>(Gt, p, lane, o);        // dV^T[d][j] = sum_i dO[i][d] P[i][j]
        wm_store<HD, ND>(dbase + (long long)(2 * d.C + head * HD) * HW, HW, pix2, lane, o, 1.0f);
      }
      f32x16 dp[2][2];
      wm_tok_tok<HD>(Gt, Vt, lane, dp);           // dP[i][j] = sum_d dO[i][d] V[j][d]
#pragma unroll
      for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float del = stat_d[it * 32 + wm_row(r) + 4 * h];
#pragma unroll
          for (int jt = 0; jt < 2; ++jt) p[it][jt][r] *= dp[it][jt][r] - del;   // dS
        }
      f32x16 o[ND][2];
      wm_ch_tok<HD, ND>(Qt, p, lane, o);          // dK^T[d][j] = sum_i (scale q)[i][d] dS[i][j]
      wm_store<HD, ND>(dbase + (long long)(d.C + head * HD) * HW, HW, pix2, lane, o, 1.0f);
    }
    {
      float* slab = d.dtable_ws + ((long long)win * d.heads + head) * WM_NTAB;
      for (int e = lane; e < WM_NTAB; e += 64) slab[e] = dbias[e];
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
  }
}

typedef void (*WmFn)(const WmDesc);
static bool wm_pick(int hd, WmFn& f, WmFn& b, size_t& lf, size_t& lb) {
  switch (hd) {
#define C_(n)                                               \
  case n:                                                   \
    f = winattn_mfma_fwd_kernel<n>;                         \
    b = winattn_mfma_bwd_kernel<n>;                         \
    lf = ((size_t)4 * WmLds<n, 3>::kFloats + WM_T + WM_T / 4) * sizeof(float);  \
    lb = ((size_t)4 * WmLds<n, 4>::kFloats + WM_T + WM_T / 4) * sizeof(float);  \
    return true;
    C_(8) C_(16) C_(24) C_(32) C_(48)
#undef C_
    default: return false;
  }
}

static bool wm_fill(WmDesc& d, int N, int C, int H, int W, int heads, int ws, int shift) {
  if (ws != WM_WS || heads <= 0 || C % heads || H % ws || W % ws || shift < 0 || shift >= ws) return false;
  d.N = N; d.C = C; d.H = H; d.W = W; d.heads = heads; d.shift = shift;
  d.nwx = W / ws; d.nwy = H / ws;
  d.scale = 1.0f / sqrtf((float)(C / heads));
  return true;
}

// returns ICM_OK when the MFMA path took the launch, -1 when this geometry is not covered (caller falls through to
// the generic kernel of winattn.hip), an ICM_ERR_* code on failure
int winattn_mfma_fwd(const float* qkv, const float* table, float* out, int N, int C, int H, int W, int heads, int ws,
                     int shift, hipStream_t stream) {
  WmDesc d{};
  WmFn f, b;
  size_t lf, lb;
  if (!wm_fill(d, N, C, H, W, heads, ws, shift) || !wm_pick(C / heads, f, b, lf, lb)) return -1;
  d.qkv = qkv; d.table = table; d.out = out;
  if (lf > 64 * 1024 && !ensure_max_lds(reinterpret_cast<const void*>(f))) return ICM_ERR_LAUNCH;
  hipLaunchKernelGGL(f, dim3(N * d.nwy * d.nwx), dim3(256), lf, stream, d);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}

int winattn_mfma_bwd(const float* qkv, const float* table, const float* dout, float* dqkv, float* dtable_ws, int N, int C,
                     int H, int W, int heads, int ws, int shift, hipStream_t stream) {
  WmDesc d{};
  WmFn f, b;
  size_t lf, lb;
  if (!wm_fill(d, N, C, H, W, heads, ws, shift) || !wm_pick(C / heads, f, b, lf, lb)) return -1;
  d.qkv = qkv; d.table = table; d.dout = dout; d.dqkv = dqkv; d.dtable_ws = dtable_ws;
  if (lb > 160 * 1024) return -1;
  if (lb > 64 * 1024 && !ensure_max_lds(reinterpret_cast<const void*>(b))) return ICM_ERR_LAUNCH;
  hipLaunchKernelGGL(b, dim3(N * d.nwy * d.nwx), dim3(256), lb, stream, d);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}

}  // namespace icm

// =====================================================================================================================
// 4x4 windows (16 tokens): stf's Swin blocks and the dim-320 gates.  v_mfma_f32_16x16x4_f32: one 16 x 16 tile per
// (window, head) for S^T = K Q^T, ceil(hd / 16) tiles for O^T = V^T P^T -- the probabilities again feed the second
// product straight from the accumulator registers (C/D: col = lane & 15, row = 4 (lane >> 4) + r, so register r holds
// rows {r, 4 + r, 8 + r, 12 + r} = the four k-slots of one K = 4 step; the A operand reads V^T[d][4 (lane >> 4) + r]).
// A wave owns FOUR horizontally adjacent windows: its global loads then cover 16 consecutive pixels per row (64-byte
// segments instead of the 16-byte rows of a single 4-pixel window); the heads of the four windows go through LDS
// slabs [window][d][token] (row stride 17), one MFMA sequence per window.
namespace icm {

#define W4_T 16
#define W4_TS 17
#define W4_NTAB 49   // (2 * 4 - 1)^2

struct W4Desc {
  const float* qkv;
  const float* table;
  float* out;
  const float* dout;
  float* dqkv;
  float* dtable_ws;
  int N, C, H, W, heads, shift, ngx, nwy;   // ngx = groups of 4 windows per row
  float scale;
};

typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int w4_region(int s, int L, int shift) { return s < L - 4 ? 0 : (s < L - shift ? 1 : 2); }

// acc (+)= sum_d rowT[d][row token] * colT[d][col token] for ONE window slab pair ([HD][17] each)
template <int HD>
__device__ __forceinline__ f32x4v w4_tok_tok(const float* __restrict__ rowT, const float* __restrict__ colT, int lane) {
  const int t = lane & 15, g = lane >> 4;
  f32x4v acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int ks = 0; ks < HD / 4; ++ks) {
    const int off = (4 * ks + g) * W4_TS + t;
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(rowT[off], colT[off], acc, 0, 0, 0);
  }
  return acc;
}
// o[dt] = sum_x chT[d][x] * M[x][col]: M's rows x = 4 g + r sit in register r of lane group g
template <int HD, int ND>
__device__ __forceinline__ void w4_ch_tok(const float* __restrict__ chT, const f32x4v M, int lane, f32x4v (&o)[ND]) {
  const int t = lane & 15, g = lane >> 4;
#pragma unroll
  for (int dt = 0; dt < ND; ++dt) {
    o[dt] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
    const int dd = min(dt * 16 + t, HD - 1);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(chT[dd * W4_TS + 4 * g + r], M[r], o[dt], 0, 0, 0);
  }
}
template <int HD, int ND>
__device__ __forceinline__ void w4_store(float* __restrict__ base, long long HW, int pix, int lane, const f32x4v (&o)[ND],
                                         float mul) {
  const int g = lane >> 4;
#pragma unroll
  for (int dt = 0; dt < ND; ++dt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int dd = dt * 16 + 4 * g + r;
      if (dd < HD) base[(long long)dd * HW + pix] = o[dt][r] * mul;
    }
}
__device__ __forceinline__ float w4_xgroups(float v, bool is_max) {
  const float a = __shfl_xor(v, 16, 64);
  v = is_max ? fmaxf(v, a) : v + a;
  const float b = __shfl_xor(v, 32, 64);
  return is_max ? fmaxf(v, b) : v + b;
}
__device__ __forceinline__ int w4_relidx(int i, int j) { return ((i >> 2) - (j >> 2) + 3) * 7 + ((i & 3) - (j & 3) + 3); }

// per-wave LDS: NS slabs x 4 windows x [HD][17], bias table (49 -> 52), [bwd: table gradient 52, stats 3 x 64]
template <int HD, int NS>
struct W4Lds {
  static constexpr int kSlab = HD * W4_TS;          // one window, one tensor
  static constexpr int kFloats = NS * 4 * kSlab + 52 + (NS > 3 ? 52 + 3 * 64 : 0);
};

template <int HD, int NS>
__device__ __forceinline__ void w4_load_all(const float* const (&src)[NS], long long HW, int pix, int lane, float mul0,
                                            float* __restrict__ dst0) {
  // lane = (window w = lane >> 4, token t = lane & 15): slab (k, w) element [d][t]
  const int w = lane >> 4, t = lane & 15;
  float v[NS][HD];
#pragma unroll
  for (int k = 0; k < NS; ++k)
#pragma unroll
    for (int dd = 0; dd < HD; ++dd) v[k][dd] = src[k][(long long)dd * HW + pix];
#pragma unroll
  for (int k = 0; k < NS; ++k)
#pragma unroll
    for (int dd = 0; dd < HD; ++dd)
      dst0[((k * 4 + w) * HD + dd) * W4_TS + t] = k == 0 ? v[k][dd] * mul0 : v[k][dd];
}

// geometry of the wave's four windows: pixel / mask label of (window w, token t) for the lane's own (w, t)
__device__ __forceinline__ void w4_token(const W4Desc& d, int wy, int gx, int w, int t, int& pix, int& lab, bool& valid) {
  const int wx = gx * 4 + w;
  valid = wx * 4 < d.W;
  const int r = t >> 2, c = t & 3;
  const int sy = wy * 4 + r, sx = min(wx, d.W / 4 - 1) * 4 + c;
  int oy = sy + d.shift, ox = sx + d.shift;
  if (oy >= d.H) oy -= d.H;
  if (ox >= d.W) ox -= d.W;
  pix = oy * d.W + ox;
  lab = d.shift > 0 ? w4_region(sy, d.H, d.shift) * 3 + w4_region(sx, d.W, d.shift) : 0;
}

template <int HD>
__global__ __launch_bounds__(256) void winattn_mfma16_fwd_kernel(const W4Desc d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int ND = (HD + 15) / 16;
  using LDS = W4Lds<HD, 3>;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* slabs = smem + wave * LDS::kFloats;
  float* bias = slabs + 3 * 4 * LDS::kSlab;
  // one wave = (group of four windows, head): heads run in parallel waves -- the deep levels have few windows (16 x 16
  // maps: 64 groups) but many heads (24), and a serial head loop leaves the chip idle
  const int wt = blockIdx.x * 4 + wave;
  const int ntask = d.N * d.nwy * d.ngx;
  if (wt >= ntask * d.heads) return;
  const int head = wt % d.heads, task = wt / d.heads;
  const int gx = task % d.ngx;
  const int wy = (task / d.ngx) % d.nwy;
  const int n = task / (d.ngx * d.nwy);
  const long long HW = (long long)d.H * d.W;
  const int t = lane & 15, g = lane >> 4;
  int mypix, mylab;
  bool myvalid;
  w4_token(d, wy, gx, g, t, mypix, mylab, myvalid);   // load mapping: lane = (window g, token t)
  const float* base = d.qkv + (long long)n * 3 * d.C * HW;
  const bool masked = d.shift > 0;
  {
    {
      const float* const srcs[3] = {base + (long long)(head * HD) * HW, base + (long long)(d.C + head * HD) * HW,
                                    base + (long long)(2 * d.C + head * HD) * HW};
      w4_load_all<HD, 3>(srcs, HW, mypix, lane, d.scale, slabs);
    }
    if (lane < W4_NTAB) bias[lane] = d.table[lane * d.heads + head];
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
      // compute mapping: lane = (k-slot / row group g, token t) of window w
      int pixw, labq;
      bool validw;
      w4_token(d, wy, gx, w, t, pixw, labq, validw);
      if (!validw) break;   // wave-uniform: windows beyond the row end (image width not a multiple of 16)
      const float* Qt = slabs + (0 * 4 + w) * LDS::kSlab;
      const float* Kt = slabs + (1 * 4 + w) * LDS::kSlab;
      const float* Vt = slabs + (2 * 4 + w) * LDS::kSlab;
      f32x4v st = w4_tok_tok<HD>(Kt, Qt, lane);      // S^T: rows = keys j = 4 g + r, col = query i = t
      float mx = -3.0e38f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 4 * g + r;
        float s = st[r] + bias[w4_relidx(t, j)];
        if (masked) {
          int pj, lj; bool vj;
          w4_token(d, wy, gx, w, j, pj, lj, vj);
          if (lj != labq) s += -100.0f;
        }
        st[r] = s;
        mx = fmaxf(mx, s);
      }
      mx = w4_xgroups(mx, true);
      float sum = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) { st[r] = expf(st[r] - mx); sum += st[r]; }
      sum = w4_xgroups(sum, false);
      const float inv = 1.0f / sum;
#pragma unroll
      for (int r = 0; r < 4; ++r) st[r] *= inv;
      f32x4v o[ND];
      w4_ch_tok<HD, ND>(Vt, st, lane, o);
      w4_store<HD, ND>(d.out + ((long long)n * d.C + head * HD) * HW, HW, pixw, lane, o, 1.0f);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
  }
}

template <int HD>
__global__ __launch_bounds__(128) void winattn_mfma16_bwd_kernel(const W4Desc d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int ND = (HD + 15) / 16;
  using LDS = W4Lds<HD, 4>;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* slabs = smem + wave * LDS::kFloats;
  float* bias = slabs + 4 * 4 * LDS::kSlab;
  float* dbias = bias + 52;
  float* stat_m = dbias + 52;       // [64]: per (window, query): max, 1 / sum, delta
  float* stat_l = stat_m + 64;
  float* stat_d = stat_l + 64;
  const int wt = blockIdx.x * 2 + wave;
  const int ntask = d.N * d.nwy * d.ngx;
  if (wt >= ntask * d.heads) return;
  const int head = wt % d.heads, task = wt / d.heads;
  const int gx = task % d.ngx;
  const int wy = (task / d.ngx) % d.nwy;
  const int n = task / (d.ngx * d.nwy);
  const long long HW = (long long)d.H * d.W;
  const int t = lane & 15, g = lane >> 4;
  int mypix, mylab;
  bool myvalid;
  w4_token(d, wy, gx, g, t, mypix, mylab, myvalid);
  const float* base = d.qkv + (long long)n * 3 * d.C * HW;
  float* dbase = d.dqkv + (long long)n * 3 * d.C * HW;
  const bool masked = d.shift > 0;
  {
    {
      const float* const srcs[4] = {base + (long long)(head * HD) * HW, base + (long long)(d.C + head * HD) * HW,
                                    base + (long long)(2 * d.C + head * HD) * HW,
                                    d.dout + ((long long)n * d.C + head * HD) * HW};
      w4_load_all<HD, 4>(srcs, HW, mypix, lane, d.scale, slabs);
    }
    if (lane < W4_NTAB) { bias[lane] = d.table[lane * d.heads + head]; dbias[lane] = 0.0f; }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
      int pixw, labq;
      bool validw;
      w4_token(d, wy, gx, w, t, pixw, labq, validw);
      if (!validw) break;
      const float* Qt = slabs + (0 * 4 + w) * LDS::kSlab;
      const float* Kt = slabs + (1 * 4 + w) * LDS::kSlab;
      const float* Vt = slabs + (2 * 4 + w) * LDS::kSlab;
      const float* Gt = slabs + (3 * 4 + w) * LDS::kSlab;
      int labr[4];   // labels of tokens 4 g + r (the row tokens of this lane's accumulator registers)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int pj; bool vj;
        w4_token(d, wy, gx, w, 4 * g + r, pj, labr[r], vj);
      }
      // ---- column = query: P^T, dP^T, delta, dS^T, table gradient, dQ^T
      {
        f32x4v pt = w4_tok_tok<HD>(Kt, Qt, lane);
        float mx = -3.0e38f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float s = pt[r] + bias[w4_relidx(t, 4 * g + r)];
          if (masked && labr[r] != labq) s += -100.0f;
          pt[r] = s;
          mx = fmaxf(mx, s);
        }
        mx = w4_xgroups(mx, true);
        float sum = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { pt[r] = expf(pt[r] - mx); sum += pt[r]; }
        sum = w4_xgroups(sum, false);
        const float inv = 1.0f / sum;
        const f32x4v dpt = w4_tok_tok<HD>(Vt, Gt, lane);      // dP^T[j][i]
        float del = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { pt[r] *= inv; del += pt[r] * dpt[r]; }
        del = w4_xgroups(del, false);
        if (g == 0) { stat_m[w * 16 + t] = mx; stat_l[w * 16 + t] = inv; stat_d[w * 16 + t] = del; }
#pragma unroll
        for (int r = 0; r < 4; ++r) pt[r] *= dpt[r] - del;      // dS^T
        // table gradient: the 16 lanes of one group hit 16 distinct entries per register; groups take turns (LDS
        // atomics: cross-lane accumulation needs them, the turn order keeps every entry's sum order fixed)
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
          if (g == gg) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              __hip_atomic_fetch_add(dbias + w4_relidx(t, 4 * g + r), pt[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          }
          __builtin_amdgcn_s_waitcnt(0);
          __builtin_amdgcn_wave_barrier();
        }
        f32x4v o[ND];
        w4_ch_tok<HD, ND>(Kt, pt, lane, o);                     // dQ^T
        w4_store<HD, ND>(dbase + (long long)(head * HD) * HW, HW, pixw, lane, o, d.scale);
      }
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_wave_barrier();
      // ---- column = key: P, dV^T, dP, dS, dK^T   (rows = queries i = 4 g + r, col = key j = t)
      {
        f32x4v p = w4_tok_tok<HD>(Qt, Kt, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 4 * g + r;
          float s = p[r] + bias[w4_relidx(i, t)];
          if (masked && labr[r] != labq) s += -100.0f;          // labq = label of token t (here: the key)
          p[r] = expf(s - stat_m[w * 16 + i]) * stat_l[w * 16 + i];
        }
        {
          f32x4v o[ND];
          w4_ch_tok<HD, ND>(Gt, p, lane, o);                    // dV^T
          w4_store<HD, ND>(dbase + (long long)(2 * d.C + head * HD) * HW, HW, pixw, lane, o, 1.0f);
        }
        const f32x4v dp = w4_tok_tok<HD>(Gt, Vt, lane);         // dP[i][j]
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] *= dp[r] - stat_d[w * 16 + 4 * g + r];
        f32x4v o[ND];
        w4_ch_tok<HD, ND>(Qt, p, lane, o);                      // dK^T
        w4_store<HD, ND>(dbase + (long long)(d.C + head * HD) * HW, HW, pixw, lane, o, 1.0f);
      }
    }
    if (lane < W4_NTAB) d.dtable_ws[((long long)task * d.heads + head) * W4_NTAB + lane] = dbias[lane];
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
  }
}

typedef void (*W4Fn)(const W4Desc);
static bool w4_pick(int hd, W4Fn& f, W4Fn& b, size_t& lf, size_t& lb) {
  switch (hd) {
#define C_(n)                                                 \
  case n:                                                     \
    f = winattn_mfma16_fwd_kernel<n>;                         \
    b = winattn_mfma16_bwd_kernel<n>;                         \
    lf = (size_t)4 * W4Lds<n, 3>::kFloats * sizeof(float);    \
    lb = (size_t)2 * W4Lds<n, 4>::kFloats * sizeof(float);    \
    return true;
    C_(8) C_(16) C_(24) C_(32) C_(40)
#undef C_
    default: return false;
  }
}
static bool w4_fill(W4Desc& d, int N, int C, int H, int W, int heads, int ws, int shift) {
  if (ws != 4 || heads <= 0 || C % heads || H % 4 || W % 4 || shift < 0 || shift >= 4) return false;
  d.N = N; d.C = C; d.H = H; d.W = W; d.heads = heads; d.shift = shift;
  d.ngx = (W / 4 + 3) / 4; d.nwy = H / 4;
  d.scale = 1.0f / sqrtf((float)(C / heads));
  return true;
}

// number of table-gradient slabs the 4x4 backward writes (one per wave task = group of four windows), or -1
int winattn_mfma16_slabs(int N, int C, int H, int W, int heads, int ws, int shift) {
  W4Desc d{};
  W4Fn f, b;
  size_t lf, lb;
  if (!w4_fill(d, N, C, H, W, heads, ws, shift) || !w4_pick(C / heads, f, b, lf, lb) || lb > 160 * 1024) return -1;
  return N * d.nwy * d.ngx;
}

int winattn_mfma16_fwd(const float* qkv, const float* table, float* out, int N, int C, int H, int W, int heads, int ws,
                       int shift, hipStream_t stream) {
  W4Desc d{};
  W4Fn f, b;
  size_t lf, lb;
  if (!w4_fill(d, N, C, H, W, heads, ws, shift) || !w4_pick(C / heads, f, b, lf, lb) || lf > 160 * 1024) return -1;
  d.qkv = qkv; d.table = table; d.out = out;
  if (lf > 64 * 1024 && !ensure_max_lds(reinterpret_cast<const void*>(f))) return ICM_ERR_LAUNCH;
  const int nwt = N * d.nwy * d.ngx * heads;
  hipLaunchKernelGGL(f, dim3((nwt + 3) / 4), dim3(256), lf, stream, d);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}
int winattn_mfma16_bwd(const float* qkv, const float* table, const float* dout, float* dqkv, float* dtable_ws, int N,
                       int C, int H, int W, int heads, int ws, int shift, hipStream_t stream) {
  W4Desc d{};
  W4Fn f, b;
  size_t lf, lb;
  if (!w4_fill(d, N, C, H, W, heads, ws, shift) || !w4_pick(C / heads, f, b, lf, lb) || lb > 160 * 1024) return -1;
  d.qkv = qkv; d.table = table; d.dout = dout; d.dqkv = dqkv; d.dtable_ws = dtable_ws;
  if (lb > 64 * 1024 && !ensure_max_lds(reinterpret_cast<const void*>(b))) return ICM_ERR_LAUNCH;
  const int nwt = N * d.nwy * d.ngx * heads;
  hipLaunchKernelGGL(b, dim3((nwt + 1) / 2), dim3(128), lb, stream, d);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}

}  // namespace icm
