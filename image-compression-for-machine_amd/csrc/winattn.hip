// Window multi-head self-attention core (forward + backward) on NCHW tensors, gfx950.
//
// Replaces WinBasedAttention / WindowAttention between the qkv and proj Linears
// (layers/win_attention.py:84-115,153-207).  The reference materialises six permuted copies
// (NCHW->NHWC, roll, window_partition, qkv reshape/permute, window_reverse, roll back, NHWC->NCHW);
// here the cyclic shift, window partition, head split, relative-position-bias gather and the 0/-100
// shift mask are pure address arithmetic on the NCHW qkv tensor produced by the 1x1-conv GEMM.
//
// One workgroup = one window of one image; each wave walks heads.  Lane = query token.  K and V of the
// head live in LDS (broadcast reads), the score row lives in a per-wave LDS matrix with odd row stride
// (conflict-free row and column sweeps), so the backward pass gets P^T / dS^T for dV, dK without any
// cross-lane shuffles.  Relative-position-bias gradients are reduced per wave in LDS (all lanes of one
// instruction hit distinct table entries); every (window, head) writes its partial table to a workspace slab
// and two small kernels add the slabs in window order: bitwise reproducible, no float atomics.
#include <algorithm>
#include "icm_common.h"

namespace icm {

// matrix-core kernels for 8x8 windows (winattn_mfma.hip): ICM_OK = launched, -1 = geometry not covered
int winattn_mfma_fwd(const float* qkv, const float* table, float* out, int N, int C, int H, int W, int heads, int ws,
                     int shift, hipStream_t stream);
int winattn_mfma_bwd(const float* qkv, const float* table, const float* dout, float* dqkv, float* dtable_ws, int N, int C,
                     int H, int W, int heads, int ws, int shift, hipStream_t stream);
int winattn_mfma16_fwd(const float* qkv, const float* table, float* out, int N, int C, int H, int W, int heads, int ws,
                       int shift, hipStream_t stream);
int winattn_mfma16_bwd(const float* qkv, const float* table, const float* dout, float* dqkv, float* dtable_ws, int N,
                       int C, int H, int W, int heads, int ws, int shift, hipStream_t stream);
int winattn_mfma16_slabs(int N, int C, int H, int W, int heads, int ws, int shift);
static int g_force_valu = 0;   // test hook: 1 = always take the generic (VALU) kernels below

struct WaDesc {
  const float* qkv;
  const float* table;
  float* out;          // fwd
  const float* dout;   // bwd
  float* dqkv;         // bwd
  float* dtable_ws;    // bwd: per-(window, head) partial tables [nwin][heads][(2ws-1)^2]
  int N, C, H, W, heads, ws, shift, hd, T, nwx, nwy, G;
  float scale;
};

__device__ __forceinline__ int region(int s, int L, int ws, int shift) {
  return s < L - ws ? 0 : (s < L - shift ? 1 : 2);
}

// Common per-token geometry
struct Tok {
  int pix;   // oy*W + ox in the original (un-shifted) image
  int lab;   // shift-mask region label
  int r, c;  // row / col inside the window
};
__device__ __forceinline__ Tok token(const WaDesc& d, int wy, int wx, int j) {
  Tok t;
  t.r = j / d.ws;
  t.c = j - t.r * d.ws;
  const int sy = wy * d.ws + t.r, sx = wx * d.ws + t.c;
  int oy = sy + d.shift, ox = sx + d.shift;
  if (oy >= d.H) oy -= d.H;
  if (ox >= d.W) ox -= d.W;
  t.pix = oy * d.W + ox;
  t.lab = d.shift > 0 ? region(sy, d.H, d.ws, d.shift) * 3 + region(sx, d.W, d.ws, d.shift) : 0;
  return t;
}

// s = sum_d a[d] * row[d]; row is an LDS row shared by all lanes (broadcast): 16-byte reads when HD % 4 == 0
template <int HD>
__device__ __forceinline__ float dot_row(const float (&a)[HD], const float* __restrict__ row) {
  float s = 0.0f;
  if constexpr (HD % 4 == 0) {
    const f32x4* r4 = reinterpret_cast<const f32x4*>(row);
#pragma unroll
    for (int q = 0; q < HD / 4; ++q) {
      const f32x4 v = r4[q];
      s += a[4 * q] * v[0];
      s += a[4 * q + 1] * v[1];
      s += a[4 * q + 2] * v[2];
      s += a[4 * q + 3] * v[3];
    }
  } else {
#pragma unroll
    for (int dd = 0; dd < HD; ++dd) s += a[dd] * row[dd];
  }
  return s;
}
// o[d] += p * row[d]
template <int HD>
__device__ __forceinline__ void axpy_row(float (&o)[HD], float p, const float* __restrict__ row) {
  if constexpr (HD % 4 == 0) {
    const f32x4* r4 = reinterpret_cast<const f32x4*>(row);
#pragma unroll
    for (int q = 0; q < HD / 4; ++q) {
      const f32x4 v = r4[q];
      o[4 * q] += p * v[0];
      o[4 * q + 1] += p * v[1];
      o[4 * q + 2] += p * v[2];
      o[4 * q + 3] += p * v[3];
    }
  } else {
#pragma unroll
    for (int dd = 0; dd < HD; ++dd) o[dd] += p * row[dd];
  }
}

template <int HD>
__global__ __launch_bounds__(256) void winattn_fwd_kernel(const WaDesc d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane64 = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int T = d.T, TS = T + 1;
  // small windows (T = 16): G = 64 / T heads share a wave, lane = (head group, token)
  const int G = d.G, grp = lane64 / T, lane = lane64 - grp * T;
  float* Ksh = smem + (wave * G + grp) * (2 * T * HD + T * TS);
  float* Vsh = Ksh + T * HD;
  float* Ssh = Vsh + T * HD;
  // XCD-aware window order: hardware id b runs on XCD b % 8; a contiguous run of windows per XCD lets horizontally
  // adjacent windows (which share every 128-B line of a 4- or 8-pixel-wide row segment) meet in one L2
  int bid;
  {
    const int nb = gridDim.x, hb = blockIdx.x;
    const int xcd = hb & 7, q = hb >> 3;
    bid = xcd * (nb >> 3) + min(xcd, nb & 7) + q;
  }
  const int wx = bid % d.nwx; bid /= d.nwx;
  const int wy = bid % d.nwy;
  const int n = bid / d.nwy;
  const long long HW = (long long)d.H * d.W;
  const float* base = d.qkv + (long long)n * 3 * d.C * HW;
  const Tok me = token(d, wy, wx, lane);
  const int tw = 2 * d.ws - 1;

  for (int head0 = wave * G; head0 < d.heads; head0 += nwaves * G) {
    const int head = min(head0 + grp, d.heads - 1);
    const bool active = grp < G && head0 + grp < d.heads;
    const float* qp = base + (long long)(head * HD) * HW;
    const float* kp = base + (long long)(d.C + head * HD) * HW;
    const float* vp = base + (long long)(2 * d.C + head * HD) * HW;
    float q[HD];
    if (active) {
#pragma unroll
      for (int dd = 0; dd < HD; ++dd) {
        q[dd] = qp[dd * HW + me.pix] * d.scale;
        Ksh[lane * HD + dd] = kp[dd * HW + me.pix];
        Vsh[lane * HD + dd] = vp[dd * HW + me.pix];
      }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    float m = -3.0e38f;
    if (active) {
      for (int j = 0; j < T; ++j) {
        float s = dot_row<HD>(q, Ksh + j * HD);
        const Tok tj = token(d, wy, wx, j);
        const int idx = (me.r - tj.r + d.ws - 1) * tw + (me.c - tj.c + d.ws - 1);
        s += d.table[idx * d.heads + head];
        if (d.shift > 0 && tj.lab != me.lab) s += -100.0f;
        Ssh[lane * TS + j] = s;
        m = fmaxf(m, s);
      }
      float l = 0.0f;
      for (int j = 0; j < T; ++j) {
        const float p = expf(Ssh[lane * TS + j] - m);
        Ssh[lane * TS + j] = p;
        l += p;
      }
      const float inv = 1.0f / l;
      float o[HD];
#pragma unroll
      for (int dd = 0; dd < HD; ++dd) o[dd] = 0.0f;
      for (int j = 0; j < T; ++j) {
        const float p = Ssh[lane * TS + j] * inv;
        axpy_row<HD>(o, p, Vsh + j * HD);
      }
      float* op = d.out + ((long long)n * d.C + head * HD) * HW;
#pragma unroll
      for (int dd = 0; dd < HD; ++dd) op[dd * HW + me.pix] = o[dd];
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <int HD>
__global__ __launch_bounds__(128) void winattn_bwd_kernel(const WaDesc d) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane64 = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int T = d.T, TS = T + 1;
  const int tw = 2 * d.ws - 1, ntab = tw * tw;
  // LDS per head: K | V | dO | S.  Two regions are recycled so that two workgroups fit one CU (35 KB per head for
  // 8x8 windows): dO is dead after dV -> it becomes the per-head table-gradient accumulator; V is dead after
  // dP / dq -> the scaled q (kept in registers until then) is parked there for dK.
  const int per_wave = (3 * T * HD + T * TS + 3) & ~3;   // keep every slab 16-B aligned (b128 reads)
  const int G = d.G, grp = lane64 / T, lane = lane64 - grp * T;     // G heads per wave for small windows
  float* Ksh = smem + (wave * G + grp) * per_wave;
  float* Vsh = Ksh + T * HD;
  float* Gsh = Vsh + T * HD;   // dO
  float* Ssh = Gsh + T * HD;   // P then dS
  float* Qsh = Vsh;            // scaled q, written after the dq phase
  float* Bsh = Gsh;            // per-head table gradient, zeroed after the dV phase
  // XCD-aware window order: hardware id b runs on XCD b % 8; a contiguous run of windows per XCD lets horizontally
  // adjacent windows (which share every 128-B line of a 4- or 8-pixel-wide row segment) meet in one L2
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
  const float* base = d.qkv + (long long)n * 3 * d.C * HW;
  float* dbase = d.dqkv + (long long)n * 3 * d.C * HW;
  const Tok me = token(d, wy, wx, lane);

  for (int head0 = wave * G; head0 < d.heads; head0 += nwaves * G) {
    const int head = min(head0 + grp, d.heads - 1);
    const bool active = grp < G && head0 + grp < d.heads;
    const float* qp = base + (long long)(head * HD) * HW;
    const float* kp = base + (long long)(d.C + head * HD) * HW;
    const float* vp = base + (long long)(2 * d.C + head * HD) * HW;
    const float* gp = d.dout + ((long long)n * d.C + head * HD) * HW;
    float q[HD], go[HD];
    if (active) {
#pragma unroll
      for (int dd = 0; dd < HD; ++dd) {
        q[dd] = qp[dd * HW + me.pix] * d.scale;
        go[dd] = gp[dd * HW + me.pix];
        Gsh[lane * HD + dd] = go[dd];
        Ksh[lane * HD + dd] = kp[dd * HW + me.pix];
        Vsh[lane * HD + dd] = vp[dd * HW + me.pix];
      }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    // ---- recompute P (row per lane)
    float inv = 0.0f;
    if (active) {
      float m = -3.0e38f;
      for (int j = 0; j < T; ++j) {
        float s = dot_row<HD>(q, Ksh + j * HD);
        const Tok tj = token(d, wy, wx, j);
        const int idx = (me.r - tj.r + d.ws - 1) * tw + (me.c - tj.c + d.ws - 1);
        s += d.table[idx * d.heads + head];
        if (d.shift > 0 && tj.lab != me.lab) s += -100.0f;
        Ssh[lane * TS + j] = s;
        m = fmaxf(m, s);
      }
      float l = 0.0f;
      for (int j = 0; j < T; ++j) {
        const float p = expf(Ssh[lane * TS + j] - m);
        Ssh[lane * TS + j] = p;
        l += p;
      }
      inv = 1.0f / l;
      for (int j = 0; j < T; ++j) Ssh[lane * TS + j] *= inv;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    // ---- dV[j][:] = sum_i P[i][j] dO[i][:]   (lane = j, column sweep of P)
    if (active) {
      float acc[HD];
#pragma unroll
      for (int dd = 0; dd < HD; ++dd) acc[dd] = 0.0f;
      for (int i = 0; i < T; ++i) {
        const float p = Ssh[i * TS + lane];
        axpy_row<HD>(acc, p, Gsh + i * HD);
      }
      float* dv = dbase + (long long)(2 * d.C + head * HD) * HW;
#pragma unroll
      for (int dd = 0; dd < HD; ++dd) dv[dd * HW + me.pix] = acc[dd];
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (grp < G)   // dO rows are dead: the region now accumulates the table gradient
      for (int i = lane; i < ntab; i += T) Bsh[i] = 0.0f;
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    // ---- dP, delta, dS (row per lane), dq, table gradient
    if (active) {
      float delta = 0.0f;
      for (int j = 0; j < T; ++j) {
        const float dp = dot_row<HD>(go, Vsh + j * HD);
        delta += Ssh[lane * TS + j] * dp;
      }
      float dq[HD];
#pragma unroll
      for (int dd = 0; dd < HD; ++dd) dq[dd] = 0.0f;
      for (int j = 0; j < T; ++j) {
        const float dp = dot_row<HD>(go, Vsh + j * HD);
        const float ds = Ssh[lane * TS + j] * (dp - delta);
        Ssh[lane * TS + j] = ds;
        axpy_row<HD>(dq, ds, Ksh + j * HD);
        const Tok tj = token(d, wy, wx, j);
        const int idx = (me.r - tj.r + d.ws - 1) * tw + (me.c - tj.c + d.ws - 1);
        Bsh[idx] += ds;  // distinct idx across the active lanes of this instruction
      }
      float* dqp = dbase + (long long)(head * HD) * HW;
#pragma unroll
      for (int dd = 0; dd < HD; ++dd) dqp[dd * HW + me.pix] = dq[dd] * d.scale;
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    if (active) {   // V rows are dead: park the scaled q there for the dK sweep
#pragma unroll
      for (int dd = 0; dd < HD; ++dd) Qsh[lane * HD + dd] = q[dd];
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    // ---- dK[j][:] = sum_i dS[i][j] q[i][:]   (lane = j)
    if (active) {
      float acc[HD];
#pragma unroll
      for (int dd = 0; dd < HD; ++dd) acc[dd] = 0.0f;
      for (int i = 0; i < T; ++i) {
        const float ds = Ssh[i * TS + lane];
        axpy_row<HD>(acc, ds, Qsh + i * HD);
      }
      float* dk = dbase + (long long)(d.C + head * HD) * HW;
#pragma unroll
      for (int dd = 0; dd < HD; ++dd) dk[dd * HW + me.pix] = acc[dd];
    }
    if (active) {
      float* slab = d.dtable_ws + ((long long)win * d.heads + head) * ntab;
      for (int i = lane; i < ntab; i += T) slab[i] = Bsh[i];
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
  }
}

// Table-gradient reduction, stage 1: part[s][e] = sum over the windows of chunk s of slab[w][e], e = head*ntab + i.
// Workgroup = 64 entries x 4 window lanes; window lanes are combined in lane order through LDS.
__global__ __launch_bounds__(256) void dtable_reduce1_kernel(const float* __restrict__ slabs, float* __restrict__ part,
                                                             int E, int nwin, int chunk) {
  __shared__ float red[4][64];
  const int e = blockIdx.x * 64 + (threadIdx.x & 63), wl = threadIdx.x >> 6, s = blockIdx.y;
  const int w0 = s * chunk, w1 = min(nwin, w0 + chunk);
  float acc = 0.0f;
  if (e < E)
    for (int w = w0 + wl; w < w1; w += 4) acc += slabs[(long long)w * E + e];
  red[wl][threadIdx.x & 63] = acc;
  __syncthreads();
  if (wl == 0 && e < E)
    part[(long long)s * E + e] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// stage 2: dtable[i*heads + head] (+)= sum_s part[s][head*ntab + i]
__global__ void dtable_reduce2_kernel(const float* __restrict__ part, float* __restrict__ dtable, int E, int S,
                                      int ntab, int heads, int accum) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  float acc = 0.0f;
  for (int s = 0; s < S; ++s) acc += part[(long long)s * E + e];
  const int head = e / ntab, i = e - head * ntab;
  float* o = dtable + i * heads + head;
  *o = accum ? *o + acc : acc;
}

typedef void (*WaFn)(const WaDesc);
template <int HD> struct WaK {
  static WaFn fwd() { return winattn_fwd_kernel<HD>; }
  static WaFn bwd() { return winattn_bwd_kernel<HD>; }
};
static bool pick(int hd, WaFn& f, WaFn& b) {
  switch (hd) {
#define C_(n) case n: f = WaK<n>::fwd(); b = WaK<n>::bwd(); return true;
    C_(8) C_(10) C_(16) C_(24) C_(32) C_(40) C_(48)
#undef C_
    default: return false;
  }
}

static int fill_desc(WaDesc& d, int N, int C, int H, int W, int heads, int ws, int shift) {
  if (N <= 0 || C <= 0 || heads <= 0 || C % heads != 0 || ws <= 0) return ICM_ERR_ARG;
  if (shift < 0 || shift >= ws) return ICM_ERR_ARG;            // assert at win_attention.py:144
  if (H % ws != 0 || W % ws != 0) return ICM_ERR_ARG;          // view() would raise in window_partition
  if (ws * ws > 64) return ICM_ERR_UNSUPPORTED;
  d.N = N; d.C = C; d.H = H; d.W = W; d.heads = heads; d.ws = ws; d.shift = shift; d.hd = C / heads;
  d.T = ws * ws; d.nwx = W / ws; d.nwy = H / ws;
  d.G = 64 / d.T;   // heads per wave (1 for 8x8 windows, 4 for 4x4)
  d.scale = 1.0f / sqrtf((float)d.hd);
  return ICM_OK;
}

}  // namespace icm

using namespace icm;
extern "C" {

void icm_debug_force_winattn_valu(int on) { icm::g_force_valu = on ? 1 : 0; }

int icm_winattn_fwd(const float* qkv, const float* table, float* out, int N, int C, int H, int W, int heads, int ws,
                    int shift, void* stream) {
  if (!qkv || !table || !out) return ICM_ERR_ARG;
  WaDesc d{};
  int rc = fill_desc(d, N, C, H, W, heads, ws, shift);
  if (rc) return rc;
  d.qkv = qkv; d.table = table; d.out = out;
  if (!g_force_valu) {
    int rm = winattn_mfma_fwd(qkv, table, out, N, C, H, W, heads, ws, shift, (hipStream_t)stream);
    if (rm < 0) rm = winattn_mfma16_fwd(qkv, table, out, N, C, H, W, heads, ws, shift, (hipStream_t)stream);
    if (rm >= 0) return rm;
  }
  WaFn f, b;
  if (!pick(d.hd, f, b)) return ICM_ERR_UNSUPPORTED;
  const int waves = std::min(4, (heads + d.G - 1) / d.G);
  const size_t lds = (size_t)waves * d.G * (2 * d.T * d.hd + d.T * (d.T + 1)) * 4;
  if (lds > 160 * 1024) return ICM_ERR_UNSUPPORTED;
  if (lds > 64 * 1024 && !ensure_max_lds(reinterpret_cast<const void*>(f))) return ICM_ERR_LAUNCH;
  hipLaunchKernelGGL(f, dim3(N * d.nwy * d.nwx), dim3(64 * waves), lds, (hipStream_t)stream, d);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}

static void dtable_plan(int N, int H, int W, int heads, int ws, int& nwin, int& E, int& S, int& chunk) {
  const int tw = 2 * ws - 1;
  nwin = N * (H / ws) * (W / ws);
  E = heads * tw * tw;
  S = std::max(1, std::min(64, nwin / 16));
  chunk = (nwin + S - 1) / S;
  S = (nwin + chunk - 1) / chunk;
}

int64_t icm_winattn_bwd_workspace_floats(int N, int C, int H, int W, int heads, int ws) {
  if (N <= 0 || heads <= 0 || ws <= 0 || H % ws || W % ws) return -1;
  int nwin, E, S, chunk;
  dtable_plan(N, H, W, heads, ws, nwin, E, S, chunk);
  return (int64_t)nwin * E + (int64_t)S * E;
}

int icm_winattn_bwd(const float* qkv, const float* table, const float* dout, float* dqkv, float* dtable,
                    int accum_table, float* wsp, int64_t ws_floats, int N, int C, int H, int W, int heads, int ws,
                    int shift, void* stream) {
  if (!qkv || !table || !dout || !dqkv || !dtable || !wsp) return ICM_ERR_ARG;
  WaDesc d{};
  int rc = fill_desc(d, N, C, H, W, heads, ws, shift);
  if (rc) return rc;
  int nwin, E, S, chunk;
  dtable_plan(N, H, W, heads, ws, nwin, E, S, chunk);
  if (ws_floats < (int64_t)nwin * E + (int64_t)S * E) return ICM_ERR_ARG;
  d.qkv = qkv; d.table = table; d.dout = dout; d.dqkv = dqkv; d.dtable_ws = wsp;
  const int tw = 2 * ws - 1;
  int rm = -1;
  if (!g_force_valu) {
    rm = winattn_mfma_bwd(qkv, table, dout, dqkv, wsp, N, C, H, W, heads, ws, shift, (hipStream_t)stream);
    if (rm < 0) {
      // 4x4 windows: one table-gradient slab per wave task (four windows), fewer than the per-window slabs planned for
      const int slabs = winattn_mfma16_slabs(N, C, H, W, heads, ws, shift);
      if (slabs > 0) {
        rm = winattn_mfma16_bwd(qkv, table, dout, dqkv, wsp, N, C, H, W, heads, ws, shift, (hipStream_t)stream);
        if (rm == ICM_OK) {
          nwin = slabs;
          S = std::max(1, std::min(64, nwin / 16));
          chunk = (nwin + S - 1) / S;
          S = (nwin + chunk - 1) / chunk;
        }
      }
    }
  }
  if (rm > 0) return rm;
  if (rm < 0) {
    WaFn f, b;
    if (!pick(d.hd, f, b)) return ICM_ERR_UNSUPPORTED;
    const int waves = std::min(2, (heads + d.G - 1) / d.G);
    if (tw * tw > d.T * d.hd) return ICM_ERR_UNSUPPORTED;   // the table gradient is accumulated in the dO region
    const size_t lds = (size_t)waves * d.G * ((3 * d.T * d.hd + d.T * (d.T + 1) + 3) & ~3) * 4;
    if (lds > 160 * 1024) return ICM_ERR_UNSUPPORTED;
    if (lds > 64 * 1024 && !ensure_max_lds(reinterpret_cast<const void*>(b))) return ICM_ERR_LAUNCH;
    hipLaunchKernelGGL(b, dim3(N * d.nwy * d.nwx), dim3(64 * waves), lds, (hipStream_t)stream, d);
    ICM_CHECK_LAUNCH();
  }
  float* part = wsp + (long long)nwin * E;
  hipLaunchKernelGGL(dtable_reduce1_kernel, dim3((E + 63) / 64, S), dim3(256), 0, (hipStream_t)stream, wsp, part, E,
                     nwin, chunk);
  ICM_CHECK_LAUNCH();
  hipLaunchKernelGGL(dtable_reduce2_kernel, dim3((E + 255) / 256), dim3(256), 0, (hipStream_t)stream, part, dtable, E, S,
                     tw * tw, heads, accum_table);
  ICM_CHECK_LAUNCH();
  return ICM_OK;
}

}  // extern "C"
