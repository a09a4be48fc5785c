// Shared by the convolution kernels (conv_igemm.hip, conv_1x1.hip): launch descriptor and the fused epilogues.
#pragma once
#include "icm_common.h"

namespace icm {

#define ICM_MAX_TAPS 32
#define ICM_MAX_GROUPS 12   /* problems of identical geometry per launch (blockIdx.y): the independent slice chains */

struct ConvPtrs {
  const float* x;
  const float* wp;
  const float* bias;
  float* y;
  const float* res;
  const float* aux;
  const float* aux2;
  float* y2;
};

struct ConvDesc {
  ConvPtrs g[ICM_MAX_GROUPS];
  long long y_bs, res_bs, aux_bs, aux2_bs, y2_bs;
  PatchGeom pg;                // input tensor + LDS patch layout
  int Cout, OHf, OWf;
  int OHv, OWv;
  int out_sy, out_oy, out_sx, out_ox;
  int iy0, ix0;
  int ntaps;
  int lgTW, lgTH, lgTI;
  int tiles_x, tiles_y, tiles_n;
  int ncot, nchunks8, ckm, ncb;
  int epi, accum, ps2;
  int tapoff[ICM_MAX_TAPS];   // dword entries: read with s_load (a 16-bit entry forces a VMEM load + vmcnt(0))
};

// Epilogue of one 32x32 accumulator tile.  The fused-neighbour kind is a template parameter so that the 16 rows
// form ONE basic block: all operand loads (bias, residual, aux, old value) are issued back to back and waited for
// once, instead of a load -> wait -> store chain per element.
template <int EPI, int half>
__device__ __forceinline__ void store_half_e(const ConvDesc& d, const ConvPtrs& P, const f32x16 acc, int cot, int h,
                                             int n, int oy, int ox, bool pvalid) {
  const int plane = d.OHf * d.OWf;
  constexpr bool kRes = EPI == ICM_EPI_RES || EPI == ICM_EPI_RES_GELU || EPI == ICM_EPI_RES_MUL_DGELU;
  constexpr bool kAux = EPI == ICM_EPI_GDN || EPI == ICM_EPI_IGDN || EPI == ICM_EPI_MUL_DGELU ||
                        EPI == ICM_EPI_AXPY2 || EPI == ICM_EPI_LRP || EPI == ICM_EPI_RES_MUL_DGELU;
  float* yb = P.y + n * d.y_bs;
  const float* resb = kRes ? P.res + n * d.res_bs : nullptr;
  const float* auxb = kAux ? P.aux + n * d.aux_bs : nullptr;
  const float* aux2b = (EPI == ICM_EPI_AXPY2) ? P.aux2 + n * d.aux2_bs : nullptr;
  float* y2b = P.y2 ? P.y2 + n * d.y2_bs : nullptr;
  const bool has_bias = P.bias != nullptr;
  // rows in two halves of 8: bounds the live registers of the load batch (the kernel's VGPR budget sets occupancy)
  {
    int off[8];
    bool ok[8];
    float bv[8], rv[8], av[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int r = half * 8 + q;
      const int co = cot * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      ok[q] = pvalid && co < d.Cout;
      off[q] = d.ps2 ? (co >> 2) * plane + (oy * 2 + ((co >> 1) & 1)) * d.OWf + ox * 2 + (co & 1)
                     : co * plane + oy * d.OWf + ox;
      bv[q] = (has_bias && ok[q]) ? P.bias[co] : 0.0f;
      if constexpr (kRes) rv[q] = ok[q] ? resb[off[q]] : 0.0f;
      if constexpr (kAux) av[q] = ok[q] ? auxb[off[q]] : 0.0f;
      if constexpr (EPI == ICM_EPI_AXPY2) rv[q] = ok[q] ? aux2b[off[q]] : 0.0f;
    }
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      v[q] = acc[half * 8 + q] + bv[q];
      if constexpr (EPI == ICM_EPI_RES) v[q] += rv[q];
      if constexpr (EPI == ICM_EPI_RES_GELU) v[q] += gelu_f(rv[q]);
      if constexpr (EPI == ICM_EPI_GDN || EPI == ICM_EPI_IGDN) {
        if (y2b && ok[q]) y2b[off[q]] = v[q];
        v[q] = av[q] * (EPI == ICM_EPI_GDN ? rsqrtf(v[q]) : sqrtf(v[q]));
      }
      if constexpr (EPI == ICM_EPI_MUL_DGELU) v[q] *= dgelu_f(av[q]);
      if constexpr (EPI == ICM_EPI_RES_MUL_DGELU) v[q] = (v[q] + rv[q]) * dgelu_f(av[q]);
      if constexpr (EPI == ICM_EPI_AXPY2) v[q] = rv[q] + 2.0f * av[q] * v[q];
      if constexpr (EPI == ICM_EPI_LRP) {
        const float t = tanhf(v[q]);
        if (y2b && ok[q]) y2b[off[q]] = t;
        v[q] = av[q] + 0.5f * t;
      }
    }
    if (d.accum) {   // accumulation (gradients; partial first-layer sums of the slice chains): one more batched read
      float old[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) old[q] = ok[q] ? yb[off[q]] : 0.0f;
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] += old[q];
    }
    if constexpr (EPI == ICM_EPI_NONE || EPI == ICM_EPI_RES || EPI == ICM_EPI_RES_GELU) {
      // materialised activation for the consumers of this pre-activation (forward only): y2 = gelu(y), of the value
      // AFTER an accumulation (the launch that completes a partial sum materialises it).  y2 == y (inference: nobody
      // reads the pre-activation again): only gelu(y) is stored.
      if (y2b) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float gv = gelu_f(v[q]);
          if (y2b == yb) v[q] = gv;
          else if (ok[q]) y2b[off[q]] = gv;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (ok[q]) yb[off[q]] = v[q];
  }
}
template <int EPI>
__device__ __forceinline__ void store_tile_e(const ConvDesc& d, const ConvPtrs& P, const f32x16 acc, int cot, int h,
                                             int n, int oy, int ox, bool pvalid) {
  store_half_e<EPI, 0>(d, P, acc, cot, h, n, oy, ox, pvalid);
  store_half_e<EPI, 1>(d, P, acc, cot, h, n, oy, ox, pvalid);
}

// Pipelined form: epilogue of half a 32x32 accumulator tile (8 rows of one lane's column), in two phases so that the callers can
// software-pipeline it: epi_load issues every operand load of the half tile (bias, residual, aux, old value for
// gradient accumulation) back to back; epi_finish consumes them and stores.  The callers issue the loads of half
// tile i+1 BEFORE finishing half tile i: one exposed memory round trip per wave instead of one per half tile (a
// 192 x 32 strip with a residual operand used to pay 12 dependent load -> use -> store round trips).  The fused-
// neighbour kind is a template parameter: each phase is one basic block.
struct EpiRegs {
  int n;
  int off[8];
  bool ok[8];
  float bv[8], rv[8], av[8], old[8];
};

template <int EPI>
__device__ __forceinline__ void epi_load(const ConvDesc& d, const ConvPtrs& P, int cot, int half, int h, int n, int oy,
                                         int ox, bool pvalid, EpiRegs& R) {
  const int plane = d.OHf * d.OWf;
  constexpr bool kRes = EPI == ICM_EPI_RES || EPI == ICM_EPI_RES_GELU || EPI == ICM_EPI_RES_MUL_DGELU;
  constexpr bool kAux = EPI == ICM_EPI_GDN || EPI == ICM_EPI_IGDN || EPI == ICM_EPI_MUL_DGELU ||
                        EPI == ICM_EPI_AXPY2 || EPI == ICM_EPI_LRP || EPI == ICM_EPI_RES_MUL_DGELU;
  const float* yb = P.y + n * d.y_bs;
  const float* resb = kRes ? P.res + n * d.res_bs : nullptr;
  const float* auxb = kAux ? P.aux + n * d.aux_bs : nullptr;
  const float* aux2b = (EPI == ICM_EPI_AXPY2) ? P.aux2 + n * d.aux2_bs : nullptr;
  const bool has_bias = P.bias != nullptr;
  const bool tile_ok = pvalid && cot < d.ncot;
  R.n = n;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int r = half * 8 + q;
    const int co = cot * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    R.ok[q] = tile_ok && co < d.Cout;
    R.off[q] = d.ps2 ? (co >> 2) * plane + (oy * 2 + ((co >> 1) & 1)) * d.OWf + ox * 2 + (co & 1)
                     : co * plane + oy * d.OWf + ox;
    R.bv[q] = (has_bias && R.ok[q]) ? P.bias[co] : 0.0f;
    if constexpr (kRes) R.rv[q] = R.ok[q] ? resb[R.off[q]] : 0.0f;
    if constexpr (kAux) R.av[q] = R.ok[q] ? auxb[R.off[q]] : 0.0f;
    if constexpr (EPI == ICM_EPI_AXPY2) R.rv[q] = R.ok[q] ? aux2b[R.off[q]] : 0.0f;
  }
  if (d.accum) {   // gradient accumulation: one more batched read of the destination
#pragma unroll
    for (int q = 0; q < 8; ++q) R.old[q] = R.ok[q] ? yb[R.off[q]] : 0.0f;
  }
}

template <int EPI>
__device__ __forceinline__ void epi_finish(const ConvDesc& d, const ConvPtrs& P, const float (&a8)[8], const EpiRegs& R) {
  float* yb = P.y + R.n * d.y_bs;
  float* y2b = P.y2 ? P.y2 + R.n * d.y2_bs : nullptr;
  float v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    v[q] = a8[q] + R.bv[q];
    if constexpr (EPI == ICM_EPI_RES) v[q] += R.rv[q];
    if constexpr (EPI == ICM_EPI_RES_GELU) v[q] += gelu_f(R.rv[q]);
    if constexpr (EPI == ICM_EPI_GDN || EPI == ICM_EPI_IGDN) {
      if (y2b && R.ok[q]) y2b[R.off[q]] = v[q];
      v[q] = R.av[q] * (EPI == ICM_EPI_GDN ? rsqrtf(v[q]) : sqrtf(v[q]));
    }
    if constexpr (EPI == ICM_EPI_MUL_DGELU) v[q] *= dgelu_f(R.av[q]);
    if constexpr (EPI == ICM_EPI_RES_MUL_DGELU) v[q] = (v[q] + R.rv[q]) * dgelu_f(R.av[q]);
    if constexpr (EPI == ICM_EPI_AXPY2) v[q] = R.rv[q] + 2.0f * R.av[q] * v[q];
    if constexpr (EPI == ICM_EPI_LRP) {
      const float t = tanhf(v[q]);
      if (y2b && R.ok[q]) y2b[R.off[q]] = t;
      v[q] = R.av[q] + 0.5f * t;
    }
  }
  if (d.accum) {
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] += R.old[q];
  }
  if constexpr (EPI == ICM_EPI_NONE || EPI == ICM_EPI_RES || EPI == ICM_EPI_RES_GELU) {
    if (y2b) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float gv = gelu_f(v[q]);
        if (y2b == yb) v[q] = gv;   // y2 == y: only the activated value is stored (see store_half_e)
        else if (R.ok[q]) y2b[R.off[q]] = gv;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 8; ++q)
    if (R.ok[q]) yb[R.off[q]] = v[q];
}

// All TCO x TPX accumulator tiles of one wave, pipelined over half tiles.  pn / poy / pox / pv: image, output row,
// output column and validity of this lane's pixel in pixel tile tp (a flat 1x1 caller passes row 0, column = offset
// inside the plane).
template <int EPI, int TCO, int TPX, bool PIPE>
__device__ __forceinline__ void epilogue_tiles(const ConvDesc& d, const ConvPtrs& P, const f32x16 (&acc)[TCO][TPX], int cot0,
                                               int h, const int (&pn)[TPX], const int (&poy)[TPX], const int (&pox)[TPX],
                                               const bool (&pv)[TPX]) {
  if constexpr (!PIPE) {
#pragma unroll
    for (int tp = 0; tp < TPX; ++tp)
#pragma unroll
      for (int a = 0; a < TCO; ++a) {
        if (cot0 + a < d.ncot) store_tile_e<EPI>(d, P, acc[a][tp], cot0 + a, h, pn[tp], poy[tp], pox[tp], pv[tp]);
        __builtin_amdgcn_sched_barrier(0);
      }
    return;
  }
  constexpr int S = TPX * TCO * 2;
  EpiRegs R[2];
  epi_load<EPI>(d, P, cot0, 0, h, pn[0], poy[0], pox[0], pv[0], R[0]);
#pragma unroll
  for (int i = 0; i < S; ++i) {
    const int half = i & 1, a = (i >> 1) % TCO, tp = (i >> 1) / TCO;
    if (i + 1 < S) {
      const int i2 = i + 1;
      const int half2 = i2 & 1, a2 = (i2 >> 1) % TCO, tp2 = (i2 >> 1) / TCO;
      epi_load<EPI>(d, P, cot0 + a2, half2, h, pn[tp2], poy[tp2], pox[tp2], pv[tp2], R[i2 & 1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    float a8[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) a8[q] = acc[a][tp][half * 8 + q];
    epi_finish<EPI>(d, P, a8, R[i & 1]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// the epilogue kind is dispatched ONCE (uniform branch) around the fully unrolled tile loops.  PIPE2: pipeline the
// kinds that read two full-size operands from HBM (AXPY2: x and the dn map of the GDN backward, 2 x 201 MB at
// 128x128 -- measured 0.76 -> 0.54 ms); for the single-operand kinds, whose operand is L2-warm, the extra registers
// of the pipelined form cost more than the round trips (measured), so they keep the batched half-tile form.
template <int TCO, int TPX, bool PIPE2 = false>
__device__ __forceinline__ void epilogue_dispatch(const ConvDesc& d, const ConvPtrs& P, const f32x16 (&acc)[TCO][TPX],
                                                  int cot0, int h, const int (&pn)[TPX], const int (&poy)[TPX],
                                                  const int (&pox)[TPX], const bool (&pv)[TPX]) {
  switch (d.epi) {
    case ICM_EPI_RES: epilogue_tiles<ICM_EPI_RES, TCO, TPX, false>(d, P, acc, cot0, h, pn, poy, pox, pv); break;
    case ICM_EPI_RES_GELU: epilogue_tiles<ICM_EPI_RES_GELU, TCO, TPX, false>(d, P, acc, cot0, h, pn, poy, pox, pv); break;
    case ICM_EPI_GDN: epilogue_tiles<ICM_EPI_GDN, TCO, TPX, false>(d, P, acc, cot0, h, pn, poy, pox, pv); break;
    case ICM_EPI_IGDN: epilogue_tiles<ICM_EPI_IGDN, TCO, TPX, false>(d, P, acc, cot0, h, pn, poy, pox, pv); break;
    case ICM_EPI_MUL_DGELU: epilogue_tiles<ICM_EPI_MUL_DGELU, TCO, TPX, false>(d, P, acc, cot0, h, pn, poy, pox, pv); break;
    case ICM_EPI_AXPY2: epilogue_tiles<ICM_EPI_AXPY2, TCO, TPX, PIPE2>(d, P, acc, cot0, h, pn, poy, pox, pv); break;
    case ICM_EPI_LRP: epilogue_tiles<ICM_EPI_LRP, TCO, TPX, false>(d, P, acc, cot0, h, pn, poy, pox, pv); break;
    case ICM_EPI_RES_MUL_DGELU:
      epilogue_tiles<ICM_EPI_RES_MUL_DGELU, TCO, TPX, false>(d, P, acc, cot0, h, pn, poy, pox, pv);
      break;
    default: epilogue_tiles<ICM_EPI_NONE, TCO, TPX, false>(d, P, acc, cot0, h, pn, poy, pox, pv); break;
  }
}

// latency-bound problems (conv_ks8.hip): eight MFMA waves split K over one block of four tiles (tco = 2: 64 co x 64 px;
// tco = 1: 32 co x 128 px); d must describe a stride-1-sampling, activation-free, linear-patch launch
int launch_conv_ks8(const ConvDesc& d, int tco, long long nblk, int ngroups, size_t lds_bytes, hipStream_t stream);

// Winograd F(2x2, 3x3) path (conv_wino.hip): 3x3 stride-1 pad-1 launches whose weights were packed with wino != 0
bool wino_supported(const icm_conv_args& a);
int run_conv_wino(const icm_conv_args* arr, int ngroups, hipStream_t stream);
long long wino_transform_floats(const icm_conv_args& a);
int run_wino_transform(const icm_conv_args* arr, int ngroups, hipStream_t stream);

// pointwise path (conv_1x1.hip): ICM_OK after launching, -1 when the launch should take the LDS-staged kernel
int run_conv1x1(const icm_conv_args* arr, int ngroups, long long wp_off, int force_mode, hipStream_t stream);

}  // namespace icm
