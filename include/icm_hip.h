/* icm_hip.h -- C ABI of libicm_hip.so: the MI355X (gfx950) hot path of the `cnn` (WACNN) codec.
 *
 * Drop-in boundary.  The reference has no FFI on this path: every op below
 * replaces a call the reference makes into ATen from Python.  Each entry point cites the
 * reference call site it stands in for (paths relative to the reference tree).
 *
 * Conventions
 *   - all tensors are device pointers to IEEE f32, NCHW, contiguous planes (H*W); the batch
 *     stride of every tensor is explicit (in floats) so channel-slices of a larger buffer
 *     (torch.cat / chunk in compressai/models/cnn.py:157-183) are addressed without copies;
 *   - `stream` is a hipStream_t; kernels are enqueued on it; nothing allocates, frees or
 *     synchronises (graph-capture safe); scratch memory is always a caller-owned workspace;
 *   - every reduction (bias / LayerNorm / table gradients, loss sums, gradient norm, split-K
 *     weight gradients) adds its partial sums in a FIXED order -- no float atomics -- so equal inputs
 *     give bit-identical outputs on every rank of a data-parallel job;
 *   - return value: 0 = ok, ICM_ERR_* otherwise (icm_strerror()).
 */
#ifndef ICM_HIP_H
#define ICM_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ICM_OK 0
#define ICM_ERR_ARG 1      /* bad argument (mirrors the reference's ValueError / assert sites) */
#define ICM_ERR_LAUNCH 2   /* hipLaunch failed */
#define ICM_ERR_UNSUPPORTED 3

/* activation applied to an operand while it is staged into LDS ("virtual" activations: the
 * framework stores pre-activations only; nn.GELU at layers/layers.py:59-63, cnn.py:54-127) */
#define ICM_ACT_NONE 0
#define ICM_ACT_GELU 1     /* exact erf GELU */
#define ICM_ACT_SQUARE 2   /* x*x: GDN's conv2d(x**2, gamma) (layers/gdn.py:68) */

/* epilogues of the implicit-GEMM kernels */
#define ICM_EPI_NONE 0        /* y = acc + bias.  NONE / RES / RES_GELU with y2 != NULL (forward): y2 = gelu(y) as well --
                               * the activation materialised once for every consumer of this pre-activation */
#define ICM_EPI_RES 1         /* y = acc + bias + res              (out += identity, layers.py:69) */
#define ICM_EPI_RES_GELU 2    /* y = acc + bias + gelu(res)        (identity is a virtual GELU output) */
#define ICM_EPI_GDN 3         /* y2 = n = acc + bias; y = aux * rsqrt(n)   (gdn.py:68-75) */
#define ICM_EPI_IGDN 4        /* y2 = n;              y = aux * sqrt(n)    (gdn.py:70-71) */
#define ICM_EPI_MUL_DGELU 5   /* y = acc * gelu'(aux)              (backward through a virtual GELU) */
#define ICM_EPI_AXPY2 6       /* y = aux2 + 2*aux*acc              (GDN backward dx) */
#define ICM_EPI_LRP 7         /* y = aux + 0.5*tanh(acc + bias); y2 = tanh(...)  (cnn.py:175-178) */
#define ICM_EPI_RES_MUL_DGELU 8 /* y = (acc + res) * gelu'(aux)   (dgrad into a ResidualUnit input: conv path + identity path) */

const char* icm_strerror(int code);
int icm_version(void);

/* ---- implicit-GEMM convolution family (f32 MFMA) ------------------------------------------
 * Replaces nn.Conv2d / nn.ConvTranspose2d / nn.Linear forward and backward:
 *   models/utils.py:114-132 (conv, deconv), layers/layers.py:29-43 (conv3x3, subpel_conv3x3, conv1x1),
 *   layers/gdn.py:68 (the CxC contraction), layers/win_attention.py:77-79,91,112 (qkv / proj Linear).
 *
 * transposed = 0  "gather":  y[n,o,p]        = sum_{i,t} Wg[o][i][t] * act(x[n,i,p*stride - pad + t])
 * transposed = 1  "scatter": y[n,o,q]        = sum_{i,t,p : p*stride - pad + t == q} Ws[i][o][t] * act(x[n,i,p])
 * conv fwd = gather with W; conv dgrad = scatter with W; convT fwd = scatter with Wt;
 * convT dgrad = gather with Wt.  Weights are passed pre-packed (icm_pack_weights).
 */
typedef struct icm_conv_args {
  const float* x; int64_t x_bs; int N, Cin, H, W;
  const float* wp;
  const float* bias;
  float* y; int64_t y_bs; int Cout, OH, OW;
  int KH, KW, stride, pad;
  int transposed;
  int pro_act;
  int epi;
  const float* res; int64_t res_bs;
  const float* aux; int64_t aux_bs;
  const float* aux2; int64_t aux2_bs;
  float* y2; int64_t y2_bs;   /* second output (epi NONE / RES / RES_GELU: gelu(y); GDN: the norm; LRP: tanh); y2 == y with
                               * y2_bs == y_bs: store ONLY gelu(y), in y (inference: nobody reads the pre-activation) */
  int accum;          /* y += result instead of y = result (gradient accumulation; with y2: y2 = gelu(y after the add)) */
  int pixel_shuffle;  /* 2: fuse nn.PixelShuffle(2) into the store (layers.py:34-38); y plane is (2*OH,2*OW), Cout/4 channels */
  /* blocked input-channel map (0 = none): logical channel c reads plane c + (c / x_seg_len) * x_seg_gap of x -- the
   * K-concatenation of equally long channel runs that lie x_seg_len + x_seg_gap planes apart (the first-layer input
   * gradients of all slice chains of one family, cnn.py:89-127, contracted in ONE launch) */
  int x_seg_len, x_seg_gap;
  /* ICM_ALGO_DIRECT (0): implicit GEMM over the spatial taps.  ICM_ALGO_WINOGRAD (1): F(2x2, 3x3) -- 3x3 stride-1 pad-1
   * launches only (icm_conv_winograd_ok), forward or input gradient; wp must then hold the Winograd-domain weights
   * (icm_pack_job.wino: 1 for the forward orientation, 2 for the input-gradient orientation).  Same f32 arithmetic
   * type, 4/9 of the multiply-adds; results agree with the direct form to summation-order noise. */
  int algo;
  /* ICM_ALGO_WINOGRAD only, optional: the input pre-transformed by icm_wino_transform (icm_wino_transform_floats
   * floats, caller-owned).  When set, icm_conv_run reads xv instead of x (operand activation and channel map were
   * applied by the transform): one transform per input tensor serves every launch and co-block that reads it, and
   * the convolution kernel stages its operand by 16-byte LDS-DMA with no vector work next to the matrix pipe. */
  float* xv;
} icm_conv_args;
#define ICM_ALGO_DIRECT 0
#define ICM_ALGO_WINOGRAD 1
/* 1 if icm_conv_run accepts these arguments with algo = ICM_ALGO_WINOGRAD (geometry and epilogue kind supported) */
int icm_conv_winograd_ok(const icm_conv_args* a);
/* size of the pre-transformed operand of these arguments (-1 if unsupported), and the transform itself: fills
 * arr[i].xv from arr[i].x for up to 12 same-geometry members in one launch */
int64_t icm_wino_transform_floats(const icm_conv_args* a);
int icm_wino_transform(const icm_conv_args* arr, int ngroups, void* stream);

int icm_conv_run(const icm_conv_args* a, void* stream);
/* up to 3 problems of identical geometry in one launch (cc_mean || cc_scale chains, cnn.py:164-168) */
int icm_conv_run_grouped(const icm_conv_args* a, int ngroups, void* stream);
/* named views of icm_conv_run, one per reference op */
int icm_conv2d_fwd(const icm_conv_args* a, void* stream);     /* F.conv2d forward            */
int icm_conv2d_dgrad(const icm_conv_args* a, void* stream);   /* its input gradient          */
int icm_convT2d_fwd(const icm_conv_args* a, void* stream);    /* F.conv_transpose2d forward  */
int icm_convT2d_dgrad(const icm_conv_args* a, void* stream);  /* its input gradient          */

/* number of floats of a packed weight buffer for GEMM-M `Cout`, GEMM-K channels `Cin` */
int64_t icm_packed_weight_floats(int Cout, int Cin, int KH, int KW);
/* w: canonical weights. src_out_major=1: w is [Cout][Cin][KH][KW] (Conv2d.weight seen by conv fwd,
 * ConvTranspose2d.weight seen by convT dgrad); 0: w is [Cin][Cout][KH][KW].  transposed/stride/pad
 * select the tap order the consuming icm_conv_run call expects.  nonneg=1 applies
 * NonNegativeParametrizer (ops/parametrizers.py:46-49): max(w,bound)^2 - pedestal while packing. */
int icm_pack_weights(const float* w, float* wp, int Cout, int Cin, int KH, int KW, int src_out_major,
                     int transposed, int stride, int pad, int nonneg, float bound, float pedestal,
                     void* stream);

/* many packing jobs in few launches (the trainer re-packs every weight once per step, forward and dgrad forms) */
typedef struct icm_pack_job {
  const float* w; float* wp;
  int Cout, Cin, KH, KW, src_out_major, transposed, stride, pad, nonneg;
  float bound, pedestal;
  /* sub-matrix of a wider canonical weight: the source rows hold src_ld (0 = natural) entries of the INNER matrix index
   * (Cin when src_out_major, else Cout) and the job takes the entries [src_off, src_off + count) -- the input-channel
   * blocks of the slice chains' first layers (cnn.py:89-127: latent | support slices | own slice) */
  int src_ld, src_off;
  /* concatenation along GEMM-M: the destination holds dst_ncot (0 = natural) 32-row tiles per (chunk, tap) step and
   * this job's tiles start at dst_cot_off.  (Concatenation along GEMM-K needs no field: consecutive jobs write
   * consecutive chunk ranges, i.e. wp advanced by icm_packed_weight_floats of the preceding jobs.) */
  int dst_ncot, dst_cot_off;
  /* 0: spatial taps (in the order transposed / stride / pad select).  1 / 2: Winograd F(2x2,3x3) weights G g G^T of a
   * 3x3 stride-1 pad-1 kernel, 16 transform points per (Cout, Cin) pair: 1 = forward orientation, 2 = input-gradient
   * orientation (kernel rotated by 180 degrees; pass the transposed matrix roles as for the direct dgrad pack).
   * Buffer size: icm_packed_weight_floats(Cout, Cin, 4, 4). */
  int wino;
} icm_pack_job;
int icm_pack_weights_batch(const icm_pack_job* jobs, int n, void* stream);

/* weight gradient: dW[a][b][t] = sum_{n,p} actS(gs[n,a,p]) * actB(gb[n,b,p*stride - pad + t])
 * conv:  gs = dY (a = Cout), gb = x  (b = Cin) -> Conv2d.weight.grad
 * convT: gs = x  (a = Cin),  gb = dY (b = Cout) -> ConvTranspose2d.weight.grad
 * ws: workspace of icm_wgrad_workspace_floats(...) floats. accum: dw += result. */
typedef struct icm_wgrad_args {
  const float* gs; int64_t gs_bs; int Ca, OH, OW; int act_s;
  const float* gb; int64_t gb_bs; int Cb, H, W; int act_b;
  int N, KH, KW, stride, pad;
  float* dw; float* ws; int accum;
  float* dbias; int accum_bias;   /* optional: dbias[a] (+)= sum_{n,p} actS(gs[n,a,p]) (conv bias gradient, fused) */
  int64_t ws_floats;              /* capacity of ws in floats; 0 = unchecked. Too small -> ICM_ERR_ARG, nothing launched */
  int dw_ld;                      /* 0 = Cb; else the gradient lands in a [Ca][dw_ld][KH][KW] tensor whose b-columns
                                   * start at dw (an input-channel block of a wider weight); may differ per group member */
  int algo;                       /* ICM_ALGO_DIRECT, or ICM_ALGO_WINOGRAD for 3x3 stride-1 pad-1 problems (fewer than
                                   * 65 536 2x2 tiles): dU = (A dY A^T)(.)(B^T d B) summed over tiles, dW = G^T dU G;
                                   * the workspace size depends on it (icm_wgrad_workspace_floats*) */
} icm_wgrad_args;
int64_t icm_wgrad_workspace_floats(const icm_wgrad_args* a);
/* workspace PER PROBLEM when n problems of this geometry are issued by one icm_conv_wgrad_grouped call (the pixel
 * split count, hence the slab size, depends on how many problems share the launch) */
int64_t icm_wgrad_workspace_floats_grouped(const icm_wgrad_args* a, int n);
int icm_conv_wgrad(const icm_wgrad_args* a, void* stream);
/* up to 32 weight-gradient problems of identical geometry in one launch pair (deferred, batched wgrads: a
 * weight gradient has no consumer but the optimiser, so the 150 small slice-chain wgrads are collected during
 * backward and issued together); every arr[i].ws is that problem's own workspace */
int icm_conv_wgrad_grouped(const icm_wgrad_args* arr, int n, void* stream);

/* out[c] (+)= sum_{n,p} x[n,c,p]   (bias gradients; GDN d_beta).  ws (optional, ws_floats >= 32*C for full
 * parallelism): partial sums of the pixel splits, added in split order; ws = NULL -> one workgroup per channel */
int icm_channel_sum(const float* x, int64_t x_bs, int N, int C, int HW, float* out, int accum, float* ws,
                    int64_t ws_floats, void* stream);

/* ---- GDN helpers (layers/gdn.py:62-75, ops/parametrizers.py:46-49, ops/bound_ops.py:25-27) ---- */
int icm_nonneg_fwd(const float* p, float* out, int64_t n, float bound, float pedestal, void* stream);
/* dp (+)= lb_bwd(2*max(p,bound)*g_eff) */
int icm_nonneg_bwd(const float* p, const float* g_eff, float* dp, int64_t n, float bound, int accum, void* stream);
/* dn = g*x*(-/+ 1/2)*n^(-/+1/2 - 1), t1 = g*n^(-/+ 1/2) */
int icm_gdn_bwd_pre(const float* g, const float* x, const float* nrm, float* dn, float* t1, int64_t n, int inverse, void* stream);

/* ---- elementwise pieces of Win_noShift_Attention (layers/layers.py:83-89) and cnn.py:150-152 ---- */
int icm_gelu_fwd(const float* x, float* y, int64_t n, void* stream);
/* out = gelu(a) * sigmoid(b) + x  (a is the virtual-GELU pre-activation of conv_a) */
int icm_gate_fwd(const float* a_pre, const float* b, const float* x, float* out, int64_t n, void* stream);
/* da_pre (+)= g*sig(b)*gelu'(a_pre); db = g*gelu(a_pre)*sig(b)*(1-sig(b)); dx (+)= g */
int icm_gate_bwd(const float* g, const float* a_pre, const float* b, float* da_pre, float* db, float* dx,
                 int64_t n, int accum_da, int accum_dx, void* stream);
/* dst (+)= src * (mul_dgelu_of ? gelu'(mul_dgelu_of) : 1) */
int icm_add_grad(const float* src, const float* mul_dgelu_of, float* dst, int64_t n, int accum, void* stream);
/* z_hat[n,c,p] = rint(z - med[c]) - (z-med[c]) + (z-med[c]) + med[c]   (ops/ops.py:34, cnn.py:150-152) */
int icm_ste_round_offset(const float* z, const float* quantiles, float* z_hat, int N, int C, int HW, void* stream);
/* LRP tail backward (cnn.py:177-178): dpre = g * 0.5 * (1 - t*t), t = tanh saved by ICM_EPI_LRP */
int icm_lrp_bwd(const float* g, int64_t g_bs, const float* t, int64_t t_bs, float* dpre, int64_t d_bs, int N, int C,
                int HW, void* stream);
/* inverse of nn.PixelShuffle(2) (layers.py:34-38) for the subpel conv backward: src [N][C][2H][2W] -> dst [N][4C][H][W] */
int icm_pixel_unshuffle2(const float* src, float* dst, int N, int C, int H, int W, void* stream);
/* dst[i*len + e] = srcs[i][e], i < n <= 64 (srcs: HOST array of device pointers): small parameter vectors laid end
 * to end -- the first-layer biases of the slice chains whose first layers run as one concatenated GEMM */
int icm_gather_vectors(const float* const* srcs, int n, int len, float* dst, void* stream);
/* strided 4-D copy (chunk/cat plumbing): dst[n,c,p] (+)= src[n,c,p] */
int icm_copy_strided(const float* src, int64_t src_bs, float* dst, int64_t dst_bs, int N, int C, int HW, int accum, void* stream);
/* dst[b][a][k] (+)= src[a][b][K-1-k]  (src [A][B][K], dst [B][A][K]): a Conv2d weight [Cout][Cin][KH*KW] (reference:
 * torch.nn.Conv2d in stf.py:401-404 `end_conv`) rewritten as the ConvTranspose2d weight of the same stride-1 map, and
 * that weight's gradient carried back; used by the thin-output path (very few output channels) of the host layer. */
int icm_permute_flip(const float* src, float* dst, int A, int B, int K, int accum, void* stream);

/* ---- zigzag block ordering of the stf6 / oj_ICM variants (compressai/models/stf6.py:654-762; fasterRCNN_ICM.py:103-293)
 * The latent x [B,C,H,W] (batch stride x_bs) is a grid of num_slices x num_h x num_w contiguous blocks; the zigzag
 * tensor z [B, N, C/num_slices, H/num_h, W/num_w] (contiguous, N = num_slices*num_h*num_w <= 64) lists them shell by
 * shell.  C, H, W must divide exactly (the reference's view() needs the same).  icm_zigzag_order fills `order` with
 * (c*num_h + h)*num_w + w per output block and returns N (order == NULL: just N; -1 on bad arguments / capacity).
 * Each function is its own inverse's adjoint: backward of splits = reverse on the gradient, and vice versa. */
int icm_zigzag_order(int num_slices, int num_h, int num_w, int32_t* order, int capacity);
int icm_zigzag_splits(const float* x, int64_t x_bs, float* z, int B, int C, int H, int W, int num_slices, int num_h,
                      int num_w, void* stream);
int icm_zigzag_reverse(const float* z, float* x, int64_t x_bs, int B, int C, int H, int W, int num_slices, int num_h,
                       int num_w, void* stream);

/* ---- stf (Swin) pieces on NCHW tensors ------------------------------------------------------------------
 * nn.LayerNorm(C) over the channel axis per pixel (compressai/models/stf.py:136,142,209,250,372): y = (x-mean)*rstd*gamma+beta;
 * mean/rstd [N*HW] are saved for backward (may be NULL in inference) */
int icm_layernorm_fwd(const float* x, int64_t x_bs, const float* gamma, const float* beta, float* y, int64_t y_bs,
                      float* mean, float* rstd, int N, int C, int HW, float eps, void* stream);
/* dx_extra (optional, [N,C,HW] with its own batch stride): added to dx -- the identity-path gradient of the residual
 * add around the norm (x + f(LN(x)), stf.py:190-191) without a separate elementwise pass */
int icm_layernorm_bwd(const float* x, int64_t x_bs, const float* dy, int64_t dy_bs, const float* gamma,
                      const float* mean, const float* rstd, float* dx, int64_t dx_bs, float* dgamma, float* dbeta,
                      int N, int C, int HW, int accum_dx, int accum_params, const float* dx_extra, int64_t dx_extra_bs,
                      float* ws, int64_t ws_floats /* optional split partials of dgamma / dbeta: 2*(2048 + C) floats */,
                      void* stream);
/* PatchMerging's 2x2 gather (stf.py:224-228): dst[n][k*C+c][y][x] = src[n][c][2y+(k&1)][2x+(k>>1)]; inverse=1 scatters
 * a [N,4C,H/2,W/2] gradient back into [N,C,H,W] */
int icm_space_to_depth2(const float* src, float* dst, int N, int C, int H, int W, int inverse, int accum, void* stream);
/* DropPath residual (stf.py:190-191): out[n] = shortcut[n] + scale[n]*branch[n] (shortcut may be NULL) */
int icm_residual_scale(const float* shortcut, const float* branch, const float* scale, float* out, int N,
                       int64_t per_sample, void* stream);

/* ---- thin-channel convolutions (the 3-channel ends g_a.0 / g_s.8, cnn.py:32,51) as 1x1 GEMM + column passes ----
 * im2col: cols[n][c*K*K + kh*K + kw][oy][ox] = x[n][c][oy*stride - pad + kh][ox*stride - pad + kw] (0 outside);
 *   x [N,C,H,W] -> cols [N, C*K*K, OH, OW], OH = (H + 2*pad - K)/stride + 1.  Forward of a Conv2d with few input
 *   channels (then a 1x1 GEMM over C*K*K channels), and the gradient of icm_col2im.
 * col2im (adjoint): out[n][c][y][x] (+)= bias[c] + sum over taps with (y+pad-kh) % stride == 0 of
 *   cols[n][c*K*K + t][(y+pad-kh)/stride][(x+pad-kw)/stride]; cols [N, C*K*K, OH, OW] -> out [N,C,H,W].  Tail of a
 *   ConvTranspose2d with few output channels (models/utils.py:124-132), and the gradient of icm_im2col. */
int icm_im2col(const float* x, float* cols, int N, int C, int H, int W, int K, int stride, int pad, void* stream);
int icm_col2im(const float* cols, const float* bias, float* out, int N, int C, int H, int W, int K, int stride, int pad,
               int accum, void* stream);

/* ---- window attention core (layers/win_attention.py:84-115,153-207) --------------------------
 * qkv: [N][3*C][H][W] (output of the qkv Linear run as a 1x1 conv on NCHW; channel = which*C + head*hd + d)
 * out: [N][C][H][W] (channel = head*hd + d), input of the proj Linear.  Cyclic shift, window partition,
 * relative-position bias gather, the 0/-100 shift mask and the softmax are folded into addressing.
 * table: relative_position_bias_table [(2ws-1)^2][heads]. */
int icm_winattn_fwd(const float* qkv, const float* table, float* out, int N, int C, int H, int W,
                    int heads, int ws, int shift, void* stream);
/* dqkv = gradient wrt qkv (overwritten); dtable (+)= gradient wrt table (accum_table = 0 overwrites).  wsp: workspace
 * of icm_winattn_bwd_workspace_floats() floats holding one partial table per (window, head), reduced in window order */
int64_t icm_winattn_bwd_workspace_floats(int N, int C, int H, int W, int heads, int ws);
int icm_winattn_bwd(const float* qkv, const float* table, const float* dout, float* dqkv, float* dtable,
                    int accum_table, float* wsp, int64_t ws_floats, int N, int C, int H, int W, int heads, int ws,
                    int shift, void* stream);

/* ---- EntropyBottleneck (entropy_models.py:395-433,446-489) ---------------------------------
 * params: the 13 tensors concatenated per channel is NOT required; pointers are passed separately.
 * z [N][C][HW]; noise NULL -> eval (dequantize around medians) else z + noise.
 * lik out [N][C][HW]; zt (z tilde) optional out. */
typedef struct icm_eb_params {
  const float* matrix[5]; const float* bias[5]; const float* factor[4]; const float* quantiles;
} icm_eb_params;
typedef struct icm_eb_grads {
  float* matrix[5]; float* bias[5]; float* factor[4];
  float* dmedian; /* [C], eval mode only (z~ = round(z-med)+med passes gradient to med, not z); may be NULL */
} icm_eb_grads;
int icm_eb_likelihood_fwd(const float* z, const float* noise, const icm_eb_params* p, float* lik, float* zt,
                          int N, int C, int HW, float lik_bound, void* stream);
/* dz (+)= ..., param grads overwritten (one workgroup owns a channel: deterministic) */
int icm_eb_likelihood_bwd(const float* z, const float* noise, const icm_eb_params* p, const float* dlik,
                          float* dz, const icm_eb_grads* g, int N, int C, int HW, float lik_bound,
                          int accum_dz, void* stream);
/* aux loss = sum |F(quantiles) - target| and its gradient wrt quantiles (entropy_models.py:395-398) */
int icm_eb_aux_loss(const icm_eb_params* p, float* loss /*1 float, overwritten*/, float* dquantiles, int C,
                    float target, void* stream);

/* ---- GaussianConditional likelihood fused with ste_round (entropy_models.py:626-659, cnn.py:171-173)
 * y slice [N][C][HW] with batch stride y_bs; mu/scale [N][C][HW] with strides; noise may be NULL (eval).
 * lik -> [N][C][HW] (lik_bs); y_hat = ste_round(y-mu)+mu -> (yh_bs), optional second copy yh2. */
int icm_gc_likelihood_ste_fwd(const float* y, int64_t y_bs, const float* mu, int64_t mu_bs, const float* scale,
                              int64_t sc_bs, const float* noise, int64_t nz_bs, float* lik, int64_t lik_bs,
                              float* yh, int64_t yh_bs, float* yh2, int64_t yh2_bs, int N, int C, int HW,
                              float scale_bound, float lik_bound, void* stream);
/* inputs: dlik, dyh (gradient wrt y_hat_pre; may be NULL). outputs: dy (+)=, dmu =, dscale = */
int icm_gc_likelihood_ste_bwd(const float* y, int64_t y_bs, const float* mu, int64_t mu_bs, const float* scale,
                              int64_t sc_bs, const float* noise, int64_t nz_bs, const float* dlik, int64_t dl_bs,
                              const float* dyh, int64_t dyh_bs, float* dy, int64_t dy_bs, float* dmu, int64_t dmu_bs,
                              float* dscale, int64_t dsc_bs, int N, int C, int HW, float scale_bound,
                              float lik_bound, int accum_dy, void* stream);

/* ---- entropy coding: CDF tables and the rANS stream -----------------------------------------
 * Replaces the calls entropy_models.py:60-63,172-290 makes into the reference's binary-only extensions
 * `compressai._CXX.pmf_to_quantized_cdf` and `compressai.ans.{RansEncoder,RansDecoder,BufferedRansEncoder}`
 * (CompressAI 1.1.6dev0 cpp_exts, sources absent from the tree; state machine = third_party/ryg_rans/rans64.h).
 * HOST functions: pointers are host memory, nothing touches the GPU.
 *
 * cdf[0..n]: quantised cumulative table of pmf[0..n-1] with cdf[n] = 2^precision and no zero-width symbol. */
int icm_pmf_to_quantized_cdf(const float* pmf, int n, int precision, int32_t* cdf);
/* cdfs: ncdf tables of cdf_stride int32 each (row i valid for cdf_sizes[i] entries, last bin = escape);
 * symbol i is coded with table indexes[i] after subtracting offsets[indexes[i]].  Returns the stream length in bytes
 * (a multiple of 4), -1 on bad arguments / capacity; out = NULL only measures. */
int64_t icm_rans_encode_with_indexes(const int32_t* symbols, const int32_t* indexes, int64_t n, const int32_t* cdfs,
                                     int cdf_stride, const int32_t* cdf_sizes, const int32_t* offsets, int ncdf,
                                     uint8_t* out, int64_t out_capacity);
int icm_rans_decode_with_indexes(const uint8_t* stream, int64_t nbytes, const int32_t* indexes, int64_t n,
                                 const int32_t* cdfs, int cdf_stride, const int32_t* cdf_sizes, const int32_t* offsets,
                                 int ncdf, int32_t* out);
/* RansDecoder.set_stream / decode_stream (cnn.py:300-318): one stream consumed by successive calls */
void* icm_rans_decoder_create(const uint8_t* stream, int64_t nbytes);
int icm_rans_decoder_decode(void* decoder, const int32_t* indexes, int64_t n, const int32_t* cdfs, int cdf_stride,
                            const int32_t* cdf_sizes, const int32_t* offsets, int ncdf, int32_t* out);
void icm_rans_decoder_destroy(void* decoder);

/* device side of update() / compress() / decompress() (stream-ordered like every other kernel entry point) */
/* EntropyBottleneck.update (entropy_models.py:354-393): minima / maxima [C] int32, then pmf [C][max_length] and
 * tail_mass [C] on the integer grid median - minima + k */
int icm_eb_table_bounds(const float* quantiles, int C, int32_t* minima, int32_t* maxima, void* stream);
int icm_eb_pmf_table(const icm_eb_params* p, const int32_t* minima, int C, int max_length, float* pmf, float* tail_mass,
                     void* stream);
/* GaussianConditional.update (entropy_models.py:598-624): centers = ceil(scale_table * multiplier); pmf [ns][max_length] */
int icm_gc_table_centers(const float* scale_table, int ns, float multiplier, int32_t* centers, void* stream);
int icm_gc_pmf_table(const float* scale_table, const int32_t* centers, int ns, int max_length, float* pmf,
                     float* tail_mass, void* stream);
/* GaussianConditional.build_indexes (entropy_models.py:661-666) */
int icm_gc_build_indexes(const float* scale, int64_t scale_bs, const float* scale_table, int ns, float scale_bound,
                         int32_t* indexes, int N, int C, int HW, void* stream);
/* EntropyModel.quantize "symbols" / "dequantize" (entropy_models.py:126-150) and dequantize (:159-166); the mean of
 * element (n,c,p) is means[n*m_bs + c*m_cs + p*m_ps] (means NULL = 0): full tensors or per-channel medians */
int icm_quantize(const float* x, int64_t x_bs, const float* means, int64_t m_bs, int64_t m_cs, int64_t m_ps,
                 int32_t* symbols, float* dequantized, int N, int C, int HW, void* stream);
int icm_dequantize(const int32_t* symbols, const float* means, int64_t m_bs, int64_t m_cs, int64_t m_ps, float* out,
                   int64_t out_bs, int N, int C, int HW, void* stream);

/* ---- R-D loss (train.py:53-61, train_czigzag.py:63,71) --------------------------------------
 * out[0]=bpp, out[1]=mse, out[2]=loss, out[3]=sum log(lik_y), out[4]=sum log(lik_z) (5 floats, overwritten);
 * ws: ICM_REDUCE_WS_FLOATS floats of scratch (per-workgroup partial sums, added in workgroup order). */
#define ICM_REDUCE_WS_FLOATS 8192
int icm_rd_loss_fwd(const float* x, const float* x_hat, int64_t n_img_elems, const float* lik_y, int64_t n_y,
                    const float* lik_z, int64_t n_z, int64_t num_pixels, float lmbda, float* out, float* ws,
                    void* stream);
/* dx_hat = gscale * lmbda*255^2*2*(x_hat-x)/n ; dlik = gscale * -1/(lik*ln2*num_pixels) */
int icm_rd_loss_bwd(const float* x, const float* x_hat, int64_t n_img_elems, const float* lik_y, int64_t n_y,
                    const float* lik_z, int64_t n_z, int64_t num_pixels, float lmbda, float gscale,
                    float* dx_hat, float* dlik_y, float* dlik_z, void* stream);

/* ---- optimiser (train.py:105-169,199-214) ---------------------------------------------------- */
/* out[0] = sum g^2; ws: ICM_REDUCE_WS_FLOATS floats of scratch (fixed-order two-stage sum: every rank of a
 * data-parallel job derives the same clip coefficient from the same all-reduced gradient) */
int icm_grad_sqnorm(const float* g, int64_t n, float* out, float* ws, void* stream);
/* Adam step with fused clip: coef = min(1, max_norm/(sqrt(*sqnorm)+1e-6)) if sqnorm!=NULL else 1;
 * g is scaled by gscale (1/world_size) then coef; torch.optim.Adam defaults semantics. */
int icm_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                  double eps, int step, const float* sqnorm, float max_norm, float gscale, void* stream);
/* the same update with the two step-dependent scalars read from DEVICE memory -- hyper[0] = lr / (1 - beta1^step),
 * hyper[1] = sqrt(1 - beta2^step), formed by the caller in double and rounded to f32 exactly as icm_adam_step forms
 * them -- so that a captured hipGraph of the training step can be replayed (kernel arguments are frozen at capture) */
int icm_adam_step_hyper(float* p, const float* g, float* m, float* v, int64_t n, double beta1, double beta2, double eps,
                        const float* hyper, const float* sqnorm, float max_norm, float gscale, void* stream);
int icm_fill(float* p, int64_t n, float v, void* stream);
/* x_hat.clamp_(0, 1) of decompress() (cnn.py:330) */
int icm_clamp(float* p, int64_t n, float lo, float hi, void* stream);
/* F.pad(x, (left, ., top, .), value) / its negative-pad crop around the codec (utils/eval_model/__main__.py:102-117,
 * 129-131): dst [N,C,OH,OW], dst[y][x] = src[y - top][x - left] inside the source, `value` outside */
int icm_pad2d(const float* src, int N, int C, int H, int W, float* dst, int OH, int OW, int top, int left, float value,
              void* stream);

/* ---- test hooks (process-global; used by the parity tests and tools/tune_conv.py only) ----------------------
 * force the implicit-GEMM tile configuration (index into the kernel table; 100 / 101 = the 8-wave K-split kernel of
 * conv_ks8.hip with 64 co x 64 px / 32 co x 128 px blocks where it is eligible; -1 = automatic) / the weight-gradient
 * kernel variant (0-2 general kernel <2,1,7> / <2,2,9> / <4,4,4>; 3-6 three-by-three tiles per wave <3,3> / <6,6> /
 * <3,6> / <6,3>; 7 its tap-per-wave form; 8 / 9 nine taps from one DMA staging, 96 / 64-wide blocks; 10 the 25-tap
 * 5x5 form; 11-14 the DMA-only 1x1 kernel <6,6> / <3,6> / <6,3> / <3,3>; -1 = automatic) and its XCD-aware
 * workgroup order (0 / 1, -1 = automatic) */
void icm_debug_force_conv_cfg(int idx);
/* the value last set (-1 = automatic): callers that choose between the direct and the Winograd form keep the direct
 * one while a tile configuration is forced */
int icm_debug_forced_conv_cfg(void);
/* pointwise (1x1 stride-1, Cin % 8 == 0) convolutions: -1 = automatic (the barrier-free direct-operand kernel when the
 * launch has >= 1024 waves), 0 = always the LDS-staged kernel, 1 = the direct kernel whenever eligible */
void icm_debug_force_conv1x1(int mode);
void icm_debug_force_wgrad_cfg(int variant, int xcd_order);
/* 1: window attention always runs the generic VALU kernels (the matrix-core kernels cover 8x8 windows) */
void icm_debug_force_winattn_valu(int on);

#ifdef __cplusplus
}
#endif
#endif
