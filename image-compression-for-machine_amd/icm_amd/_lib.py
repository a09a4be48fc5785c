"""ctypes binding of libicm_hip.so (the C ABI declared in include/icm_hip.h).

There is deliberately NO fallback: if the HIP library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ICM_LIB", os.path.join(os.path.dirname(_HERE), "lib", "libicm_hip.so"))

ACT_NONE, ACT_GELU, ACT_SQUARE = 0, 1, 2
EPI_NONE, EPI_RES, EPI_RES_GELU, EPI_GDN, EPI_IGDN, EPI_MUL_DGELU, EPI_AXPY2, EPI_LRP, EPI_RES_MUL_DGELU = range(9)

_f = C.POINTER(C.c_float)
i64, i32, f32, vp = C.c_int64, C.c_int, C.c_float, C.c_void_p


class ConvArgs(C.Structure):
    _fields_ = [
        ("x", vp), ("x_bs", i64), ("N", i32), ("Cin", i32), ("H", i32), ("W", i32),
        ("wp", vp), ("bias", vp),
        ("y", vp), ("y_bs", i64), ("Cout", i32), ("OH", i32), ("OW", i32),
        ("KH", i32), ("KW", i32), ("stride", i32), ("pad", i32),
        ("transposed", i32), ("pro_act", i32), ("epi", i32),
        ("res", vp), ("res_bs", i64), ("aux", vp), ("aux_bs", i64), ("aux2", vp), ("aux2_bs", i64),
        ("y2", vp), ("y2_bs", i64),
        ("accum", i32), ("pixel_shuffle", i32), ("x_seg_len", i32), ("x_seg_gap", i32), ("algo", i32), ("xv", vp),
    ]


class WgradArgs(C.Structure):
    _fields_ = [
        ("gs", vp), ("gs_bs", i64), ("Ca", i32), ("OH", i32), ("OW", i32), ("act_s", i32),
        ("gb", vp), ("gb_bs", i64), ("Cb", i32), ("H", i32), ("W", i32), ("act_b", i32),
        ("N", i32), ("KH", i32), ("KW", i32), ("stride", i32), ("pad", i32),
        ("dw", vp), ("ws", vp), ("accum", i32),
        ("dbias", vp), ("accum_bias", i32), ("ws_floats", i64), ("dw_ld", i32), ("algo", i32),
    ]


class PackJob(C.Structure):
    _fields_ = [("w", vp), ("wp", vp), ("Cout", i32), ("Cin", i32), ("KH", i32), ("KW", i32), ("src_out_major", i32),
                ("transposed", i32), ("stride", i32), ("pad", i32), ("nonneg", i32), ("bound", f32), ("pedestal", f32),
                ("src_ld", i32), ("src_off", i32), ("dst_ncot", i32), ("dst_cot_off", i32), ("wino", i32)]


class EbParams(C.Structure):
    _fields_ = [("matrix", vp * 5), ("bias", vp * 5), ("factor", vp * 4), ("quantiles", vp)]


class EbGrads(C.Structure):
    _fields_ = [("matrix", vp * 5), ("bias", vp * 5), ("factor", vp * 4), ("dmedian", vp)]


class IcmError(RuntimeError):
    pass


_lib = None

# every symbol include/icm_hip.h declares (tests check that the .so exports them all)
SYMBOLS = [
    "icm_strerror", "icm_version", "icm_conv_run", "icm_conv_run_grouped", "icm_conv2d_fwd", "icm_conv2d_dgrad",
    "icm_convT2d_fwd", "icm_convT2d_dgrad", "icm_packed_weight_floats", "icm_pack_weights", "icm_pack_weights_batch",
    "icm_wgrad_workspace_floats", "icm_wgrad_workspace_floats_grouped", "icm_conv_wgrad", "icm_conv_wgrad_grouped", "icm_channel_sum", "icm_nonneg_fwd", "icm_nonneg_bwd",
    "icm_gdn_bwd_pre", "icm_gelu_fwd", "icm_gate_fwd", "icm_gate_bwd", "icm_add_grad", "icm_ste_round_offset",
    "icm_lrp_bwd", "icm_pixel_unshuffle2", "icm_layernorm_fwd", "icm_layernorm_bwd", "icm_space_to_depth2",
    "icm_residual_scale", "icm_im2col", "icm_col2im", "icm_copy_strided", "icm_permute_flip", "icm_gather_vectors", "icm_conv_winograd_ok", "icm_wino_transform_floats", "icm_wino_transform", "icm_winattn_fwd", "icm_winattn_bwd",
    "icm_eb_likelihood_fwd", "icm_eb_likelihood_bwd", "icm_eb_aux_loss", "icm_gc_likelihood_ste_fwd",
    "icm_gc_likelihood_ste_bwd", "icm_rd_loss_fwd", "icm_rd_loss_bwd", "icm_grad_sqnorm", "icm_adam_step", "icm_adam_step_hyper", "icm_fill",
    "icm_winattn_bwd_workspace_floats", "icm_debug_force_conv_cfg", "icm_debug_forced_conv_cfg", "icm_debug_force_conv1x1",
    "icm_debug_force_wgrad_cfg",
    "icm_debug_force_winattn_valu",
    "icm_zigzag_order", "icm_zigzag_splits", "icm_zigzag_reverse",
    "icm_pmf_to_quantized_cdf", "icm_rans_encode_with_indexes", "icm_rans_decode_with_indexes",
    "icm_rans_decoder_create", "icm_rans_decoder_decode", "icm_rans_decoder_destroy",
    "icm_eb_table_bounds", "icm_eb_pmf_table", "icm_gc_table_centers", "icm_gc_pmf_table", "icm_gc_build_indexes",
    "icm_quantize", "icm_dequantize", "icm_clamp", "icm_pad2d",
]
REDUCE_WS_FLOATS = 8192   # ICM_REDUCE_WS_FLOATS


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IcmError(
                f"{LIB_PATH} not found: build it with image-compression-for-machine_amd/build.sh "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        L.icm_strerror.restype = C.c_char_p
        L.icm_packed_weight_floats.restype = i64
        L.icm_wgrad_workspace_floats.restype = i64
        L.icm_wgrad_workspace_floats_grouped.restype = i64
        L.icm_packed_weight_floats.argtypes = [i32, i32, i32, i32]
        L.icm_pack_weights.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, f32, f32, vp]
        L.icm_pack_weights_batch.argtypes = [C.POINTER(PackJob), i32, vp]
        L.icm_conv_run.argtypes = [C.POINTER(ConvArgs), vp]
        L.icm_conv_winograd_ok.argtypes = [C.POINTER(ConvArgs)]
        L.icm_wino_transform_floats.argtypes = [C.POINTER(ConvArgs)]
        L.icm_wino_transform_floats.restype = i64
        L.icm_wino_transform.argtypes = [C.POINTER(ConvArgs), i32, vp]
        L.icm_conv_run_grouped.argtypes = [C.POINTER(ConvArgs), i32, vp]
        for n in ("icm_conv2d_fwd", "icm_conv2d_dgrad", "icm_convT2d_fwd", "icm_convT2d_dgrad"):
            getattr(L, n).argtypes = [C.POINTER(ConvArgs), vp]
        L.icm_wgrad_workspace_floats.argtypes = [C.POINTER(WgradArgs)]
        L.icm_wgrad_workspace_floats_grouped.argtypes = [C.POINTER(WgradArgs), i32]
        L.icm_conv_wgrad.argtypes = [C.POINTER(WgradArgs), vp]
        L.icm_conv_wgrad_grouped.argtypes = [C.POINTER(WgradArgs), i32, vp]
        L.icm_channel_sum.argtypes = [vp, i64, i32, i32, i32, vp, i32, vp, i64, vp]
        L.icm_gather_vectors.argtypes = [C.POINTER(vp), i32, i32, vp, vp]
        L.icm_nonneg_fwd.argtypes = [vp, vp, i64, f32, f32, vp]
        L.icm_nonneg_bwd.argtypes = [vp, vp, vp, i64, f32, i32, vp]
        L.icm_gdn_bwd_pre.argtypes = [vp, vp, vp, vp, vp, i64, i32, vp]
        L.icm_gelu_fwd.argtypes = [vp, vp, i64, vp]
        L.icm_gate_fwd.argtypes = [vp, vp, vp, vp, i64, vp]
        L.icm_gate_bwd.argtypes = [vp, vp, vp, vp, vp, vp, i64, i32, i32, vp]
        L.icm_add_grad.argtypes = [vp, vp, vp, i64, i32, vp]
        L.icm_ste_round_offset.argtypes = [vp, vp, vp, i32, i32, i32, vp]
        L.icm_lrp_bwd.argtypes = [vp, i64, vp, i64, vp, i64, i32, i32, i32, vp]
        L.icm_pixel_unshuffle2.argtypes = [vp, vp, i32, i32, i32, i32, vp]
        L.icm_layernorm_fwd.argtypes = [vp, i64, vp, vp, vp, i64, vp, vp, i32, i32, i32, f32, vp]
        L.icm_layernorm_bwd.argtypes = [vp, i64, vp, i64, vp, vp, vp, vp, i64, vp, vp, i32, i32, i32, i32, i32, vp, i64,
                                        vp, i64, vp]
        L.icm_space_to_depth2.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, vp]
        L.icm_residual_scale.argtypes = [vp, vp, vp, vp, i32, i64, vp]
        L.icm_im2col.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
        L.icm_col2im.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]
        L.icm_copy_strided.argtypes = [vp, i64, vp, i64, i32, i32, i32, i32, vp]
        L.icm_permute_flip.argtypes = [vp, vp, i32, i32, i32, i32, vp]
        L.icm_winattn_fwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
        L.icm_winattn_bwd.argtypes = [vp, vp, vp, vp, vp, i32, vp, i64, i32, i32, i32, i32, i32, i32, i32, vp]
        L.icm_winattn_bwd_workspace_floats.argtypes = [i32, i32, i32, i32, i32, i32]
        L.icm_winattn_bwd_workspace_floats.restype = i64
        L.icm_debug_force_conv_cfg.argtypes = [i32]
        L.icm_debug_force_conv_cfg.restype = None
        L.icm_debug_force_conv1x1.argtypes = [i32]
        L.icm_debug_force_conv1x1.restype = None
        L.icm_debug_force_wgrad_cfg.argtypes = [i32, i32]
        L.icm_debug_force_wgrad_cfg.restype = None
        L.icm_debug_force_winattn_valu.argtypes = [i32]
        L.icm_debug_force_winattn_valu.restype = None
        L.icm_eb_likelihood_fwd.argtypes = [vp, vp, C.POINTER(EbParams), vp, vp, i32, i32, i32, f32, vp]
        L.icm_eb_likelihood_bwd.argtypes = [vp, vp, C.POINTER(EbParams), vp, vp, C.POINTER(EbGrads), i32, i32, i32,
                                            f32, i32, vp]
        L.icm_eb_aux_loss.argtypes = [C.POINTER(EbParams), vp, vp, i32, f32, vp]
        L.icm_gc_likelihood_ste_fwd.argtypes = [vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, i32,
                                                i32, i32, f32, f32, vp]
        L.icm_gc_likelihood_ste_bwd.argtypes = [vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp,
                                                i64, vp, i64, i32, i32, i32, f32, f32, i32, vp]
        L.icm_rd_loss_fwd.argtypes = [vp, vp, i64, vp, i64, vp, i64, i64, f32, vp, vp, vp]
        L.icm_rd_loss_bwd.argtypes = [vp, vp, i64, vp, i64, vp, i64, i64, f32, f32, vp, vp, vp, vp]
        L.icm_grad_sqnorm.argtypes = [vp, i64, vp, vp, vp]
        L.icm_adam_step.argtypes = [vp, vp, vp, vp, i64, C.c_double, C.c_double, C.c_double, C.c_double, i32, vp, f32, f32, vp]
        L.icm_adam_step_hyper.argtypes = [vp, vp, vp, vp, i64, C.c_double, C.c_double, C.c_double, vp, vp, f32, f32, vp]
        L.icm_fill.argtypes = [vp, i64, f32, vp]
        # entropy coding (host) + its device-side table / symbol kernels
        i32p, u8p = C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
        L.icm_pmf_to_quantized_cdf.argtypes = [C.POINTER(C.c_float), i32, i32, i32p]
        L.icm_rans_encode_with_indexes.argtypes = [i32p, i32p, i64, i32p, i32, i32p, i32p, i32, vp, i64]
        L.icm_rans_encode_with_indexes.restype = i64
        L.icm_rans_decode_with_indexes.argtypes = [vp, i64, i32p, i64, i32p, i32, i32p, i32p, i32, i32p]
        L.icm_rans_decoder_create.argtypes = [vp, i64]
        L.icm_rans_decoder_create.restype = vp
        L.icm_rans_decoder_decode.argtypes = [vp, i32p, i64, i32p, i32, i32p, i32p, i32, i32p]
        L.icm_rans_decoder_destroy.argtypes = [vp]
        L.icm_rans_decoder_destroy.restype = None
        L.icm_eb_table_bounds.argtypes = [vp, i32, vp, vp, vp]
        L.icm_eb_pmf_table.argtypes = [C.POINTER(EbParams), vp, i32, i32, vp, vp, vp]
        L.icm_gc_table_centers.argtypes = [vp, i32, f32, vp, vp]
        L.icm_gc_pmf_table.argtypes = [vp, vp, i32, i32, vp, vp, vp]
        L.icm_gc_build_indexes.argtypes = [vp, i64, vp, i32, f32, vp, i32, i32, i32, vp]
        L.icm_quantize.argtypes = [vp, i64, vp, i64, i64, i64, vp, vp, i32, i32, i32, vp]
        L.icm_dequantize.argtypes = [vp, vp, i64, i64, i64, vp, i64, i32, i32, i32, vp]
        L.icm_clamp.argtypes = [vp, i64, f32, f32, vp]
        L.icm_pad2d.argtypes = [vp, i32, i32, i32, i32, vp, i32, i32, i32, i32, f32, vp]
        L.icm_zigzag_order.argtypes = [i32, i32, i32, vp, i32]
        L.icm_zigzag_splits.argtypes = [vp, i64, vp, i32, i32, i32, i32, i32, i32, i32, vp]
        L.icm_zigzag_reverse.argtypes = [vp, vp, i64, i32, i32, i32, i32, i32, i32, i32, vp]
        _lib = L
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().icm_strerror(rc).decode()
        if rc == 1:
            raise ValueError(f"icm {what}: {msg}")
        raise IcmError(f"icm {what}: {msg} (code {rc})")


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return 0 if t is None else t.data_ptr()


def bs(t) -> int:
    """batch stride (floats) of a [N,C,H,W] tensor whose C/H/W strides are canonical (channel slices allowed)."""
    if t is None:
        return 0
    assert t.dtype == torch.float32 and t.is_cuda, "icm ops need float32 device tensors"
    if t.dim() == 4:
        N, Cc, H, W = t.shape
        st = t.stride()
        assert (W == 1 or st[3] == 1) and (H == 1 or st[2] == W) and (Cc == 1 or st[1] == H * W), \
            f"non-canonical plane strides {st} for shape {tuple(t.shape)}"
        return st[0] if N > 1 else Cc * H * W
    assert t.is_contiguous()
    return t[0].numel() if t.dim() > 0 else 1
