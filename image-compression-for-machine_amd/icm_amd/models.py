"""compressai/models mirror for the `cnn` model: CompressionModel (models/base.py) and WACNN (models/cnn.py).

``WACNN.forward(x)`` keeps the reference contract ``{"x_hat", "likelihoods": {"y", "z"}}`` and the 585-key
state-dict, but executes as ONE hand-scheduled tape of HIP kernels (``wacnn_forward``): torch.cat / chunk are
replaced by writes into persistent support buffers, GELU / PixelShuffle / residual adds / the LRP tail are
fused into the implicit-GEMM prologues and epilogues, and the backward pass is the tape's closures."""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _lib as L
from . import engine as E
from ._lib import ACT_GELU
from .engine import VT
from .entropy_models import EntropyBottleneck, GaussianConditional
from .layers import (GDN, Win_noShift_Attention, conv, conv3x3, deconv, subpel_conv3x3, _named)

SCALES_MIN, SCALES_MAX, SCALES_LEVELS = 0.11, 256, 64


def get_scale_table(min=SCALES_MIN, max=SCALES_MAX, levels=SCALES_LEVELS):
    return torch.exp(torch.linspace(math.log(min), math.log(max), levels))


class CompressionModel(nn.Module):
    """models/base.py:5-70."""

    def __init__(self, init_weights=True):
        super().__init__()
        # the reference calls _initialize_weights() here, before any sub-module exists: a no-op

    def aux_loss(self):
        return sum(m.loss() for m in self.modules() if isinstance(m, EntropyBottleneck))

    def forward(self, *args):
        raise NotImplementedError()

    def update(self, force=False):
        raise NotImplementedError("CDF update / entropy coding is outside the training hot path (SURVEY 8 f2)")

    def load_state_dict(self, state_dict, strict: bool = False):
        # models/base.py:62-70: resize the CDF buffers to whatever the checkpoint holds, then strict=False
        sd = dict(state_dict)
        for mod_name in ("entropy_bottleneck", "gaussian_conditional"):
            mod = getattr(self, mod_name, None)
            if mod is None:
                continue
            for b in ("_quantized_cdf", "_offset", "_cdf_length", "scale_table"):
                k = f"{mod_name}.{b}"
                if k in sd and hasattr(mod, b):
                    buf = getattr(mod, b)
                    if buf.numel() == 0 and sd[k].numel() != 0:
                        setattr(mod, b, torch.empty(sd[k].shape, dtype=buf.dtype, device=buf.device))
        return super().load_state_dict(sd, strict=False)


def _chain(tape, P, p, xv: VT, strides=(1, 1, 1, 1, 1), out=None, lrp_aux=None):
    """five conv3x3 with virtual GELUs between (cnn.py:54-64, 89-127)."""
    for j, i in enumerate((0, 2, 4, 6, 8)):
        last = j == 4
        t = E.conv2d(tape, xv, P[f"{p}.{i}.weight"], P[f"{p}.{i}.bias"], stride=strides[j], pad=1,
                     out=out if last else None, lrp_aux=lrp_aux if last else None)
        xv = VT(t, ACT_GELU)
    return t


def _chain_pair(tape, P, p1, p2, xv1: VT, xv2: VT):
    """cc_mean_transforms[i] and cc_scale_transforms[i] are independent and shape-identical: run each of
    their five convolutions as one grouped launch (forward and dgrad)."""
    for i in (0, 2, 4, 6, 8):
        t1, t2 = E.conv2d_group(tape, [xv1, xv2], [P[f"{p1}.{i}.weight"], P[f"{p2}.{i}.weight"]],
                                [P[f"{p1}.{i}.bias"], P[f"{p2}.{i}.bias"]], pad=1)
        xv1, xv2 = VT(t1, ACT_GELU), VT(t2, ACT_GELU)
    return t1, t2


def _h_s(tape, P, p, z_hat, out):
    """h_mean_s / h_scale_s (cnn.py:66-88): PixelShuffle fused into the subpel convs' stores."""
    u = E.conv2d(tape, VT(z_hat), P[p + ".0.weight"], P[p + ".0.bias"], pad=1)
    u = E.conv2d(tape, VT(u, ACT_GELU), P[p + ".2.0.weight"], P[p + ".2.0.bias"], pad=1, pixel_shuffle=2)
    u = E.conv2d(tape, VT(u, ACT_GELU), P[p + ".4.weight"], P[p + ".4.bias"], pad=1)
    u = E.conv2d(tape, VT(u, ACT_GELU), P[p + ".6.0.weight"], P[p + ".6.0.bias"], pad=1, pixel_shuffle=2)
    return E.conv2d(tape, VT(u, ACT_GELU), P[p + ".8.weight"], P[p + ".8.bias"], pad=1, out=out)


def _copy_op(tape, src, dst):
    """dst = src (cat/chunk plumbing) with gradient routed back"""
    E.copy_into(tape, src, dst)
    if tape.need_grad:
        def bwd():
            g = tape.grad_of(dst)
            if g is not None:
                E.accumulate(tape, src, g)
        tape.bw.append(bwd)


def wacnn_forward(tape: E.Tape, P: Dict[str, torch.Tensor], x: torch.Tensor, noise_z=None, noise_y=None,
                  num_slices: int = 10, max_support: int = 5, keep: Optional[dict] = None,
                  bucket_marks: Optional[dict] = None):
    """WACNN.forward (models/cnn.py:141-189) on the HIP engine -> (x_hat, y_likelihoods, z_likelihoods)."""
    dev = x.device
    N = x.shape[0]
    need = tape.need_grad
    # ---- g_a (cnn.py:31-41)
    t = E.conv2d(tape, VT(x), P["g_a.0.weight"], P["g_a.0.bias"], stride=2, pad=2)
    t = E.gdn(tape, t, P["g_a.1.beta"], P["g_a.1.gamma"], False)
    t = E.conv2d(tape, VT(t), P["g_a.2.weight"], P["g_a.2.bias"], stride=2, pad=2)
    t = E.gdn(tape, t, P["g_a.3.beta"], P["g_a.3.gamma"], False)
    t = E.attention_gate(tape, t, P, "g_a.4", 8, 8, 4)
    t = E.conv2d(tape, VT(t), P["g_a.5.weight"], P["g_a.5.bias"], stride=2, pad=2)
    t = E.gdn(tape, t, P["g_a.6.beta"], P["g_a.6.gamma"], False)
    t = E.conv2d(tape, VT(t), P["g_a.7.weight"], P["g_a.7.bias"], stride=2, pad=2)
    y = E.attention_gate(tape, t, P, "g_a.8", 8, 4, 2)
    M, h, w = y.shape[1], y.shape[2], y.shape[3]
    if M % num_slices != 0:
        raise ValueError("latent channels must divide into num_slices")
    sc_ = M // num_slices
    if need:  # y is consumed whole (h_a) and by slices (GaussianConditional): one aliased gradient buffer
        dY = E.zeros(y.shape, dev)
        tape.bind_grad(y, dY, True)
        for i in range(num_slices):
            tape.bind_grad(y[:, i * sc_:(i + 1) * sc_], dY[:, i * sc_:(i + 1) * sc_], True)
    if bucket_marks is not None:
        bucket_marks[2] = len(tape.bw)   # backward reaching here => hyper-path gradients are complete
    # ---- h_a + entropy bottleneck (cnn.py:144-152)
    z = _chain(tape, P, "h_a", VT(y), strides=(1, 1, 2, 1, 2))
    _, z_lik = E.eb_likelihood(tape, z, P, "entropy_bottleneck", noise_z)
    z_hat = E.ste_round_medians(tape, z, P["entropy_bottleneck.quantiles"])
    # ---- hyper synthesis straight into the support buffers (cnn.py:154-155,163,167)
    nsup = sc_ * max_support
    MS = E.new((N, M + nsup, h, w), dev)
    SS = E.new((N, M + nsup, h, w), dev)
    if need:
        dMS, dSS = E.zeros(MS.shape, dev), E.zeros(SS.shape, dev)
        for k in range(max_support + 1):
            tape.bind_grad(MS[:, :M + sc_ * k], dMS[:, :M + sc_ * k], True)
            tape.bind_grad(SS[:, :M + sc_ * k], dSS[:, :M + sc_ * k], True)
        for j in range(max_support):
            sl = slice(M + sc_ * j, M + sc_ * (j + 1))
            tape.bind_grad(MS[:, sl], dMS[:, sl], True)
            tape.bind_grad(SS[:, sl], dSS[:, sl], True)
    _h_s(tape, P, "h_scale_s", z_hat, SS[:, :M])
    _h_s(tape, P, "h_mean_s", z_hat, MS[:, :M])
    if tuple(MS.shape[2:]) != (h, w):
        raise ValueError("hyper-synthesis output does not match the latent size (input must be a multiple of 64)")
    Y_hat = E.new((N, M, h, w), dev)
    Y_lik = E.new((N, M, h, w), dev)
    if need:
        dYh = E.zeros(Y_hat.shape, dev)
        tape.bind_grad(Y_hat, dYh, True)
        for i in range(num_slices):
            tape.bind_grad(Y_hat[:, i * sc_:(i + 1) * sc_], dYh[:, i * sc_:(i + 1) * sc_], True)
    mus, scs = [], []
    if bucket_marks is not None:
        bucket_marks[1] = len(tape.bw)   # => slice-chain gradients complete
    # ---- channel-conditional slice loop (cnn.py:161-180)
    for i in range(num_slices):
        k = min(i, max_support)
        ch = slice(i * sc_, (i + 1) * sc_)
        ms, ss = MS[:, :M + sc_ * k], SS[:, :M + sc_ * k]
        mu, sc = _chain_pair(tape, P, f"cc_mean_transforms.{i}", f"cc_scale_transforms.{i}", VT(ms), VT(ss))
        LS = E.new((N, M + sc_ * (k + 1), h, w), dev)
        yh_pre = LS[:, M + sc_ * k:]
        if need:
            dLS = E.zeros(LS.shape, dev)
            tape.bind_grad(LS, dLS, True)
            tape.bind_grad(LS[:, :M + sc_ * k], dLS[:, :M + sc_ * k], True)
            tape.bind_grad(yh_pre, dLS[:, M + sc_ * k:], True)
        _copy_op(tape, ms, LS[:, :M + sc_ * k])
        E.gc_likelihood_ste(tape, y[:, ch], mu, sc, None if noise_y is None else noise_y[:, ch], Y_lik[:, ch], yh_pre)
        _chain(tape, P, f"lrp_transforms.{i}", VT(LS), out=Y_hat[:, ch], lrp_aux=yh_pre)
        if i < max_support:
            sl = slice(M + sc_ * i, M + sc_ * (i + 1))
            _copy_op(tape, Y_hat[:, ch], MS[:, sl])
            _copy_op(tape, Y_hat[:, ch], SS[:, sl])
        if keep is not None:
            mus.append(mu)
            scs.append(sc)
    if bucket_marks is not None:
        bucket_marks[0] = len(tape.bw)   # => g_s gradients complete
    # ---- g_s (cnn.py:42-52)
    t = E.attention_gate(tape, Y_hat, P, "g_s.0", 8, 4, 2)
    t = E.conv2d(tape, VT(t), P["g_s.1.weight"], P["g_s.1.bias"], stride=2, pad=2, transposed=True, output_padding=1)
    t = E.gdn(tape, t, P["g_s.2.beta"], P["g_s.2.gamma"], True)
    t = E.conv2d(tape, VT(t), P["g_s.3.weight"], P["g_s.3.bias"], stride=2, pad=2, transposed=True, output_padding=1)
    t = E.gdn(tape, t, P["g_s.4.beta"], P["g_s.4.gamma"], True)
    t = E.attention_gate(tape, t, P, "g_s.5", 8, 8, 4)
    t = E.conv2d(tape, VT(t), P["g_s.6.weight"], P["g_s.6.bias"], stride=2, pad=2, transposed=True, output_padding=1)
    t = E.gdn(tape, t, P["g_s.7.beta"], P["g_s.7.gamma"], True)
    x_hat = E.conv2d(tape, VT(t), P["g_s.8.weight"], P["g_s.8.bias"], stride=2, pad=2, transposed=True,
                     output_padding=1)
    if need:
        # runs FIRST in backward: hand the seeded d(y_likelihoods) to the per-slice consumers
        def split():
            g = tape.grad_of(Y_lik)
            if g is not None:
                for i in range(num_slices):
                    tape.bind_grad(Y_lik[:, i * sc_:(i + 1) * sc_], g[:, i * sc_:(i + 1) * sc_], True)
        tape.bw.append(split)
    if keep is not None:
        keep.update(y=y, z=z, z_hat=z_hat, y_hat=Y_hat, mu=torch.cat(mus, 1), scale=torch.cat(scs, 1),
                    lat_means=MS[:, :M], lat_scales=SS[:, :M])
    return x_hat, Y_lik, z_lik


class WACNN(CompressionModel):
    """CNN based model (models/cnn.py:23-189): same modules, names and defaults as the reference."""

    def __init__(self, N=192, M=320, **kwargs):
        super().__init__(**kwargs)
        self.num_slices = 10
        self.max_support_slices = 5
        self.g_a = nn.Sequential(
            conv(3, N, kernel_size=5, stride=2), GDN(N),
            conv(N, N, kernel_size=5, stride=2), GDN(N),
            Win_noShift_Attention(dim=N, num_heads=8, window_size=8, shift_size=4),
            conv(N, N, kernel_size=5, stride=2), GDN(N),
            conv(N, M, kernel_size=5, stride=2),
            Win_noShift_Attention(dim=M, num_heads=8, window_size=4, shift_size=2))
        self.g_s = nn.Sequential(
            Win_noShift_Attention(dim=M, num_heads=8, window_size=4, shift_size=2),
            deconv(M, N, kernel_size=5, stride=2), GDN(N, inverse=True),
            deconv(N, N, kernel_size=5, stride=2), GDN(N, inverse=True),
            Win_noShift_Attention(dim=N, num_heads=8, window_size=8, shift_size=4),
            deconv(N, N, kernel_size=5, stride=2), GDN(N, inverse=True),
            deconv(N, 3, kernel_size=5, stride=2))
        self.h_a = nn.Sequential(conv3x3(320, 320), nn.GELU(), conv3x3(320, 288), nn.GELU(),
                                 conv3x3(288, 256, stride=2), nn.GELU(), conv3x3(256, 224), nn.GELU(),
                                 conv3x3(224, 192, stride=2))

        def hs():
            return nn.Sequential(conv3x3(192, 192), nn.GELU(), subpel_conv3x3(192, 224, 2), nn.GELU(),
                                 conv3x3(224, 256), nn.GELU(), subpel_conv3x3(256, 288, 2), nn.GELU(),
                                 conv3x3(288, 320))
        self.h_mean_s = hs()
        self.h_scale_s = hs()

        def cc(extra):
            return nn.ModuleList(nn.Sequential(
                conv(320 + 32 * min(i + extra, 5 + extra), 224, stride=1, kernel_size=3), nn.GELU(),
                conv(224, 176, stride=1, kernel_size=3), nn.GELU(),
                conv(176, 128, stride=1, kernel_size=3), nn.GELU(),
                conv(128, 64, stride=1, kernel_size=3), nn.GELU(),
                conv(64, 32, stride=1, kernel_size=3)) for i in range(10))
        self.cc_mean_transforms = cc(0)
        self.cc_scale_transforms = cc(0)
        self.lrp_transforms = cc(1)
        self.entropy_bottleneck = EntropyBottleneck(N)
        self.gaussian_conditional = GaussianConditional(None)
        self._noise = None

    def inject_noise(self, noise: Optional[dict]):
        """testing hook: {"z": [B,192,h/4,w/4], "y": [B,320,h,w]} U(-1/2,1/2) samples used in train mode"""
        self._noise = noise

    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("WACNN.forward expects [B,3,H,W]")
        names, params = _named(self)
        training = self.training
        dev = x.device
        nz = ny = None
        if training:
            B, _, H, W = x.shape
            if self._noise is not None:
                nz = self._noise["z"].to(dev, torch.float32).contiguous()
                ny = self._noise["y"].to(dev, torch.float32).contiguous()
            else:
                nz = torch.rand((B, 192, H // 64, W // 64), dtype=torch.float32, device=dev) - 0.5
                ny = torch.rand((B, 320, H // 16, W // 16), dtype=torch.float32, device=dev) - 0.5
        ns, ms = self.num_slices, self.max_support_slices

        def runner(tape, xin, *ps):
            return wacnn_forward(tape, dict(zip(names, ps)), xin, nz, ny, ns, ms)

        x_hat, y_lik, z_lik = E.tape_function(runner, [x.contiguous(), *params])
        return {"x_hat": x_hat, "likelihoods": {"y": y_lik, "z": z_lik}}

    @classmethod
    def from_state_dict(cls, state_dict):
        net = cls(192, 320)
        net.load_state_dict(state_dict)
        return net
