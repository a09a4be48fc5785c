"""compressai/layers mirror (gdn.py, layers.py, win_attention.py) on the HIP engine.

Constructor signatures, parameter/buffer names and shapes are those of the reference so state-dicts are
interchangeable; ``forward`` runs the kernels of libicm_hip.so through ``engine`` (no ATen compute)."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import engine as E
from .engine import VT
from .ops import NonNegativeParametrizer

__all__ = ["GDN", "conv3x3", "subpel_conv3x3", "conv1x1", "Win_noShift_Attention", "WinBasedAttention",
           "WindowAttention", "Conv2d", "ConvTranspose2d", "conv", "deconv"]


def _named(mod: nn.Module):
    """(names, tensors) of the module's parameters, cached on the module"""
    c = getattr(mod, "_icm_params", None)
    if c is None or len(c[0]) != sum(1 for _ in mod.parameters()):
        items = list(mod.named_parameters())
        c = ([n for n, _ in items], [p for _, p in items])
        object.__setattr__(mod, "_icm_params", c)
    return c


def run_module(mod: nn.Module, fn, *inputs):
    """fn(tape, P, *inputs) -> tuple of tensors, executed as one autograd node over (inputs, params)."""
    names, params = _named(mod)
    n_in = len(inputs)

    def runner(tape, *ts):
        P = dict(zip(names, ts[n_in:]))
        return fn(tape, P, *ts[:n_in])

    return E.tape_function(runner, [*inputs, *params])


def _square(mod, *names):
    """the HIP kernels take one stride / padding for both axes: refuse anything else instead of silently using [0]"""
    out = []
    for n in names:
        v = getattr(mod, n)
        if isinstance(v, (tuple, list)):
            if len(set(v)) != 1:
                raise NotImplementedError(f"icm {type(mod).__name__}: asymmetric {n} {tuple(v)}")
            v = v[0]
        out.append(int(v))
    return out


class Conv2d(nn.Conv2d):
    """nn.Conv2d parameter holder whose forward is the implicit-GEMM HIP kernel (models/utils.py:114-121)."""

    def forward(self, x):
        if self.padding_mode != "zeros" or self.groups != 1 or tuple(self.dilation) != (1, 1):
            raise NotImplementedError("icm Conv2d: only dense zero-padded convolutions")
        if self.kernel_size[0] != self.kernel_size[1]:
            raise NotImplementedError("icm Conv2d: square kernels only")
        s, p = _square(self, "stride", "padding")
        return run_module(self, lambda tape, P, t: (E.conv2d(tape, VT(t), P["weight"], P.get("bias"), stride=s, pad=p),),
                          x.contiguous())[0]


class ConvTranspose2d(nn.ConvTranspose2d):
    """nn.ConvTranspose2d parameter holder on the HIP kernel (models/utils.py:124-132)."""

    def forward(self, x):
        if self.padding_mode != "zeros" or self.groups != 1 or tuple(self.dilation) != (1, 1):
            raise NotImplementedError("icm ConvTranspose2d: only dense zero-padded transposed convolutions")
        if self.kernel_size[0] != self.kernel_size[1]:
            raise NotImplementedError("icm ConvTranspose2d: square kernels only")
        s, p, op = _square(self, "stride", "padding", "output_padding")
        return run_module(self, lambda tape, P, t: (E.conv2d(tape, VT(t), P["weight"], P.get("bias"), stride=s, pad=p,
                                                             transposed=True, output_padding=op),), x.contiguous())[0]


def conv(in_channels, out_channels, kernel_size=5, stride=2):
    return Conv2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=kernel_size // 2)


def deconv(in_channels, out_channels, kernel_size=5, stride=2):
    return ConvTranspose2d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                           output_padding=stride - 1, padding=kernel_size // 2)


def conv3x3(in_ch: int, out_ch: int, stride: int = 1) -> nn.Module:
    return Conv2d(in_ch, out_ch, kernel_size=3, stride=stride, padding=1)


class PixelShuffle2(nn.PixelShuffle):
    """standalone nn.PixelShuffle stand-in; fused into the preceding conv's store on the model path"""


def subpel_conv3x3(in_ch: int, out_ch: int, r: int = 1) -> nn.Sequential:
    return nn.Sequential(Conv2d(in_ch, out_ch * r ** 2, kernel_size=3, padding=1), PixelShuffle2(r))


def conv1x1(in_ch: int, out_ch: int, stride: int = 1) -> nn.Module:
    return Conv2d(in_ch, out_ch, kernel_size=1, stride=stride)


class GDN(nn.Module):
    """Generalized Divisive Normalization (layers/gdn.py:39-75)."""

    def __init__(self, in_channels: int, inverse: bool = False, beta_min: float = 1e-6, gamma_init: float = 0.1):
        super().__init__()
        beta_min = float(beta_min)
        gamma_init = float(gamma_init)
        self.inverse = bool(inverse)
        self.beta_min = beta_min
        self.beta_reparam = NonNegativeParametrizer(minimum=beta_min)
        beta = torch.ones(in_channels)
        self.beta = nn.Parameter(self.beta_reparam.init(beta))
        self.gamma_reparam = NonNegativeParametrizer()
        gamma = gamma_init * torch.eye(in_channels)
        self.gamma = nn.Parameter(self.gamma_reparam.init(gamma))

    def forward(self, x):
        inv, bm = self.inverse, self.beta_min
        return run_module(self, lambda tape, P, t: (E.gdn(tape, t, P["beta"], P["gamma"], inv, bm),), x.contiguous())[0]


class WindowAttention(nn.Module):
    """Parameter holder of the W-MSA block (layers/win_attention.py:37-82)."""

    def __init__(self, dim=192, window_size=(8, 8), num_heads=8, qkv_bias=True, qk_scale=None, attn_drop=0.,
                 proj_drop=0.):
        super().__init__()
        if attn_drop != 0.0 or proj_drop != 0.0 or qk_scale is not None:
            raise NotImplementedError("icm WindowAttention: dropout / qk_scale override are unused by the reference")
        self.dim, self.window_size, self.num_heads = dim, window_size, num_heads
        ws = window_size[0]
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) * (2 * ws - 1), num_heads))
        ch = torch.arange(ws)
        coords = torch.stack(torch.meshgrid(ch, ch, indexing="ij")).flatten(1)
        rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += ws - 1
        rel[:, :, 1] += ws - 1
        rel[:, :, 0] *= 2 * ws - 1
        self.register_buffer("relative_position_index", rel.sum(-1))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02, a=-2.0, b=2.0)


class WinBasedAttention(nn.Module):
    """layers/win_attention.py:118-207 (shift-window attention + shortcut, no LayerNorm / MLP)."""

    def __init__(self, dim=192, num_heads=8, window_size=8, shift_size=0, qkv_bias=True, qk_scale=None, drop=0.,
                 attn_drop=0., drop_path=0.):
        super().__init__()
        self.dim, self.num_heads, self.window_size, self.shift_size = dim, num_heads, window_size, shift_size
        assert 0 <= self.shift_size < self.window_size, "shift_size must in 0-window_size"
        if drop_path != 0.0:
            raise NotImplementedError("icm WinBasedAttention: drop_path is unused by the reference")
        self.attn = WindowAttention(dim, window_size=(window_size, window_size), num_heads=num_heads,
                                    qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = nn.Identity()

    def forward(self, x):
        h, ws, sh = self.num_heads, self.window_size, self.shift_size
        return run_module(self, lambda tape, P, t: (E.window_attention(tape, t, {"w." + k: p for k, p in P.items()},
                                                                       "w", h, ws, sh),), x.contiguous())[0]


class Win_noShift_Attention(nn.Module):
    """Window-attention gate a*sigmoid(b)+x (layers/layers.py:45-89)."""

    def __init__(self, dim, num_heads=8, window_size=8, shift_size=0):
        super().__init__()
        N = dim
        self.num_heads, self.window_size, self.shift_size = num_heads, window_size, shift_size

        class ResidualUnit(nn.Module):
            def __init__(self):
                super().__init__()
                self.conv = nn.Sequential(conv1x1(N, N // 2), nn.GELU(), conv3x3(N // 2, N // 2), nn.GELU(),
                                          conv1x1(N // 2, N))
                self.relu = nn.GELU()

            def forward(self, x):
                def f(tape, P, t):
                    # (last=True: this module applies the closing GELU itself, so it wants the pre-activation)
                    v = E.residual_unit(tape, VT(t), {"u." + k: p for k, p in P.items()}, "u", last=True)
                    out = E.new(t)
                    E.check(E.L.lib().icm_gelu_fwd(E.ptr(v.t), E.ptr(out), t.numel(), tape.st), "gelu")
                    if tape.need_grad:
                        def bwd():
                            g = tape.grad_of(out)
                            if g is not None:
                                E.accumulate(tape, v.t, g.contiguous(), v.t)
                        tape.bw.append(bwd)
                    return (out,)
                return run_module(self, f, x.contiguous())[0]

        self.conv_a = nn.Sequential(ResidualUnit(), ResidualUnit(), ResidualUnit())
        self.conv_b = nn.Sequential(
            WinBasedAttention(dim=dim, num_heads=num_heads, window_size=window_size, shift_size=shift_size),
            ResidualUnit(), ResidualUnit(), ResidualUnit(), conv1x1(N, N))

    def forward(self, x):
        h, ws, sh = self.num_heads, self.window_size, self.shift_size
        return run_module(self, lambda tape, P, t: (E.attention_gate(tape, t, {"g." + k: p for k, p in P.items()},
                                                                     "g", h, ws, sh),), x.contiguous())[0]
