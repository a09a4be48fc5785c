"""compressai/entropy_models mirror: the *likelihood* halves of EntropyBottleneck and GaussianConditional
(entropy_models.py:293-489, 525-659) on the fused HIP kernels.  update()/compress()/decompress() (rANS,
CPU, inference-time: SURVEY.md 8 f2) are outside this round's scope and raise NotImplementedError."""
from __future__ import annotations

import ctypes as C
from typing import Any, List, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import engine as E
from ._lib import check, ptr
from .layers import run_module
from .ops import LowerBound


class EntropyModel(nn.Module):
    """entropy_models.py:70-170 (quantize / buffers); the coder proxy is not instantiated."""

    def __init__(self, likelihood_bound: float = 1e-9, entropy_coder: Optional[str] = None,
                 entropy_coder_precision: int = 16):
        super().__init__()
        self.entropy_coder_precision = int(entropy_coder_precision)
        self.use_likelihood_bound = likelihood_bound > 0
        self._lik_bound = float(likelihood_bound)
        if self.use_likelihood_bound:
            self.likelihood_lower_bound = LowerBound(likelihood_bound)
        self.register_buffer("_offset", torch.IntTensor())
        self.register_buffer("_quantized_cdf", torch.IntTensor())
        self.register_buffer("_cdf_length", torch.IntTensor())
        self._injected_noise = None

    offset = property(lambda self: self._offset)
    quantized_cdf = property(lambda self: self._quantized_cdf)
    cdf_length = property(lambda self: self._cdf_length)

    def inject_noise(self, noise: Optional[torch.Tensor]):
        """testing hook: use this U(-1/2,1/2) sample instead of drawing one in train mode"""
        self._injected_noise = noise

    def _noise_like(self, x):
        if self._injected_noise is not None:
            return self._injected_noise.to(x.device, torch.float32).contiguous()
        return torch.rand(x.shape, dtype=torch.float32, device=x.device) - 0.5  # RNG plumbing (entropy_models.py:131-135)

    def quantize(self, inputs, mode, means=None):
        if mode not in ("noise", "dequantize", "symbols"):
            raise ValueError(f'Invalid quantization mode: "{mode}"')
        raise NotImplementedError("standalone quantize(): fused into the likelihood kernels on this path")

    def update(self, *a, **k):
        raise NotImplementedError("CDF tables / rANS coding are not part of the training hot path (SURVEY 8 f2)")

    compress = decompress = update


class EntropyBottleneck(EntropyModel):
    """entropy_models.py:293-489.  forward(x) -> (x_tilde, likelihood)."""

    def __init__(self, channels: int, *args: Any, tail_mass: float = 1e-9, init_scale: float = 10,
                 filters: Tuple[int, ...] = (3, 3, 3, 3), **kwargs: Any):
        super().__init__(*args, **kwargs)
        self.channels = int(channels)
        self.filters = tuple(int(f) for f in filters)
        if self.filters != (3, 3, 3, 3):
            raise NotImplementedError("icm EntropyBottleneck: the kernel is specialised for filters=(3,3,3,3)")
        self.init_scale = float(init_scale)
        self.tail_mass = float(tail_mass)
        filters = (1,) + self.filters + (1,)
        scale = self.init_scale ** (1 / (len(self.filters) + 1))
        channels = self.channels
        for i in range(len(self.filters) + 1):
            init = np.log(np.expm1(1 / scale / filters[i + 1]))
            matrix = torch.Tensor(channels, filters[i + 1], filters[i])
            matrix.data.fill_(init)
            self.register_parameter(f"_matrix{i:d}", nn.Parameter(matrix))
            bias = torch.Tensor(channels, filters[i + 1], 1)
            nn.init.uniform_(bias, -0.5, 0.5)
            self.register_parameter(f"_bias{i:d}", nn.Parameter(bias))
            if i < len(self.filters):
                factor = torch.Tensor(channels, filters[i + 1], 1)
                nn.init.zeros_(factor)
                self.register_parameter(f"_factor{i:d}", nn.Parameter(factor))
        self.quantiles = nn.Parameter(torch.Tensor(channels, 1, 3))
        init = torch.Tensor([-self.init_scale, 0, self.init_scale])
        self.quantiles.data = init.repeat(self.quantiles.size(0), 1, 1)
        target = np.log(2 / self.tail_mass - 1)
        self._target = float(target)
        self.register_buffer("target", torch.Tensor([-target, 0, target]))

    def _get_medians(self):
        return self.quantiles[:, :, 1:2]

    def forward(self, x, training: Optional[bool] = None):
        if training is None:
            training = self.training
        noise = self._noise_like(x) if training else None
        lb = self._lik_bound if self.use_likelihood_bound else 0.0

        def f(tape, P, t):
            zt, lik = E.eb_likelihood(tape, t, {"e." + k: p for k, p in P.items()}, "e", noise, lb, want_zt=True)
            return (zt, lik)
        return run_module(self, f, x.contiguous())

    def loss(self):
        """aux loss (entropy_models.py:395-398): gradient flows to ``quantiles`` only."""
        return _EbAux.apply(self, self.quantiles)


class _EbAux(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mod, quantiles):
        P = {"e." + k: p.detach() for k, p in mod.named_parameters()}
        prm = E._eb_params(P, "e")
        loss = torch.empty(1, dtype=torch.float32, device=quantiles.device)
        dq = torch.empty_like(quantiles)
        check(L.lib().icm_eb_aux_loss(C.byref(prm), ptr(loss), ptr(dq), mod.channels, mod._target, L.stream()), "eb_aux")
        ctx.save_for_backward(dq)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dq,) = ctx.saved_tensors
        return None, dq * g


class GaussianConditional(EntropyModel):
    """entropy_models.py:525-659.  forward(inputs, scales, means) -> (outputs, likelihood)."""

    def __init__(self, scale_table: Optional[Union[List, Tuple]], *args: Any, scale_bound: float = 0.11,
                 tail_mass: float = 1e-9, **kwargs: Any):
        super().__init__(*args, **kwargs)
        if not isinstance(scale_table, (type(None), list, tuple)):
            raise ValueError(f'Invalid type for scale_table "{type(scale_table)}"')
        if isinstance(scale_table, (list, tuple)) and len(scale_table) < 1:
            raise ValueError(f'Invalid scale_table length "{len(scale_table)}"')
        if scale_table and (scale_table != sorted(scale_table) or any(s <= 0 for s in scale_table)):
            raise ValueError(f'Invalid scale_table "({scale_table})"')
        self.tail_mass = float(tail_mass)
        if scale_bound is None and scale_table:
            scale_bound = scale_table[0]
        if scale_bound <= 0:
            raise ValueError("Invalid parameters")
        self._scale_bound = float(scale_bound)
        self.lower_bound_scale = LowerBound(scale_bound)
        self.register_buffer("scale_table", torch.Tensor(tuple(float(s) for s in scale_table)) if scale_table
                             else torch.Tensor())
        self.register_buffer("scale_bound", torch.Tensor([float(scale_bound)]) if scale_bound is not None else None)

    def forward(self, inputs, scales, means=None, training: Optional[bool] = None):
        if training is None:
            training = self.training
        if means is None:
            means = torch.zeros_like(inputs)
        noise = self._noise_like(inputs) if training else None
        lb = self._lik_bound if self.use_likelihood_bound else 0.0
        sb = self._scale_bound

        def f(tape, y, sc, mu):
            lik = E.new(y)
            E.gc_likelihood_ste(tape, y, mu, sc, noise, lik, None, None, sb, lb)
            return (lik,)
        (lik,) = E.tape_function(f, [inputs.contiguous(), scales.contiguous(), means.contiguous()])
        with torch.no_grad():  # the quantised values themselves are plumbing (WACNN.forward discards them, cnn.py:171)
            outputs = inputs + noise if training else torch.round(inputs - means) + means
        return outputs, lik
