"""compressai/ops mirror: ste_round (ops.py:20-34), LowerBound (bound_ops.py:21-65),
NonNegativeParametrizer (parametrizers.py:23-49).  On the hot path these are fused into the GDN and
likelihood kernels; the standalone forms below exist for API parity and call the same HIP kernels."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib as L
from ._lib import check, ptr


class _SteRound(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        xc = x.contiguous()
        y = torch.empty_like(xc)
        q = torch.zeros(3, dtype=torch.float32, device=x.device)
        check(L.lib().icm_ste_round_offset(ptr(xc), ptr(q), ptr(y), 1, 1, xc.numel(), L.stream()), "ste_round")
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, g):
        return g


def ste_round(x: torch.Tensor) -> torch.Tensor:
    """Rounding with identity gradient: (round(x) - x) + x."""
    return _SteRound.apply(x)


class _LowerBoundFn(torch.autograd.Function):
    """max(x, bound); backward passes g where (x >= bound) | (g < 0)."""

    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x, bound)
        return torch.max(x, bound)

    @staticmethod
    def backward(ctx, g):
        x, bound = ctx.saved_tensors
        return ((x >= bound) | (g < 0)) * g, None


class LowerBound(nn.Module):
    bound: torch.Tensor

    def __init__(self, bound: float):
        super().__init__()
        self.register_buffer("bound", torch.Tensor([float(bound)]))

    def forward(self, x):
        return _LowerBoundFn.apply(x, self.bound)


class _NonNegFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bound, pedestal):
        xc = x.contiguous()
        out = torch.empty_like(xc)
        check(L.lib().icm_nonneg_fwd(ptr(xc), ptr(out), xc.numel(), bound, pedestal, L.stream()), "nonneg_fwd")
        ctx.save_for_backward(xc)
        ctx.bound = bound
        return out.view(x.shape)

    @staticmethod
    def backward(ctx, g):
        (xc,) = ctx.saved_tensors
        gc = g.contiguous()
        dx = torch.empty_like(xc)
        check(L.lib().icm_nonneg_bwd(ptr(xc), ptr(gc), ptr(dx), xc.numel(), ctx.bound, 0, L.stream()), "nonneg_bwd")
        return dx.view(g.shape), None, None


class NonNegativeParametrizer(nn.Module):
    pedestal: torch.Tensor

    def __init__(self, minimum: float = 0, reparam_offset: float = 2 ** -18):
        super().__init__()
        self.minimum = float(minimum)
        self.reparam_offset = float(reparam_offset)
        pedestal = self.reparam_offset ** 2
        self.register_buffer("pedestal", torch.Tensor([pedestal]))
        self._bound = (self.minimum + self.reparam_offset ** 2) ** 0.5
        self._pedestal = pedestal
        self.lower_bound = LowerBound(self._bound)

    def init(self, x):
        return torch.sqrt(torch.max(x + self.pedestal, self.pedestal))

    def forward(self, x):
        return _NonNegFn.apply(x, self._bound, self._pedestal)
