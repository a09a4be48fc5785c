"""RateDistortionLoss (train.py:44-76; loss form of train_czigzag.py:63,71) on the fused HIP reductions."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib as L
from ._lib import check, ptr


class _RDFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, x_hat, lik_y, lik_z, lmbda):
        x, x_hat, lik_y, lik_z = (t.contiguous() for t in (x, x_hat, lik_y, lik_z))
        N, _, H, W = x.shape
        out = torch.empty(5, dtype=torch.float32, device=x.device)
        ws = torch.empty(L.REDUCE_WS_FLOATS, dtype=torch.float32, device=x.device)
        check(L.lib().icm_rd_loss_fwd(ptr(x), ptr(x_hat), x.numel(), ptr(lik_y), lik_y.numel(), ptr(lik_z),
                                      lik_z.numel(), N * H * W, lmbda, ptr(out), ptr(ws), L.stream()), "rd_loss_fwd")
        ctx.save_for_backward(x, x_hat, lik_y, lik_z)
        ctx.lmbda, ctx.npix = lmbda, N * H * W
        return out[2], out[0], out[1]

    @staticmethod
    def backward(ctx, g_loss, g_bpp, g_mse):
        x, x_hat, lik_y, lik_z = ctx.saved_tensors
        dxh, dly, dlz = torch.empty_like(x_hat), torch.empty_like(lik_y), torch.empty_like(lik_z)
        check(L.lib().icm_rd_loss_bwd(ptr(x), ptr(x_hat), x.numel(), ptr(lik_y), lik_y.numel(), ptr(lik_z),
                                      lik_z.numel(), ctx.npix, ctx.lmbda, 1.0, ptr(dxh), ptr(dly), ptr(dlz),
                                      L.stream()), "rd_loss_bwd")
        g = g_loss if g_loss is not None else torch.ones((), device=x.device)
        return None, dxh * g, dly * g, dlz * g, None


class RateDistortionLoss(nn.Module):
    """loss = lmbda * 255^2 * mse + bpp; returns {"loss","bpp_loss","mse_loss"} like the reference."""

    def __init__(self, lmbda=1e-2):
        super().__init__()
        self.lmbda = float(lmbda)

    def forward(self, output, target):
        lik = output["likelihoods"]
        loss, bpp, mse = _RDFn.apply(target, output["x_hat"], lik["y"], lik["z"], self.lmbda)
        return {"loss": loss, "bpp_loss": bpp.detach(), "mse_loss": mse.detach()}
