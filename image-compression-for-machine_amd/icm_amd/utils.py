"""Inference-side helpers around the codec: the padding / cropping / metrics of the reference's evaluation loop
(compressai/utils/eval_model/__main__.py:78-80,97-139,143-225) on the HIP path (``icm_pad2d``)."""
from __future__ import annotations

import math
import time
from typing import Dict, Tuple

import torch

from . import _lib as L
from ._lib import check, ptr


def pad2d(x: torch.Tensor, left: int, right: int, top: int, bottom: int, value: float = 0.0) -> torch.Tensor:
    """F.pad(x, (left, right, top, bottom), "constant", value); negative amounts crop (eval_model/__main__.py:107-117,129)"""
    if x.dim() != 4:
        raise ValueError("pad2d expects [N,C,H,W]")
    xc = x.to(torch.float32).contiguous()
    N, Cc, H, W = xc.shape
    OH, OW = H + top + bottom, W + left + right
    if OH <= 0 or OW <= 0:
        raise ValueError("pad2d: empty result")
    out = torch.empty((N, Cc, OH, OW), dtype=torch.float32, device=x.device)
    check(L.lib().icm_pad2d(ptr(xc), N, Cc, H, W, ptr(out), OH, OW, top, left, float(value), L.stream()), "pad2d")
    return out


def pad_to_multiple(x: torch.Tensor, p: int = 64) -> Tuple[torch.Tensor, Tuple[int, int, int, int]]:
    """centre zero-padding to the next multiple of p (64 = six stride-2 stages; eval_model/__main__.py:102-117).
    Returns (x_padded, (left, right, top, bottom))."""
    h, w = x.size(2), x.size(3)
    new_h, new_w = (h + p - 1) // p * p, (w + p - 1) // p * p
    left = (new_w - w) // 2
    right = new_w - w - left
    top = (new_h - h) // 2
    bottom = new_h - h - top
    if (left, right, top, bottom) == (0, 0, 0, 0):
        return x, (0, 0, 0, 0)
    return pad2d(x, left, right, top, bottom), (left, right, top, bottom)


def crop(x: torch.Tensor, pads: Tuple[int, int, int, int]) -> torch.Tensor:
    left, right, top, bottom = pads
    if pads == (0, 0, 0, 0):
        return x
    return pad2d(x, -left, -right, -top, -bottom)


def psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    """eval_model/__main__.py:78-80 (inputs in [0, 1]); the mean is one fused HIP reduction"""
    a, b = a.to(torch.float32).contiguous(), b.to(torch.float32).contiguous()
    out = torch.empty(5, dtype=torch.float32, device=a.device)
    ws = torch.empty(L.REDUCE_WS_FLOATS, dtype=torch.float32, device=a.device)
    one = torch.ones(1, dtype=torch.float32, device=a.device)
    check(L.lib().icm_rd_loss_fwd(ptr(a), ptr(b), a.numel(), ptr(one), 1, ptr(one), 1, 1, 0.0, ptr(out), ptr(ws),
                                  L.stream()), "mse")
    return -10.0 * math.log10(out[1].item())


@torch.no_grad()
def inference(model, x: torch.Tensor, recon=None) -> Dict[str, float]:
    """eval_model/__main__.py:96-139: actual bit-stream size and reconstruction quality of one image x [3,H,W] or
    [1,3,H,W] in [0,1]; ``recon(x_hat)`` (optional) receives the cropped reconstruction (the reference saves it, :130)"""
    if x.dim() == 3:
        x = x.unsqueeze(0)
    xp, pads = pad_to_multiple(x, 64)
    t0 = time.time()
    enc = model.compress(xp)
    torch.cuda.synchronize() if x.is_cuda else None
    t1 = time.time()
    dec = model.decompress(enc["strings"], enc["shape"])
    torch.cuda.synchronize() if x.is_cuda else None
    t2 = time.time()
    x_hat = crop(dec["x_hat"], pads)
    if recon is not None:
        recon(x_hat)
    num_pixels = x.size(0) * x.size(2) * x.size(3)
    bpp = sum(len(s[0]) for s in enc["strings"]) * 8.0 / num_pixels
    return {"psnr": psnr(x, x_hat), "bpp": bpp, "encoding_time": t1 - t0, "decoding_time": t2 - t1}


@torch.no_grad()
def inference_entropy_estimation(model, x: torch.Tensor, recon=None) -> Dict[str, float]:
    """eval_model/__main__.py:143-225: forward pass with estimated rates (no entropy coder)"""
    if x.dim() == 3:
        x = x.unsqueeze(0)
    xp, pads = pad_to_multiple(x, 64)
    t0 = time.time()
    out = model(xp)
    torch.cuda.synchronize() if x.is_cuda else None
    dt = time.time() - t0
    x_hat = crop(out["x_hat"], pads)
    if recon is not None:
        recon(x_hat)
    num_pixels = x.size(0) * x.size(2) * x.size(3)
    bpp = sum((torch.log(l).sum() / (-math.log(2) * num_pixels)).item() for l in out["likelihoods"].values())
    return {"psnr": psnr(x, x_hat), "bpp": bpp, "encoding_time": dt / 2.0, "decoding_time": dt / 2.0}
