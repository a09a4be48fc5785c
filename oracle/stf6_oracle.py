"""CPU ORACLE for the zigzag / Swin-refined stf variant -- TEST INFRASTRUCTURE ONLY.

Restatement of ``SymmetricalTransFormer3.forward`` (compressai/models/stf6.py:764-872, registered as ``stf6``) on a
plain state dict, built from the pieces of oracle/stf_oracle.py (stf6.py:24-383 is line-identical to stf.py's building
blocks) and oracle/zigzag_oracle.py.  Pinned bit for bit against the real class by tests/golden/make_golden_stf6.py.
Only tests/ import this module."""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import Tensor

from . import stf_oracle as S
from . import wacnn_oracle as O
from .weights import STF6_BLOCKS, STF6_SLICES, STF6_SUPPORT

NUMBER = 2                      # stf6.py:656,786: row / column halves
MU_HEADS = 4                    # stf6.py:475
RDEPTHS = S.DEPTHS[::-1]        # stf6.py:445: mu_Swin is built after `depths = depths[::-1]`


def drop_path_rates() -> Dict[str, float]:
    """stf's rates plus the refinement stacks: every mu_Swin[b][i] takes dpr[sum(rdepths[:i]) : ...] (stf6.py:469-484)"""
    out = dict(S.drop_path_rates())
    dpr = [v.item() for v in torch.linspace(0, S.DROP_PATH_RATE, sum(S.DEPTHS))]
    for b in range(STF6_BLOCKS):
        o = 0
        for i, d in enumerate(RDEPTHS):
            for j in range(d):
                out[f"mu_Swin.{b}.{i}.blocks.{j}"] = dpr[o + j]
            o += d
    return out


def zigzag_splits(x: Tensor, num_slices: int) -> Tensor:
    """torch form of oracle/zigzag_oracle.zigzag_splits (differentiable; stf6.py:654-714)"""
    from .zigzag_oracle import zigzag_order
    B, C, H, W = x.shape
    v = x.view(B, num_slices, C // num_slices, NUMBER, H // NUMBER, NUMBER, W // NUMBER)
    return torch.stack([v[:, c, :, h, :, w, :] for (c, h, w) in zigzag_order(num_slices, NUMBER, NUMBER)], 1)


def zigzag_reverse(z: Tensor, num_slices: int) -> Tensor:
    """stf6.py:716-762"""
    from .zigzag_oracle import zigzag_order
    B, N, Cs, Hb, Wb = z.shape
    blocks = {cw: z[:, n] for n, cw in enumerate(zigzag_order(num_slices, NUMBER, NUMBER))}
    rows = []
    for c in range(num_slices):
        hs = [torch.cat([blocks[(c, h, w)] for w in range(NUMBER)], -1) for h in range(NUMBER)]
        rows.append(torch.cat(hs, -2))
    return torch.cat(rows, 1)


def hyper_slices_zigzag(y: Tensor, sd, noise, drops, round_override=None):
    """stf6.py:778-858 -> (y_hat [B,384,H,W], y_lik [B, 24*64, H/2, W/2] in zigzag order, z_lik, dbg)"""
    ro = round_override or {}
    z = O.h_a(y, sd)
    _, z_lik = O.eb_likelihood(z, sd, "entropy_bottleneck", None if noise is None else noise["z"])
    med = sd["entropy_bottleneck.quantiles"][:, :, 1:2].reshape(1, -1, 1, 1)
    z_hat = O.ste_round_as(z - med, ro.get("z")) + med
    lat_scales = O.h_s(z_hat, sd, "h_scale_s")
    lat_means = O.h_s(z_hat, sd, "h_mean_s")
    B, C, H, W = lat_scales.shape
    hb, wb, cs = H // NUMBER, W // NUMBER, 384 // STF6_SLICES
    y_zz = zigzag_splits(y, STF6_SLICES)
    sc_zz = zigzag_splits(lat_scales, STF6_SLICES)
    mu_zz = zigzag_splits(lat_means, STF6_SLICES)
    y_hat_slices, liks, mus, scales = [], [], [], []
    for i in range(STF6_BLOCKS):
        sup = y_hat_slices if STF6_SUPPORT > i else y_hat_slices[i - STF6_SUPPORT:]     # stf6.py:797
        mean_sup = torch.cat([mu_zz[:, i]] + sup, 1)
        mu = O._seq_convs(mean_sup, sd, f"cc_mean_transforms2.{i}", (0, 2, 4, 6, 8))
        scale_sup = torch.cat([sc_zz[:, i]] + sup, 1)
        sc = O._seq_convs(scale_sup, sd, f"cc_scale_transforms2.{i}", (0, 2, 4, 6, 8))
        # Swin refinement of the mean on the block map (stf6.py:806-812)
        t = mu.permute(0, 2, 3, 1).contiguous().view(-1, hb * wb, cs)
        for l in range(4):
            t, _, _ = S.basic_layer(t, hb, wb, sd, f"mu_Swin.{i}.{l}", RDEPTHS[l], MU_HEADS, S.WINDOW, None, drops)
        mu = mu + t.view(-1, hb, wb, cs).permute(0, 3, 1, 2).contiguous()
        ys = y_zz[:, i]
        ry = ro["y"][:, i] if "y" in ro else None
        _, lik = O.gaussian_likelihood(ys, sc, mu, None if noise is None else noise["y"][:, i], round_to=ry)
        liks.append(lik)
        yh = O.ste_round_as(ys - mu, ry) + mu
        lrp = O._seq_convs(torch.cat([mean_sup, yh], 1), sd, f"lrp_transforms2.{i}", (0, 2, 4, 6, 8))
        yh = yh + 0.5 * torch.tanh(lrp)
        y_hat_slices.append(yh)
        mus.append(mu)
        scales.append(sc)
    y_hat_zz = torch.cat(y_hat_slices, 1).view(-1, STF6_BLOCKS, cs, hb, wb)
    y_hat = zigzag_reverse(y_hat_zz, STF6_SLICES)
    y_lik = torch.cat(liks, 1)
    dbg = {"y": y, "z": z, "z_hat": z_hat, "y_hat": y_hat, "y_zz": y_zz, "mu": torch.stack(mus, 1),
           "scale": torch.stack(scales, 1), "y_hat_zz": y_hat_zz}
    return y_hat, y_lik, z_lik, dbg


def stf6_forward(sd: Dict[str, Tensor], x: Tensor, noise: Optional[Dict[str, Tensor]] = None,
                 drops: Optional[Dict[str, Tensor]] = None, keep: bool = False,
                 round_override: Optional[Dict[str, Tensor]] = None) -> Dict:
    """SymmetricalTransFormer3.forward (stf6.py:764-872).
    noise: None -> eval quantisation; else {"z": [B,192,h/4,w/4], "y": [B,24,64,h/2,w/2]} (zigzag block order).
    drops: {"<stack>.blocks.<j>": [2,B]} DropPath scales for layers / syn_layers / mu_Swin blocks."""
    y = S.analysis(x, sd, drops)
    y_hat, y_lik, z_lik, dbg = hyper_slices_zigzag(y, sd, noise, drops, round_override)
    x_hat = S.synthesis(y_hat, sd, drops)
    out = {"x_hat": x_hat, "likelihoods": {"y": y_lik, "z": z_lik}}
    if keep:
        out["_dbg"] = dbg
    return out
