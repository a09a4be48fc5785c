"""CPU ORACLE for the `stf` (SymmetricalTransFormer) hot path -- TEST INFRASTRUCTURE ONLY.

From-scratch CPU restatement (PyTorch-CPU, IEEE f32) of ``SymmetricalTransFormer.forward``
(compressai/models/stf.py:582-645) on a plain state-dict.  Like ``wacnn_oracle`` it is the *checker*:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

Parity pin: ``tests/golden/make_golden.py`` (``gen_stf*``) runs the real reference modules
(SwinTransformerBlock, PatchMerging, PatchSplit, the full model in eval mode and in train mode with
injected quantisation noise and DropPath masks) against the functions below and writes the committed
fixtures ``tests/golden/stf_*.npz``.  The reference has no tests of its own for this path.

The hyperprior / slice loop is shared with the cnn model (``wacnn_oracle.hyper_slices``).
Citations are relative to /root/reference.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import wacnn_oracle as O

Tensor = torch.Tensor

EMBED_DIM = 48                 # stf.py:322
DEPTHS = (2, 2, 6, 2)          # stf.py:323
HEADS = (3, 6, 12, 24)         # stf.py:324
WINDOW = 4                     # stf.py:325
NUM_SLICES = 12                # stf.py:326
MAX_SUPPORT = 6                # stf.py:347 (num_slices // 2)
DROP_PATH_RATE = 0.2           # stf.py:332
LN_EPS = 1e-5                  # nn.LayerNorm default


def drop_path_rates():
    """stf.py:357: linspace(0, drop_path_rate, sum(depths)) -> per-block rates of the analysis layers;
    the synthesis layers index the same list with the reversed depths (stf.py:382-398)."""
    dpr = [v.item() for v in torch.linspace(0, DROP_PATH_RATE, sum(DEPTHS))]
    ana, syn = {}, {}
    o = 0
    for i, d in enumerate(DEPTHS):
        for j in range(d):
            ana[f"layers.{i}.blocks.{j}"] = dpr[o + j]
        o += d
    o = 0
    for i, d in enumerate(DEPTHS[::-1]):
        for j in range(d):
            syn[f"syn_layers.{i}.blocks.{j}"] = dpr[o + j]
        o += d
    return {**ana, **syn}


def layer_norm(x: Tensor, sd, p: str) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], LN_EPS)


def window_msa(t: Tensor, sd, p: str, heads: int, ws: int, shift: int) -> Tensor:
    """Shifted-window attention on a padded NHWC map: roll, partition, WindowAttention, reverse, roll back.
    stf.py:54-121 (WindowAttention), :165-189 (block plumbing), :281-297 (mask)."""
    B, H, W, C = t.shape
    hd = C // heads
    if shift > 0:
        t = torch.roll(t, shifts=(-shift, -shift), dims=(1, 2))
    win = t.reshape(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)
    Bn, N, _ = win.shape
    qkv = F.linear(win, sd[p + ".qkv.weight"], sd[p + ".qkv.bias"])
    qkv = qkv.reshape(Bn, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (hd ** -0.5), qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    idx = O.relative_position_index(ws).reshape(-1)
    bias = sd[p + ".relative_position_bias_table"][idx].reshape(N, N, heads).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if shift > 0:
        mask = O.shift_mask(H, W, ws, shift)
        nW = mask.shape[0]
        attn = attn.reshape(Bn // nW, nW, heads, N, N) + mask[None, :, None]
        attn = attn.reshape(-1, heads, N, N)
    attn = torch.softmax(attn, dim=-1)
    o = (attn @ v).transpose(1, 2).reshape(Bn, N, C)
    o = F.linear(o, sd[p + ".proj.weight"], sd[p + ".proj.bias"])
    o = o.reshape(B, H // ws, W // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, C)
    if shift > 0:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    return o


def swin_block(x: Tensor, H: int, W: int, sd, p: str, heads: int, ws: int, shift: int,
               dp: Optional[Tensor] = None) -> Tensor:
    """SwinTransformerBlock.forward (stf.py:149-193). x: [B, H*W, C].
    dp: None (eval / rate 0) or [2, B] DropPath scales (0 or 1/keep) for the attention and MLP branches."""
    B, L, C = x.shape
    shortcut = x
    t = layer_norm(x, sd, p + ".norm1").view(B, H, W, C)
    pr, pb = (ws - W % ws) % ws, (ws - H % ws) % ws
    t = F.pad(t, (0, 0, 0, pr, 0, pb))
    t = window_msa(t, sd, p + ".attn", heads, ws, shift)
    if pr > 0 or pb > 0:
        t = t[:, :H, :W, :].contiguous()
    t = t.reshape(B, H * W, C)
    if dp is not None:
        t = t * dp[0].view(B, 1, 1)
    x = shortcut + t
    m = layer_norm(x, sd, p + ".norm2")
    m = F.linear(m, sd[p + ".mlp.fc1.weight"], sd[p + ".mlp.fc1.bias"])
    m = F.gelu(m)
    m = F.linear(m, sd[p + ".mlp.fc2.weight"], sd[p + ".mlp.fc2.bias"])
    if dp is not None:
        m = m * dp[1].view(B, 1, 1)
    return x + m


def patch_merging(x: Tensor, H: int, W: int, sd, p: str) -> Tensor:
    """PatchMerging.forward (stf.py:203-233)."""
    B, L, C = x.shape
    x = x.view(B, H, W, C)
    if H % 2 == 1 or W % 2 == 1:
        x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
    x = x.view(B, -1, 4 * C)
    x = layer_norm(x, sd, p + ".norm")
    return F.linear(x, sd[p + ".reduction.weight"])


def patch_split(x: Tensor, H: int, W: int, sd, p: str) -> Tensor:
    """PatchSplit.forward (stf.py:249-259)."""
    B, L, C = x.shape
    x = layer_norm(x, sd, p + ".norm")
    x = F.linear(x, sd[p + ".reduction.weight"])
    x = x.permute(0, 2, 1).contiguous().view(B, 2 * C, H, W)
    x = F.pixel_shuffle(x, 2)
    return x.permute(0, 2, 3, 1).contiguous().view(B, 4 * L, -1)


def basic_layer(x, H, W, sd, p, depth, heads, ws, down: Optional[str], drops):
    """BasicLayer.forward (stf.py:271-313)."""
    for j in range(depth):
        shift = 0 if j % 2 == 0 else ws // 2
        key = f"{p}.blocks.{j}"
        x = swin_block(x, H, W, sd, key, heads, ws, shift, None if drops is None else drops.get(key))
    if down == "merge":
        return patch_merging(x, H, W, sd, p + ".downsample"), (H + 1) // 2, (W + 1) // 2
    if down == "split":
        return patch_split(x, H, W, sd, p + ".downsample"), H * 2, W * 2
    return x, H, W


def patch_embed(x: Tensor, sd) -> Tensor:
    """PatchEmbed.forward with patch_size 2 and patch_norm (stf.py:331-351) -> NCHW."""
    ps = 2
    _, _, H, W = x.shape
    if W % ps != 0:
        x = F.pad(x, (0, ps - W % ps))
    if H % ps != 0:
        x = F.pad(x, (0, 0, 0, ps - H % ps))
    x = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=ps)
    B, C, Wh, Ww = x.shape
    t = layer_norm(x.flatten(2).transpose(1, 2), sd, "patch_embed.norm")
    return t.transpose(1, 2).view(-1, C, Wh, Ww)


def analysis(x: Tensor, sd, drops=None):
    """patch_embed + layers (stf.py:584-595) -> y [B, 384, H/16, W/16]."""
    x = patch_embed(x, sd)
    Wh, Ww = x.shape[2], x.shape[3]
    t = x.flatten(2).transpose(1, 2)
    for i in range(4):
        t, Wh, Ww = basic_layer(t, Wh, Ww, sd, f"layers.{i}", DEPTHS[i], HEADS[i], WINDOW,
                                "merge" if i < 3 else None, drops)
    C = EMBED_DIM * 8
    return t.view(-1, Wh, Ww, C).permute(0, 3, 1, 2).contiguous()


def synthesis(y_hat: Tensor, sd, drops=None) -> Tensor:
    """syn_layers + end_conv (stf.py:639-644)."""
    B, C, Wh, Ww = y_hat.shape
    t = y_hat.permute(0, 2, 3, 1).contiguous().view(-1, Wh * Ww, C)
    depths, heads = DEPTHS[::-1], HEADS[::-1]
    for i in range(4):
        t, Wh, Ww = basic_layer(t, Wh, Ww, sd, f"syn_layers.{i}", depths[i], heads[i], WINDOW,
                                "split" if i < 3 else None, drops)
    u = t.view(-1, Wh, Ww, EMBED_DIM).permute(0, 3, 1, 2).contiguous()
    u = F.conv2d(u, sd["end_conv.0.weight"], sd["end_conv.0.bias"], padding=2)
    u = F.pixel_shuffle(u, 2)
    return F.conv2d(u, sd["end_conv.2.weight"], sd["end_conv.2.bias"], padding=1)


def stf_forward(sd: Dict[str, Tensor], x: Tensor, noise: Optional[Dict[str, Tensor]] = None,
                drops: Optional[Dict[str, Tensor]] = None, keep: bool = False,
                round_override: Optional[Dict[str, Tensor]] = None) -> Dict:
    """SymmetricalTransFormer.forward (stf.py:582-645).

    noise: None -> eval quantisation; else {"z": [B,192,h/4,w/4], "y": [B,384,h,w]} U(-1/2,1/2) samples.
    drops: None -> no stochastic depth (eval); else {"<layer>.blocks.<j>": [2,B] scales} as drawn by
    timm's DropPath in train mode (mask / keep_prob)."""
    y = analysis(x, sd, drops)
    y_hat, y_lik, z_lik, dbg = O.hyper_slices(y, sd, noise, NUM_SLICES, MAX_SUPPORT, round_override)
    x_hat = synthesis(y_hat, sd, drops)
    out = {"x_hat": x_hat, "likelihoods": {"y": y_lik, "z": z_lik}}
    if keep:
        out["_dbg"] = dbg
    return out
