"""Weights-by-formula for the `cnn` (WACNN) state-dict -- TEST INFRASTRUCTURE ONLY.

Every entry of the 585-key reference state-dict (compressai/models/cnn.py:26-130) is a
deterministic function of (key, flat index), computed with a counter-based hash (splitmix64),
so that bit-identical weights exist wherever the tests run without shipping 300 MB and
without depending on any RNG implementation.  ``wacnn_spec()`` restates the key/shape list;
``tests/golden/make_golden.py`` checks it against the real reference ``state_dict()``.

The scales follow PyTorch's default Conv/Linear init bounds (1/sqrt(fan_in)) so activations
stay in the range the reference sees, except where noted: GDN gamma gets entries on both
sides of its lower bound, EntropyBottleneck factors are non-zero, and ``gain`` entries widen
the latents so that quantisation is non-trivial (y spans several integers).
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict

import numpy as np
import torch

PEDESTAL = 2.0 ** -36


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def uniform01(key: str, n: int, salt: int = 0) -> np.ndarray:
    """n doubles in [0,1): splitmix64(crc32(key)<<32 | salt<<28 ^ index) >> 11 / 2^53."""
    with np.errstate(over="ignore"):
        base = np.uint64((zlib.crc32(key.encode()) << 32) ^ (salt << 60 >> 4))
        idx = np.arange(n, dtype=np.uint64) + base
        h = _splitmix64(idx)
    return (h >> np.uint64(11)).astype(np.float64) / float(1 << 53)


def _u(key, shape, lo, hi, salt=0):
    n = int(np.prod(shape)) if len(shape) else 1
    v = lo + (hi - lo) * uniform01(key, n, salt)
    return torch.from_numpy(v.astype(np.float32)).reshape(shape)


# ----------------------------------------------------------------------------- spec
def _gate_spec(spec, p, dim, ws):
    def ru(q):
        spec[q + ".conv.0.weight"] = (dim // 2, dim, 1, 1)
        spec[q + ".conv.0.bias"] = (dim // 2,)
        spec[q + ".conv.2.weight"] = (dim // 2, dim // 2, 3, 3)
        spec[q + ".conv.2.bias"] = (dim // 2,)
        spec[q + ".conv.4.weight"] = (dim, dim // 2, 1, 1)
        spec[q + ".conv.4.bias"] = (dim,)
    for i in range(3):
        ru(f"{p}.conv_a.{i}")
    a = p + ".conv_b.0.attn"
    spec[a + ".relative_position_bias_table"] = ((2 * ws - 1) ** 2, 8)
    spec[a + ".relative_position_index"] = ("int64", (ws * ws, ws * ws))
    spec[a + ".qkv.weight"] = (3 * dim, dim)
    spec[a + ".qkv.bias"] = (3 * dim,)
    spec[a + ".proj.weight"] = (dim, dim)
    spec[a + ".proj.bias"] = (dim,)
    for i in (1, 2, 3):
        ru(f"{p}.conv_b.{i}")
    spec[p + ".conv_b.4.weight"] = (dim, dim, 1, 1)
    spec[p + ".conv_b.4.bias"] = (dim,)


def _gdn_spec(spec, p, C):
    spec[p + ".beta"] = (C,)
    spec[p + ".gamma"] = (C, C)
    spec[p + ".beta_reparam.pedestal"] = (1,)
    spec[p + ".beta_reparam.lower_bound.bound"] = (1,)
    spec[p + ".gamma_reparam.pedestal"] = (1,)
    spec[p + ".gamma_reparam.lower_bound.bound"] = (1,)


def _conv_spec(spec, p, co, ci, k, transposed=False):
    spec[p + ".weight"] = (ci, co, k, k) if transposed else (co, ci, k, k)
    spec[p + ".bias"] = (co,)


def wacnn_spec(N: int = 192, M: int = 320) -> "OrderedDict[str, tuple]":
    """Ordered key -> shape (or ("int64"/"int32", shape)) of WACNN.state_dict(). cnn.py:26-130."""
    s: "OrderedDict[str, tuple]" = OrderedDict()
    _conv_spec(s, "g_a.0", N, 3, 5); _gdn_spec(s, "g_a.1", N)
    _conv_spec(s, "g_a.2", N, N, 5); _gdn_spec(s, "g_a.3", N)
    _gate_spec(s, "g_a.4", N, 8)
    _conv_spec(s, "g_a.5", N, N, 5); _gdn_spec(s, "g_a.6", N)
    _conv_spec(s, "g_a.7", M, N, 5)
    _gate_spec(s, "g_a.8", M, 4)
    _gate_spec(s, "g_s.0", M, 4)
    _conv_spec(s, "g_s.1", N, M, 5, True); _gdn_spec(s, "g_s.2", N)
    _conv_spec(s, "g_s.3", N, N, 5, True); _gdn_spec(s, "g_s.4", N)
    _gate_spec(s, "g_s.5", N, 8)
    _conv_spec(s, "g_s.6", N, N, 5, True); _gdn_spec(s, "g_s.7", N)
    _conv_spec(s, "g_s.8", 3, N, 5, True)
    for i, (ci, co) in zip((0, 2, 4, 6, 8), ((320, 320), (320, 288), (288, 256), (256, 224), (224, 192))):
        _conv_spec(s, f"h_a.{i}", co, ci, 3)
    for h in ("h_mean_s", "h_scale_s"):
        _conv_spec(s, f"{h}.0", 192, 192, 3)
        _conv_spec(s, f"{h}.2.0", 224 * 4, 192, 3)
        _conv_spec(s, f"{h}.4", 256, 224, 3)
        _conv_spec(s, f"{h}.6.0", 288 * 4, 256, 3)
        _conv_spec(s, f"{h}.8", 320, 288, 3)
    chain = (224, 176, 128, 64, 32)
    for fam, extra in (("cc_mean_transforms", 0), ("cc_scale_transforms", 0), ("lrp_transforms", 1)):
        for i in range(10):
            ci = 320 + 32 * min(i + extra, 5 + extra)
            for j, co in zip((0, 2, 4, 6, 8), chain):
                _conv_spec(s, f"{fam}.{i}.{j}", co, ci, 3)
                ci = co
    e = "entropy_bottleneck"
    filt = (1, 3, 3, 3, 3, 1)
    for i in range(5):
        s[f"{e}._matrix{i}"] = (N, filt[i + 1], filt[i])
        s[f"{e}._bias{i}"] = (N, filt[i + 1], 1)
        if i < 4:
            s[f"{e}._factor{i}"] = (N, filt[i + 1], 1)
    s[e + ".quantiles"] = (N, 1, 3)
    s[e + "._offset"] = ("int32", (0,))
    s[e + "._quantized_cdf"] = ("int32", (0,))
    s[e + "._cdf_length"] = ("int32", (0,))
    s[e + ".target"] = (3,)
    s[e + ".likelihood_lower_bound.bound"] = (1,)
    g = "gaussian_conditional"
    s[g + "._offset"] = ("int32", (0,))
    s[g + "._quantized_cdf"] = ("int32", (0,))
    s[g + "._cdf_length"] = ("int32", (0,))
    s[g + ".scale_table"] = (0,)
    s[g + ".scale_bound"] = (1,)
    s[g + ".likelihood_lower_bound.bound"] = (1,)
    s[g + ".lower_bound_scale.bound"] = (1,)
    return s


def _rel_index(ws):
    ch = torch.arange(ws)
    coords = torch.stack(torch.meshgrid(ch, ch, indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


# gains that spread the latents over a few integers (default-scale init gives |y| << 0.5)
GAINS = {"g_a.7.weight": 20.0, "h_a.8.weight": 80.0}
for _i in range(10):
    GAINS[f"cc_scale_transforms.{_i}.8.weight"] = 12.0   # sigma spans both sides of the 0.11 bound
    GAINS[f"cc_mean_transforms.{_i}.8.weight"] = 8.0
    GAINS[f"lrp_transforms.{_i}.8.weight"] = 8.0


def make_wacnn_state_dict(salt: int = 0) -> "OrderedDict[str, torch.Tensor]":
    return _make(wacnn_spec(), GAINS, salt)


def _make(spec, gains, salt: int = 0) -> "OrderedDict[str, torch.Tensor]":
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for key, shp in spec.items():
        if isinstance(shp[0], str):
            dt = torch.int64 if shp[0] == "int64" else torch.int32
            if key.endswith("relative_position_index"):
                sd[key] = _rel_index(int(math.isqrt(shp[1][0])))
            else:
                sd[key] = torch.zeros(shp[1], dtype=dt)
            continue
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "pedestal":
            t = torch.tensor([PEDESTAL], dtype=torch.float32)
        elif leaf == "bound":
            if "beta_reparam" in key:
                t = torch.tensor([(1e-6 + PEDESTAL) ** 0.5], dtype=torch.float32)
            elif "gamma_reparam" in key:
                t = torch.tensor([PEDESTAL ** 0.5], dtype=torch.float32)
            elif "lower_bound_scale" in key:
                t = torch.tensor([0.11], dtype=torch.float32)
            else:
                t = torch.tensor([1e-9], dtype=torch.float32)
        elif leaf == "scale_bound":
            t = torch.tensor([0.11], dtype=torch.float32)
        elif leaf == "scale_table":
            t = torch.zeros(0, dtype=torch.float32)
        elif leaf == "target":
            tt = math.log(2 / 1e-9 - 1)
            t = torch.tensor([-tt, 0.0, tt], dtype=torch.float32)
        elif leaf == "beta":
            t = math.sqrt(1.0 + PEDESTAL) + _u(key, shp, -0.2, 0.3, salt)
        elif leaf == "gamma":
            C = shp[0]
            t = torch.sqrt(0.1 * torch.eye(C) + PEDESTAL) + _u(key, shp, -0.004, 0.012, salt)
        elif leaf == "relative_position_bias_table":
            t = _u(key, shp, -0.5, 0.5, salt)
        elif leaf == "quantiles":
            q = _u(key, shp, -0.4, 0.4, salt)
            t = q + torch.tensor([-10.0, 0.0, 10.0])
        elif leaf.startswith("_matrix"):
            i = int(leaf[-1])
            filt = (1, 3, 3, 3, 3, 1)
            init = math.log(math.expm1(1 / (10 ** 0.2) / filt[i + 1]))
            t = init + _u(key, shp, -0.3, 0.3, salt)
        elif leaf.startswith("_bias"):
            t = _u(key, shp, -0.5, 0.5, salt)
        elif leaf.startswith("_factor"):
            t = _u(key, shp, -0.5, 0.5, salt)
        elif leaf == "weight" and len(shp) == 1:
            t = 1.0 + _u(key, shp, -0.2, 0.2, salt)       # LayerNorm weight
        elif leaf == "weight":
            if len(shp) == 4:
                # Conv2d [co,ci,k,k] and ConvTranspose2d [ci,co,k,k]: torch uses size(1)*k*k as fan_in
                fan_in = shp[1] * shp[2] * shp[3]
            else:
                fan_in = shp[1]
            b = 1.0 / math.sqrt(fan_in)
            t = _u(key, shp, -b, b, salt)
            t = t * gains.get(key, 1.0)
        elif leaf == "bias":
            if key.startswith("cc_scale_transforms") and key.endswith(".8.bias"):
                t = _u(key, shp, 0.0, 1.6, salt)
            else:
                t = _u(key, shp, -0.05, 0.05, salt)
        else:
            raise KeyError(key)
        sd[key] = t.to(torch.float32).reshape(shp).contiguous()
    return sd


# ----------------------------------------------------------------------------- stf (SymmetricalTransFormer)
def _swin_spec(s, p, dim, heads, ws=4):
    s[p + ".norm1.weight"] = (dim,)
    s[p + ".norm1.bias"] = (dim,)
    s[p + ".attn.relative_position_bias_table"] = ((2 * ws - 1) ** 2, heads)
    s[p + ".attn.relative_position_index"] = ("int64", (ws * ws, ws * ws))
    s[p + ".attn.qkv.weight"] = (3 * dim, dim)
    s[p + ".attn.qkv.bias"] = (3 * dim,)
    s[p + ".attn.proj.weight"] = (dim, dim)
    s[p + ".attn.proj.bias"] = (dim,)
    s[p + ".norm2.weight"] = (dim,)
    s[p + ".norm2.bias"] = (dim,)
    s[p + ".mlp.fc1.weight"] = (4 * dim, dim)
    s[p + ".mlp.fc1.bias"] = (4 * dim,)
    s[p + ".mlp.fc2.weight"] = (dim, 4 * dim)
    s[p + ".mlp.fc2.bias"] = (dim,)


def stf_spec(embed: int = 48) -> "OrderedDict[str, tuple]":
    """Ordered key -> shape of SymmetricalTransFormer.state_dict() (compressai/models/stf.py:318-500)."""
    s: "OrderedDict[str, tuple]" = OrderedDict()
    depths, heads = (2, 2, 6, 2), (3, 6, 12, 24)
    s["patch_embed.proj.weight"] = (embed, 3, 2, 2)
    s["patch_embed.proj.bias"] = (embed,)
    s["patch_embed.norm.weight"] = (embed,)
    s["patch_embed.norm.bias"] = (embed,)
    for i in range(4):
        dim = embed * 2 ** i
        for j in range(depths[i]):
            _swin_spec(s, f"layers.{i}.blocks.{j}", dim, heads[i])
        if i < 3:   # PatchMerging (stf.py:196-201)
            s[f"layers.{i}.downsample.reduction.weight"] = (2 * dim, 4 * dim)
            s[f"layers.{i}.downsample.norm.weight"] = (4 * dim,)
            s[f"layers.{i}.downsample.norm.bias"] = (4 * dim,)
    for i in range(4):
        dim = embed * 2 ** (3 - i)
        for j in range(depths[3 - i]):
            _swin_spec(s, f"syn_layers.{i}.blocks.{j}", dim, heads[3 - i])
        if i < 3:   # PatchSplit (stf.py:242-247)
            s[f"syn_layers.{i}.downsample.reduction.weight"] = (2 * dim, dim)
            s[f"syn_layers.{i}.downsample.norm.weight"] = (dim,)
            s[f"syn_layers.{i}.downsample.norm.bias"] = (dim,)
    _conv_spec(s, "end_conv.0", embed * 4, embed, 5)
    _conv_spec(s, "end_conv.2", 3, embed, 3)
    for i, (ci, co) in zip((0, 2, 4, 6, 8), ((384, 384), (384, 336), (336, 288), (288, 240), (240, 192))):
        _conv_spec(s, f"h_a.{i}", co, ci, 3)
    for h in ("h_mean_s", "h_scale_s"):
        _conv_spec(s, f"{h}.0", 240, 192, 3)
        _conv_spec(s, f"{h}.2.0", 288 * 4, 240, 3)
        _conv_spec(s, f"{h}.4", 336, 288, 3)
        _conv_spec(s, f"{h}.6.0", 384 * 4, 336, 3)
        _conv_spec(s, f"{h}.8", 384, 384, 3)
    chain = (224, 176, 128, 64, 32)
    for fam, extra in (("cc_mean_transforms", 0), ("cc_scale_transforms", 0), ("lrp_transforms", 1)):
        for i in range(12):
            ci = 384 + 32 * min(i + extra, 6 + extra)
            for j, co in zip((0, 2, 4, 6, 8), chain):
                _conv_spec(s, f"{fam}.{i}.{j}", co, ci, 3)
                ci = co
    tail = wacnn_spec()
    for k, v in tail.items():
        if k.startswith(("entropy_bottleneck.", "gaussian_conditional.")):
            s[k] = v
    return s


STF_GAINS = {"layers.3.blocks.1.mlp.fc2.weight": 6.0, "h_a.8.weight": 80.0}
for _i in range(12):
    STF_GAINS[f"cc_scale_transforms.{_i}.8.weight"] = 12.0
    STF_GAINS[f"cc_mean_transforms.{_i}.8.weight"] = 8.0
    STF_GAINS[f"lrp_transforms.{_i}.8.weight"] = 8.0


def make_stf_state_dict(salt: int = 0) -> "OrderedDict[str, torch.Tensor]":
    return _make(stf_spec(), STF_GAINS, salt)


# ----------------------------------------------------------------------------- stf6 (SymmetricalTransFormer3)
STF6_SLICES = 6            # stf6.py:393 num_slices -> 6 * 2 * 2 = 24 zigzag blocks of 64 channels
STF6_BLOCKS = 24
STF6_SUPPORT = 16          # stf6.py:414 max_support_slices


def stf6_spec(embed: int = 48) -> "OrderedDict[str, tuple]":
    """Ordered key -> shape of SymmetricalTransFormer3.state_dict() (compressai/models/stf6.py:384-622): the stf
    analysis / synthesis stacks, 24 + 6 + 6 four-layer Swin refinement stacks on 64-channel block maps (mu_Swin is the
    only one forward() uses, :806-810; sigma_Swin / LRP_Swin are registered but idle), the hyper path of stf and 24
    cc / lrp chains whose first layer sees 64 * (1 + min(i, 16)) (+ 64 for lrp) channels (:565-604)."""
    s: "OrderedDict[str, tuple]" = OrderedDict()
    depths, heads = (2, 2, 6, 2), (3, 6, 12, 24)
    base = stf_spec(embed)
    for k, v in base.items():            # patch_embed, layers, syn_layers come first, in stf's order
        if k.startswith(("patch_embed.", "layers.", "syn_layers.")):
            s[k] = v
    cs = 384 // STF6_SLICES
    rdepths = depths[::-1]
    for fam, n in (("mu_Swin", STF6_BLOCKS), ("sigma_Swin", STF6_SLICES), ("LRP_Swin", STF6_SLICES)):
        for b in range(n):
            for i in range(4):
                for j in range(rdepths[i]):
                    _swin_spec(s, f"{fam}.{b}.{i}.blocks.{j}", cs, 4)
    for k, v in base.items():
        if k.startswith(("end_conv.", "h_a.", "h_mean_s.", "h_scale_s.")):
            s[k] = v
    chain = (224, 176, 128, 64, cs)
    for fam, extra in (("cc_mean_transforms2", 0), ("cc_scale_transforms2", 0), ("lrp_transforms2", 1)):
        for i in range(STF6_BLOCKS):
            ci = cs + cs * min(i + extra, STF6_SUPPORT + extra)
            for j, co in zip((0, 2, 4, 6, 8), chain):
                _conv_spec(s, f"{fam}.{i}.{j}", co, ci, 3)
                ci = co
    for k, v in base.items():
        if k.startswith(("entropy_bottleneck.", "gaussian_conditional.")):
            s[k] = v
    return s


STF6_GAINS = {"layers.3.blocks.1.mlp.fc2.weight": 6.0, "h_a.8.weight": 80.0}
for _i in range(STF6_BLOCKS):
    STF6_GAINS[f"cc_scale_transforms2.{_i}.8.weight"] = 12.0
    STF6_GAINS[f"cc_mean_transforms2.{_i}.8.weight"] = 8.0
    STF6_GAINS[f"lrp_transforms2.{_i}.8.weight"] = 8.0
    STF6_GAINS[f"mu_Swin.{_i}.3.blocks.1.mlp.fc2.weight"] = 4.0    # the refinement must visibly move mu


def make_stf6_state_dict(salt: int = 0) -> "OrderedDict[str, torch.Tensor]":
    return _make(stf6_spec(), STF6_GAINS, salt)
