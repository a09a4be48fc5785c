"""CPU ORACLE for the `cnn` (WACNN) hot path -- TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement (PyTorch-CPU, IEEE f32) of the reference's
``CompressionModel.forward()`` path for the ``cnn`` model and of the training-step
semantics around it.  It is the *checker*: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  The product path
(``image-compression-for-machine_amd/``) never imports it and has no CPU fallback.

Parity pin: ``tests/golden/make_golden.py`` (run in the build container, where the
reference is importable) checks every function below against the *real* reference modules
on seeded inputs -- forward values and autograd gradients -- and writes the committed
fixtures ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` re-checks this file against
those fixtures everywhere (no reference needed).  The reference ships no tests / golden
vectors of its own for this path, so those generated fixtures are the
pin.

All functions work on a plain ``dict[str, Tensor]`` state-dict whose keys/shapes are those
of the reference model (``WACNN.state_dict()``: 585 entries).

Reference citations are relative to /root/reference.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

PEDESTAL = 2.0 ** -36          # compressai/ops/parametrizers.py:37-38 (reparam_offset 2^-18 squared)
SCALE_BOUND = 0.11             # compressai/entropy_models/entropy_models.py:540
LIK_BOUND = 1e-9               # compressai/entropy_models/entropy_models.py:82
NUM_SLICES = 10                # compressai/models/cnn.py:28
MAX_SUPPORT = 5                # compressai/models/cnn.py:29


# --------------------------------------------------------------------------- ops
class _LowerBound(torch.autograd.Function):
    """max(x, b) whose backward passes g where (x >= b) or (g < 0).
    compressai/ops/bound_ops.py:21-43."""

    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x, bound)
        return torch.max(x, bound)

    @staticmethod
    def backward(ctx, g):
        x, bound = ctx.saved_tensors
        keep = (x >= bound) | (g < 0)
        return keep * g, None


def lower_bound(x: Tensor, bound: float) -> Tensor:
    return _LowerBound.apply(x, torch.tensor([float(bound)], dtype=x.dtype))


def ste_round(x: Tensor) -> Tensor:
    """(round(x) - x) + x with identity gradient. compressai/ops/ops.py:20-34."""
    return torch.round(x) - x.detach() + x


def nonneg_param(p: Tensor, minimum: float = 0.0) -> Tensor:
    """NonNegativeParametrizer.forward. compressai/ops/parametrizers.py:23-49."""
    bound = (float(minimum) + PEDESTAL) ** 0.5
    out = lower_bound(p, bound)
    return out ** 2 - torch.tensor([PEDESTAL], dtype=p.dtype)


def gelu(x: Tensor) -> Tensor:
    return F.gelu(x)  # exact erf form (nn.GELU default), compressai/layers/layers.py:59-63


# --------------------------------------------------------------------------- layers
def gdn(x: Tensor, beta_p: Tensor, gamma_p: Tensor, inverse: bool, beta_min: float = 1e-6) -> Tensor:
    """GDN / IGDN. compressai/layers/gdn.py:62-75."""
    C = x.shape[1]
    beta = nonneg_param(beta_p, beta_min)
    gamma = nonneg_param(gamma_p, 0.0).reshape(C, C, 1, 1)
    norm = F.conv2d(x ** 2, gamma, beta)
    norm = torch.sqrt(norm) if inverse else torch.rsqrt(norm)
    return x * norm


def conv(x, sd, p, stride, k):
    """models/utils.py:114-121 (k5 s2 p2) and layers/layers.py:29-43 (3x3 p1 / 1x1 p0)."""
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=stride, padding=k // 2)


def deconv(x, sd, p, stride=2, k=5):
    """models/utils.py:124-132."""
    return F.conv_transpose2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=stride,
                              padding=k // 2, output_padding=stride - 1)


def residual_unit(x, sd, p):
    """layers/layers.py:52-72: gelu(x + conv1x1(gelu(conv3x3(gelu(conv1x1 x)))))."""
    out = gelu(conv(x, sd, p + ".conv.0", 1, 1))
    out = gelu(conv(out, sd, p + ".conv.2", 1, 3))
    out = conv(out, sd, p + ".conv.4", 1, 1)
    return gelu(out + x)


def relative_position_index(ws: int) -> Tensor:
    """win_attention.py:64-74 -> int64 [ws*ws, ws*ws]."""
    ch = torch.arange(ws)
    coords = torch.stack(torch.meshgrid(ch, ch, indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def shift_mask(H: int, W: int, ws: int, shift: int) -> Tensor:
    """SW-MSA mask [nW, ws*ws, ws*ws] of 0 / -100. win_attention.py:159-177."""
    img = torch.zeros(H, W)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[hs, wsl] = cnt
            cnt += 1
    mw = img.view(H // ws, ws, W // ws, ws).permute(0, 2, 1, 3).reshape(-1, ws * ws)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return torch.where(am != 0, torch.full_like(am, -100.0), torch.zeros_like(am))


def win_based_attention(x: Tensor, sd, p: str, heads: int, ws: int, shift: int) -> Tensor:
    """WinBasedAttention.forward + WindowAttention.forward.
    win_attention.py:84-115,153-207.  x: [B,C,H,W] -> [B,C,H,W] (shortcut added)."""
    B, C, H, W = x.shape
    hd = C // heads
    t = x.permute(0, 2, 3, 1)
    if shift > 0:
        t = torch.roll(t, shifts=(-shift, -shift), dims=(1, 2))
    win = t.reshape(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)
    Bn, N, _ = win.shape
    qkv = F.linear(win, sd[p + ".attn.qkv.weight"], sd[p + ".attn.qkv.bias"])
    qkv = qkv.reshape(Bn, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (hd ** -0.5), qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    idx = relative_position_index(ws).reshape(-1)
    bias = sd[p + ".attn.relative_position_bias_table"][idx].reshape(N, N, heads).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if shift > 0:
        mask = shift_mask(H, W, ws, shift)
        nW = mask.shape[0]
        attn = attn.reshape(Bn // nW, nW, heads, N, N) + mask[None, :, None]
        attn = attn.reshape(-1, heads, N, N)
    attn = torch.softmax(attn, dim=-1)
    o = (attn @ v).transpose(1, 2).reshape(Bn, N, C)
    o = F.linear(o, sd[p + ".attn.proj.weight"], sd[p + ".attn.proj.bias"])
    o = o.reshape(B, H // ws, W // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, C)
    if shift > 0:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    return x + o.permute(0, 3, 1, 2)


def win_attention_gate(x: Tensor, sd, p: str, heads: int, ws: int, shift: int) -> Tensor:
    """Win_noShift_Attention.forward: a*sigmoid(b)+x. layers/layers.py:45-89."""
    a = x
    for i in range(3):
        a = residual_unit(a, sd, f"{p}.conv_a.{i}")
    b = win_based_attention(x, sd, p + ".conv_b.0", heads, ws, shift)
    for i in (1, 2, 3):
        b = residual_unit(b, sd, f"{p}.conv_b.{i}")
    b = conv(b, sd, p + ".conv_b.4", 1, 1)
    return a * torch.sigmoid(b) + x


# --------------------------------------------------------------------------- entropy models
def eb_logits_cumulative(v: Tensor, sd, p: str, detach: bool) -> Tensor:
    """EntropyBottleneck._logits_cumulative; v: [C,1,L]. entropy_models.py:400-419."""
    logits = v
    for i in range(5):
        m = sd[f"{p}._matrix{i}"]
        b = sd[f"{p}._bias{i}"]
        if detach:
            m, b = m.detach(), b.detach()
        logits = torch.matmul(F.softplus(m), logits) + b
        if i < 4:
            f = sd[f"{p}._factor{i}"]
            if detach:
                f = f.detach()
            logits = logits + torch.tanh(f) * torch.tanh(logits)
    return logits


def eb_likelihood(z: Tensor, sd, p: str = "entropy_bottleneck", noise: Optional[Tensor] = None):
    """EntropyBottleneck.forward -> (z_tilde, likelihood), both [B,C,H,W].
    entropy_models.py:421-433,446-489; quantize :126-150.
    noise=None -> eval ("dequantize" around the medians); else z + noise ("noise")."""
    B, C = z.shape[:2]
    vals = z.transpose(0, 1).reshape(C, 1, -1)
    med = sd[p + ".quantiles"][:, :, 1:2]
    if noise is None:
        out = torch.round(vals - med) + med
    else:
        out = vals + noise.transpose(0, 1).reshape(C, 1, -1)
    lower = eb_logits_cumulative(out - 0.5, sd, p, False)
    upper = eb_logits_cumulative(out + 0.5, sd, p, False)
    sign = -torch.sign(lower + upper).detach()
    lik = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))
    lik = lower_bound(lik, LIK_BOUND)
    shp = (C, B) + tuple(z.shape[2:])
    return out.reshape(shp).transpose(0, 1), lik.reshape(shp).transpose(0, 1)


def eb_aux_loss(sd, p: str = "entropy_bottleneck", tail_mass: float = 1e-9) -> Tensor:
    """EntropyBottleneck.loss / CompressionModel.aux_loss. entropy_models.py:395-398; base.py:22-29."""
    t = math.log(2 / tail_mass - 1)
    target = torch.tensor([-t, 0.0, t], dtype=torch.float32)
    logits = eb_logits_cumulative(sd[p + ".quantiles"], sd, p, True)
    return torch.abs(logits - target).sum()


def gaussian_likelihood(y: Tensor, scales: Tensor, means: Tensor, noise: Optional[Tensor] = None,
                        round_to: Optional[Tensor] = None):
    """GaussianConditional.forward -> (y_tilde, likelihood). entropy_models.py:578-582,626-659.
    round_to (tests only, eval mode): adopt these rounding decisions round(y - means) instead of taking them here, so
    that a latent within float noise of a half-integer yields the likelihood of the symbol the checked implementation
    coded (flips are counted separately by the tests)."""
    if noise is None:
        out = (torch.round(y - means) if round_to is None else round_to.to(y.dtype)) + means
    else:
        out = y + noise
    s = lower_bound(scales, SCALE_BOUND)
    v = torch.abs(out - means)
    c = -(2 ** -0.5)
    upper = 0.5 * torch.erfc(c * ((0.5 - v) / s))
    lower = 0.5 * torch.erfc(c * ((-0.5 - v) / s))
    lik = lower_bound(upper - lower, LIK_BOUND)
    return out, lik


# --------------------------------------------------------------------------- CDF tables (update())
def eb_update_tables(sd, p: str = "entropy_bottleneck"):
    """EntropyBottleneck.update up to the pmf (entropy_models.py:354-390): returns (offset [C] int, pmf [C,L],
    tail_mass [C,1], pmf_length [C] int, max_length).  The quantisation of each row to a CDF is the coder's
    pmf_to_quantized_cdf (oracle/rans_oracle.py)."""
    q = sd[p + ".quantiles"].detach()
    medians = q[:, 0, 1]
    minima = torch.clamp(torch.ceil(medians - q[:, 0, 0]).int(), min=0)
    maxima = torch.clamp(torch.ceil(q[:, 0, 2] - medians).int(), min=0)
    pmf_start = medians - minima
    pmf_length = maxima + minima + 1
    max_length = int(pmf_length.max().item())
    samples = torch.arange(max_length)[None, :] + pmf_start[:, None, None]
    lower = eb_logits_cumulative(samples - 0.5, sd, p, True).detach()
    upper = eb_logits_cumulative(samples + 0.5, sd, p, True).detach()
    sign = -torch.sign(lower + upper)
    pmf = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))[:, 0, :]
    tail_mass = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
    return -minima, pmf, tail_mass, pmf_length, max_length


def gc_update_tables(scale_table: Tensor, tail_mass: float = 1e-9):
    """GaussianConditional.update up to the pmf (entropy_models.py:598-620): (offset, pmf, tail_mass, pmf_length, max_length)"""
    import scipy.stats
    multiplier = -scipy.stats.norm.ppf(tail_mass / 2)
    pmf_center = torch.ceil(scale_table * multiplier).int()
    pmf_length = 2 * pmf_center + 1
    max_length = int(torch.max(pmf_length).item())
    samples = torch.abs(torch.arange(max_length).int() - pmf_center[:, None]).float()
    sc = scale_table.unsqueeze(1).float()
    c = -(2 ** -0.5)
    upper = 0.5 * torch.erfc(c * ((0.5 - samples) / sc))
    lower = 0.5 * torch.erfc(c * ((-0.5 - samples) / sc))
    return -pmf_center, upper - lower, 2 * lower[:, :1], pmf_length, max_length


def gc_build_indexes(scales: Tensor, scale_table: Tensor) -> Tensor:
    """GaussianConditional.build_indexes (entropy_models.py:661-666)"""
    s = torch.max(scales, torch.tensor(SCALE_BOUND))
    idx = torch.full(s.shape, len(scale_table) - 1, dtype=torch.int32)
    for t in scale_table[:-1]:
        idx -= (s <= t).int()
    return idx


def scale_table(lo: float = 0.11, hi: float = 256, levels: int = 64) -> Tensor:
    """get_scale_table (models/cnn.py:14-21 region: SCALES_MIN / MAX / LEVELS)"""
    return torch.exp(torch.linspace(math.log(lo), math.log(hi), levels))


# --------------------------------------------------------------------------- model
def _seq_convs(x, sd, p, idxs, strides=None, final_act=False):
    """conv3x3 (+GELU between) stacks of h_a / cc / lrp. cnn.py:54-127."""
    for j, i in enumerate(idxs):
        s = 1 if strides is None else strides[j]
        x = conv(x, sd, f"{p}.{i}", s, 3)
        if j + 1 < len(idxs) or final_act:
            x = gelu(x)
    return x


def h_a(y, sd):
    return _seq_convs(y, sd, "h_a", (0, 2, 4, 6, 8), (1, 1, 2, 1, 2))  # cnn.py:54-64


def h_s(z_hat, sd, p):
    """h_mean_s / h_scale_s: conv3, subpel(2), conv3, subpel(2), conv3 with GELU. cnn.py:66-88."""
    x = gelu(conv(z_hat, sd, p + ".0", 1, 3))
    x = gelu(F.pixel_shuffle(conv(x, sd, p + ".2.0", 1, 3), 2))
    x = gelu(conv(x, sd, p + ".4", 1, 3))
    x = gelu(F.pixel_shuffle(conv(x, sd, p + ".6.0", 1, 3), 2))
    return conv(x, sd, p + ".8", 1, 3)


def g_a(x, sd):
    """cnn.py:31-41."""
    x = conv(x, sd, "g_a.0", 2, 5)
    x = gdn(x, sd["g_a.1.beta"], sd["g_a.1.gamma"], False)
    x = conv(x, sd, "g_a.2", 2, 5)
    x = gdn(x, sd["g_a.3.beta"], sd["g_a.3.gamma"], False)
    x = win_attention_gate(x, sd, "g_a.4", 8, 8, 4)
    x = conv(x, sd, "g_a.5", 2, 5)
    x = gdn(x, sd["g_a.6.beta"], sd["g_a.6.gamma"], False)
    x = conv(x, sd, "g_a.7", 2, 5)
    return win_attention_gate(x, sd, "g_a.8", 8, 4, 2)


def g_s(y_hat, sd):
    """cnn.py:42-52."""
    x = win_attention_gate(y_hat, sd, "g_s.0", 8, 4, 2)
    x = deconv(x, sd, "g_s.1")
    x = gdn(x, sd["g_s.2.beta"], sd["g_s.2.gamma"], True)
    x = deconv(x, sd, "g_s.3")
    x = gdn(x, sd["g_s.4.beta"], sd["g_s.4.gamma"], True)
    x = win_attention_gate(x, sd, "g_s.5", 8, 8, 4)
    x = deconv(x, sd, "g_s.6")
    x = gdn(x, sd["g_s.7.beta"], sd["g_s.7.gamma"], True)
    return deconv(x, sd, "g_s.8")


def ste_round_as(x: Tensor, r: Optional[Tensor]) -> Tensor:
    """ste_round with the rounding DECISION taken from ``r`` (integers of x's shape) when given: same value and
    gradient structure, but a latent that sits within float noise of a half-integer rounds the way the checked
    implementation rounded it, so the comparison downstream stays smooth (flips are counted separately)."""
    if r is None:
        return ste_round(x)
    return (r.to(x.dtype) - x).detach() + x


def hyper_slices(y: Tensor, sd: Dict[str, Tensor], noise: Optional[Dict[str, Tensor]], num_slices: int,
                 max_support: int, round_override: Optional[Dict[str, Tensor]] = None):
    """Hyperprior + channel-conditional slice loop shared by the cnn and stf models
    (cnn.py:144-183 == stf.py:596-637).  Returns (y_hat, y_lik, z_lik, dbg).
    round_override (tests only): {"y": round(y - mu) [B,M,H,W], "z": round(z - med)} decisions to adopt."""
    ro = round_override or {}
    z = h_a(y, sd)
    _, z_lik = eb_likelihood(z, sd, "entropy_bottleneck", None if noise is None else noise["z"])
    med = sd["entropy_bottleneck.quantiles"][:, :, 1:2].reshape(1, -1, 1, 1)
    z_hat = ste_round_as(z - med, ro.get("z")) + med
    lat_scales = h_s(z_hat, sd, "h_scale_s")
    lat_means = h_s(z_hat, sd, "h_mean_s")
    y_slices = y.chunk(num_slices, 1)
    n_slices = None if noise is None else noise["y"].chunk(num_slices, 1)
    y_hat_slices, liks, mus, scales = [], [], [], []
    for i, ys in enumerate(y_slices):
        sup = y_hat_slices[:max_support]
        mean_sup = torch.cat([lat_means] + sup, 1)
        mu = _seq_convs(mean_sup, sd, f"cc_mean_transforms.{i}", (0, 2, 4, 6, 8))
        mu = mu[:, :, :y.shape[2], :y.shape[3]]
        scale_sup = torch.cat([lat_scales] + sup, 1)
        sc = _seq_convs(scale_sup, sd, f"cc_scale_transforms.{i}", (0, 2, 4, 6, 8))
        sc = sc[:, :, :y.shape[2], :y.shape[3]]
        ry = ro["y"].chunk(num_slices, 1)[i] if "y" in ro else None
        _, lik = gaussian_likelihood(ys, sc, mu, None if noise is None else n_slices[i], round_to=ry)
        liks.append(lik)
        yh = ste_round_as(ys - mu, ry) + mu
        lrp = _seq_convs(torch.cat([mean_sup, yh], 1), sd, f"lrp_transforms.{i}", (0, 2, 4, 6, 8))
        yh = yh + 0.5 * torch.tanh(lrp)
        y_hat_slices.append(yh)
        mus.append(mu)
        scales.append(sc)
    y_hat = torch.cat(y_hat_slices, 1)
    y_lik = torch.cat(liks, 1)
    dbg = {"y": y, "z": z, "z_hat": z_hat, "y_hat": y_hat, "mu": torch.cat(mus, 1),
           "scale": torch.cat(scales, 1), "lat_means": lat_means, "lat_scales": lat_scales}
    return y_hat, y_lik, z_lik, dbg


def wacnn_forward(sd: Dict[str, Tensor], x: Tensor, noise: Optional[Dict[str, Tensor]] = None,
                  keep: bool = False, round_override: Optional[Dict[str, Tensor]] = None) -> Dict:
    """WACNN.forward. compressai/models/cnn.py:141-189.

    noise: None -> eval-mode quantisation.  Else {"z": [B,192,h,w], "y": [B,320,H,W]} uniform
    (-0.5,0.5) samples injected where the reference draws them in train mode
    (entropy_models.py:131-135) -- only the *likelihood* inputs are noised; z_hat / y_hat use
    ste_round in both modes (cnn.py:150-152,173)."""
    y = g_a(x, sd)
    y_hat, y_lik, z_lik, dbg = hyper_slices(y, sd, noise, NUM_SLICES, MAX_SUPPORT, round_override)
    x_hat = g_s(y_hat, sd)
    out = {"x_hat": x_hat, "likelihoods": {"y": y_lik, "z": z_lik}}
    if keep:
        out["_dbg"] = dbg
    return out


# --------------------------------------------------------------------------- loss / optimiser
def rd_loss(x: Tensor, out: Dict, lmbda: float = 0.0067) -> Dict[str, Tensor]:
    """bpp = sum_k sum log(lik_k) / (-ln2 * N*H*W); mse = mean((x-x_hat)^2);
    loss = lmbda*255^2*mse + bpp.  train.py:53-61 (bpp), train_czigzag.py:63,71 (loss form)."""
    N, _, H, W = x.shape
    npix = N * H * W
    bpp = sum(torch.log(l).sum() / (-math.log(2) * npix) for l in out["likelihoods"].values())
    mse = F.mse_loss(out["x_hat"], x)
    return {"bpp_loss": bpp, "mse_loss": mse, "loss": lmbda * 255 ** 2 * mse + bpp}


def psnr(mse: float) -> float:
    """utils/eval_model/__main__.py:78-80 (inputs in [0,1])."""
    return -10.0 * math.log10(mse)


def clip_grad_norm_(grads, max_norm: float, eps: float = 1e-6) -> Tensor:
    """torch.nn.utils.clip_grad_norm_ semantics used at train.py:208-209 (L2, all params)."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + eps), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
              b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8) -> None:
    """torch.optim.Adam defaults (train.py:158-165): in-place, step counted from 1."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)
