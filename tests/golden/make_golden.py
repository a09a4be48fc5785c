"""Generate the committed golden fixtures from the REAL reference (build container only).

    python tests/golden/make_golden.py

Imports /root/reference through ``_ref_loader``, runs the reference's own
modules (GDN, WinBasedAttention, Win_noShift_Attention, EntropyBottleneck, GaussianConditional,
ste_round, LowerBound, NonNegativeParametrizer, WACNN) on formula-generated inputs/weights, checks
the CPU oracle (``oracle/wacnn_oracle.py``) against them, and stores inputs + expected outputs +
autograd gradients as small ``.npz`` files in this directory.  Fixtures are data only.
"""
import json
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

from _ref_loader import load_reference  # noqa: E402
from oracle import weights as W  # noqa: E402
from oracle import wacnn_oracle as O  # noqa: E402
from oracle import stf_oracle as S  # noqa: E402

torch.set_num_threads(8)
cnn_mod, stf_mod = load_reference()
import compressai.layers.gdn as ref_gdn  # noqa: E402
import compressai.layers.layers as ref_layers  # noqa: E402
import compressai.layers.win_attention as ref_wa  # noqa: E402
import compressai.entropy_models.entropy_models as ref_em  # noqa: E402
import compressai.ops.ops as ref_ops  # noqa: E402
import compressai.ops.bound_ops as ref_bound  # noqa: E402
import compressai.ops.parametrizers as ref_par  # noqa: E402


def U(key, shape, lo, hi):
    return W._u(key, shape, lo, hi)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {name}.npz  {os.path.getsize(path)/1024:.1f} KB")


def check(a, b, what, tol=0.0):
    d = (a - b).abs().max().item() if a.numel() else 0.0
    ref = b.abs().max().item() if b.numel() else 1.0
    assert d <= tol * max(ref, 1e-30) + (0 if tol == 0 else 1e-30) or d == 0.0, f"{what}: oracle != reference, maxdiff {d} (ref max {ref})"
    return d


def fill_module(mod, prefix):
    """Overwrite a reference module's parameters with formula values; return {prefix.key: tensor}."""
    sd = mod.state_dict()
    new = {}
    for k, v in sd.items():
        leaf = k.rsplit(".", 1)[-1]
        key = prefix + "." + k
        if not v.dtype.is_floating_point or leaf in ("pedestal", "bound", "target", "scale_bound", "scale_table"):
            new[k] = v
        elif leaf == "beta":
            new[k] = math.sqrt(1 + W.PEDESTAL) + U(key, v.shape, -0.2, 0.3)
        elif leaf == "gamma":
            C = v.shape[0]
            new[k] = torch.sqrt(0.1 * torch.eye(C) + W.PEDESTAL) + U(key, v.shape, -0.004, 0.012)
        elif leaf == "relative_position_bias_table":
            new[k] = U(key, v.shape, -0.5, 0.5)
        elif leaf == "quantiles":
            new[k] = U(key, v.shape, -0.4, 0.4) + torch.tensor([-10.0, 0.0, 10.0])
        elif leaf.startswith("_matrix"):
            new[k] = v + U(key, v.shape, -0.3, 0.3)
        elif leaf.startswith("_bias") or leaf.startswith("_factor"):
            new[k] = U(key, v.shape, -0.5, 0.5)
        elif leaf == "weight" and v.dim() == 1:
            new[k] = 1.0 + U(key, v.shape, -0.2, 0.2)
        elif leaf == "weight":
            fan_in = int(np.prod(v.shape[1:]))
            b = 1.0 / math.sqrt(fan_in)
            new[k] = U(key, v.shape, -b, b) * 1.7
        elif leaf == "bias":
            new[k] = U(key, v.shape, -0.1, 0.1)
        else:
            raise KeyError(k)
        new[k] = new[k].to(v.dtype).reshape(v.shape)
    mod.load_state_dict(new)
    return {prefix + "." + k: v.clone() for k, v in mod.state_dict().items()}


def grads_of(out, g, tensors):
    gs = torch.autograd.grad(out, tensors, g, allow_unused=True)
    return [torch.zeros_like(t) if x is None else x for x, t in zip(gs, tensors)]


def leafify(sd):
    return {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point else v) for k, v in sd.items()}


# ------------------------------------------------------------------ ops
def gen_ops():
    print("ops")
    x = torch.tensor([-2.5, -1.5, -0.5, 0.5, 1.5, 2.5, 0.49999997, 0.50000006, 1e-9, -7.25, 3.75, 1234.5],
                     dtype=torch.float32, requires_grad=True)
    y = ref_ops.ste_round(x)
    assert torch.equal(y, O.ste_round(x))
    gx, = torch.autograd.grad(y, x, torch.arange(12, dtype=torch.float32))
    lb_x = torch.tensor([0.05, 0.11, 0.2, 0.05, 0.11, 0.2, -3.0, 0.10999999], requires_grad=True)
    lb_g = torch.tensor([1.0, 1.0, 1.0, -1.0, -1.0, -1.0, 0.0, -0.0])
    lb = ref_bound.LowerBound(0.11)
    lo = lb(lb_x)
    lgx, = torch.autograd.grad(lo, lb_x, lb_g)
    lo2 = O.lower_bound(lb_x, 0.11)
    lgx2, = torch.autograd.grad(lo2, lb_x, lb_g)
    assert torch.equal(lo, lo2) and torch.equal(lgx, lgx2)
    par = ref_par.NonNegativeParametrizer(minimum=1e-6)
    px = torch.tensor([-1.0, 0.0, 5e-4, 1e-3, 1.0000001e-3, 0.3, 2.0], requires_grad=True)
    pg = torch.tensor([1.0, -1.0, 1.0, -2.0, 3.0, 1.0, -1.0])
    po = par(px)
    pgx, = torch.autograd.grad(po, px, pg)
    po2 = O.nonneg_param(px, 1e-6)
    pgx2, = torch.autograd.grad(po2, px, pg)
    assert torch.equal(po, po2) and torch.equal(pgx, pgx2)
    save("ops", ste_x=x, ste_y=y, ste_gx=gx, lb_x=lb_x, lb_g=lb_g, lb_y=lo, lb_gx=lgx,
         nn_x=px, nn_g=pg, nn_y=po, nn_gx=pgx)


def gen_gdn():
    print("gdn")
    C = 48
    for inverse in (False, True):
        name = "igdn" if inverse else "gdn"
        mod = ref_gdn.GDN(C, inverse=inverse)
        sd = fill_module(mod, name)
        x = U(name + ".x", (2, C, 8, 8), -2.0, 2.0).requires_grad_(True)
        g = U(name + ".g", (2, C, 8, 8), -1.0, 1.0)
        y = mod(x)
        gx, gb, gg = torch.autograd.grad(y, [x, mod.beta, mod.gamma], g)
        s = leafify(sd)
        y2 = O.gdn(x, s[name + ".beta"], s[name + ".gamma"], inverse)
        gx2, gb2, gg2 = torch.autograd.grad(y2, [x, s[name + ".beta"], s[name + ".gamma"]], g)
        for a, b, w in ((y2, y, "y"), (gx2, gx, "gx"), (gb2, gb, "gbeta"), (gg2, gg, "ggamma")):
            check(a, b, name + "." + w)
        frac_bound = (sd[name + ".gamma"] < W.PEDESTAL ** 0.5).float().mean().item()
        print(f"  {name}: gamma below bound {frac_bound:.2f}")
        save(name, x=x, g=g, beta=sd[name + ".beta"], gamma=sd[name + ".gamma"], y=y, gx=gx, gbeta=gb, ggamma=gg)


def gen_attention():
    print("attention")
    for tag, dim, ws, shift, hw in (("wa_d64_ws8", 64, 8, 4, 16), ("wa_d80_ws4", 80, 4, 2, 8), ("wa_d64_ws8_noshift", 64, 8, 0, 16)):
        mod = ref_wa.WinBasedAttention(dim=dim, num_heads=8, window_size=ws, shift_size=shift)
        sd = fill_module(mod, tag)
        x = U(tag + ".x", (2, dim, hw, hw), -1.5, 1.5).requires_grad_(True)
        g = U(tag + ".g", (2, dim, hw, hw), -1.0, 1.0)
        y = mod(x)
        params = [mod.attn.qkv.weight, mod.attn.qkv.bias, mod.attn.proj.weight, mod.attn.proj.bias,
                  mod.attn.relative_position_bias_table]
        gs = torch.autograd.grad(y, [x] + params, g)
        s = leafify(sd)
        y2 = O.win_based_attention(x, s, tag, 8, ws, shift)
        keys = [tag + ".attn.qkv.weight", tag + ".attn.qkv.bias", tag + ".attn.proj.weight", tag + ".attn.proj.bias",
                tag + ".attn.relative_position_bias_table"]
        gs2 = torch.autograd.grad(y2, [x] + [s[k] for k in keys], g)
        check(y2, y, tag + ".y", 1e-6)
        for a, b, k in zip(gs2, gs, ["x"] + keys):
            check(a, b, tag + ".grad." + k, 2e-6)
        save(tag, x=x, g=g, y=y, gx=gs[0], g_qkv_w=gs[1], g_qkv_b=gs[2], g_proj_w=gs[3], g_proj_b=gs[4], g_table=gs[5],
             **{k[len(tag) + 1:]: v for k, v in sd.items() if v.dtype.is_floating_point})
    # whole gate (ResidualUnits + attention + sigmoid gate)
    tag, dim, ws, shift, hw = "gate_d64_ws8", 64, 8, 4, 16
    mod = ref_layers.Win_noShift_Attention(dim=dim, num_heads=8, window_size=ws, shift_size=shift)
    sd = fill_module(mod, tag)
    x = U(tag + ".x", (1, dim, hw, hw), -1.5, 1.5).requires_grad_(True)
    g = U(tag + ".g", (1, dim, hw, hw), -1.0, 1.0)
    y = mod(x)
    pnames = [n for n, _ in mod.named_parameters()]
    gs = torch.autograd.grad(y, [x] + [p for _, p in mod.named_parameters()], g)
    s = leafify(sd)
    y2 = O.win_attention_gate(x, s, tag, 8, ws, shift)
    gs2 = torch.autograd.grad(y2, [x] + [s[tag + "." + n] for n in pnames], g)
    check(y2, y, tag + ".y", 1e-6)
    for a, b, k in zip(gs2, gs, ["x"] + pnames):
        check(a, b, tag + ".grad." + k, 5e-6)
    # store output, gx and the per-parameter gradient L2 norms (compact)
    gn = torch.stack([t.norm() for t in gs[1:]])
    save(tag, x=x, g=g, y=y, gx=gs[0], grad_norms=gn, grad_names=np.array(pnames),
         g_first_conv_w=gs[1 + pnames.index("conv_a.0.conv.0.weight")],
         g_last_conv_b=gs[1 + pnames.index("conv_b.4.bias")])


def gen_entropy():
    print("entropy models")
    C = 24
    eb = ref_em.EntropyBottleneck(C)
    sd = fill_module(eb, "entropy_bottleneck")
    z = U("eb.z", (2, C, 4, 4), -6.0, 6.0).requires_grad_(True)
    g = U("eb.g", (2, C, 4, 4), -1.0, 0.0)   # d/dlik of -log(lik)-type losses is negative
    gpos = U("eb.gpos", (2, C, 4, 4), -1.0, 1.0)
    noise = U("eb.noise", (2, C, 4, 4), -0.5, 0.5)
    pn = [n for n, _ in eb.named_parameters()]
    res = {}
    for mode in ("eval", "train"):
        eb.train(mode == "train")
        if mode == "train":
            orig = ref_em.EntropyModel.quantize

            def q(self, inputs, m, means=None, _o=orig):
                if m == "noise":
                    Cn = inputs.shape[0]
                    return inputs + noise.transpose(0, 1).reshape(Cn, 1, -1)
                return _o(self, inputs, m, means)
            ref_em.EntropyModel.quantize = q
        zt, lik = eb(z)
        if mode == "train":
            ref_em.EntropyModel.quantize = orig
        for gname, gg in (("g", g), ("gpos", gpos)):
            gs = torch.autograd.grad(lik, [z] + [p for _, p in eb.named_parameters()], gg, allow_unused=True, retain_graph=True)
            gs = [torch.zeros_like(t) if a is None else a for a, t in zip(gs, [z] + [p for _, p in eb.named_parameters()])]
            s = leafify(sd)
            zt2, lik2 = O.eb_likelihood(z, s, "entropy_bottleneck", noise if mode == "train" else None)
            gs2 = grads_of(lik2, gg, [z] + [s["entropy_bottleneck." + n] for n in pn])
            check(zt2, zt, f"eb.{mode}.zt")
            check(lik2, lik, f"eb.{mode}.lik")
            for a, b, k in zip(gs2, gs, ["z"] + pn):
                check(a, b, f"eb.{mode}.{gname}.grad.{k}")
            res[f"{mode}_{gname}_gz"] = gs[0]
            for n, a in zip(pn, gs[1:]):
                res[f"{mode}_{gname}_grad{n}"] = a
        res[f"{mode}_zt"] = zt
        res[f"{mode}_lik"] = lik
    aux = eb.loss()
    gq, = torch.autograd.grad(aux, eb.quantiles)
    s = leafify(sd)
    aux2 = O.eb_aux_loss(s)
    gq2, = torch.autograd.grad(aux2, s["entropy_bottleneck.quantiles"])
    check(aux2, aux, "eb.aux")
    check(gq2, gq, "eb.aux.gq")
    save("entropy_bottleneck", z=z, g=g, gpos=gpos, noise=noise, aux=aux, aux_gq=gq,
         **{"p" + n: sd["entropy_bottleneck." + n] for n in pn}, **res)

    gc = ref_em.GaussianConditional(None)
    shape = (2, 8, 16, 16)
    y = U("gc.y", shape, -8.0, 8.0)
    mu = U("gc.mu", shape, -1.0, 1.0)
    sc = U("gc.sc", shape, -0.3, 2.5)
    # corners: sigma exactly at / just around the bound, tiny sigma with far-away y (lik -> 1e-9 bound)
    sc.view(-1)[:6] = torch.tensor([0.11, 0.10999999, 0.11000001, 0.0, -1.0, 1e-3])
    y.view(-1)[6:10] = torch.tensor([40.0, -40.0, 0.5, 1.5])
    sc.view(-1)[6:10] = torch.tensor([0.2, 0.05, 0.3, 0.3])
    mu.view(-1)[8:10] = 0.0
    noise = U("gc.noise", shape, -0.5, 0.5)
    g = U("gc.g", shape, -1.0, 0.0)
    gpos = U("gc.gpos", shape, -1.0, 1.0)
    res = {}
    for mode in ("eval", "train"):
        gc.train(mode == "train")
        yl, ml, sl = (t.clone().requires_grad_(True) for t in (y, mu, sc))
        if mode == "train":
            orig = ref_em.EntropyModel.quantize
            ref_em.EntropyModel.quantize = lambda self, inputs, m, means=None, _o=orig: (inputs + noise) if m == "noise" else _o(self, inputs, m, means)
        yt, lik = gc(yl, sl, ml)
        if mode == "train":
            ref_em.EntropyModel.quantize = orig
        for gname, gg in (("g", g), ("gpos", gpos)):
            gy, gm, gs_ = grads_of(lik, gg, [yl, ml, sl]) if False else torch.autograd.grad(lik, [yl, ml, sl], gg, retain_graph=True, allow_unused=True)
            gy = torch.zeros_like(yl) if gy is None else gy
            y2, m2, s2 = (t.clone().requires_grad_(True) for t in (y, mu, sc))
            yt2, lik2 = O.gaussian_likelihood(y2, s2, m2, noise if mode == "train" else None)
            gy2, gm2, gs2 = grads_of(lik2, gg, [y2, m2, s2])
            check(yt2, yt, f"gc.{mode}.yt")
            check(lik2, lik, f"gc.{mode}.lik")
            check(gy2, gy, f"gc.{mode}.gy")
            check(gm2, gm, f"gc.{mode}.gm")
            check(gs2, gs_, f"gc.{mode}.gs")
            res.update({f"{mode}_{gname}_gy": gy, f"{mode}_{gname}_gmu": gm, f"{mode}_{gname}_gsc": gs_})
        res.update({f"{mode}_yt": yt, f"{mode}_lik": lik})
    save("gaussian_conditional", y=y, mu=mu, sc=sc, noise=noise, g=g, gpos=gpos, **res)


def gen_wacnn():
    print("wacnn end-to-end")
    sd = W.make_wacnn_state_dict()
    model = cnn_mod.WACNN()
    rsd = model.state_dict()
    assert list(rsd.keys()) == list(sd.keys()), "state-dict key order/list differs from reference"
    for k, v in rsd.items():
        assert tuple(v.shape) == tuple(sd[k].shape) and v.dtype == sd[k].dtype, k
        if k.rsplit(".", 1)[-1] in ("pedestal", "bound", "target", "scale_bound", "relative_position_index"):
            assert torch.equal(v, sd[k]), k
    with open(os.path.join(HERE, "wacnn_keys.json"), "w") as f:
        json.dump([[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in rsd.items()], f)
    model.load_state_dict(sd)
    x = U("wacnn.x", (1, 3, 256, 256), 0.0, 1.0)
    lmbda = 0.0067
    # --- eval mode
    model.eval()
    with torch.no_grad():
        o_ref = model(x)
        o = O.wacnn_forward(sd, x, keep=True)
    for k in ("y", "z"):
        check(o["likelihoods"][k], o_ref["likelihoods"][k], "wacnn.eval.lik." + k)
    check(o["x_hat"], o_ref["x_hat"], "wacnn.eval.x_hat")
    L = O.rd_loss(x, o_ref, lmbda)
    d = o["_dbg"]
    r = d["y"] - d["mu"]
    margin_y = (r - torch.floor(r) - 0.5).abs()
    med = sd["entropy_bottleneck.quantiles"][:, 0, 1].reshape(1, -1, 1, 1)
    rz = d["z"] - med
    margin_z = (rz - torch.floor(rz) - 0.5).abs()
    npix = 256 * 256
    ev = dict(
        bpp_y=(torch.log(o_ref["likelihoods"]["y"]).sum() / (-math.log(2) * npix)),
        bpp_z=(torch.log(o_ref["likelihoods"]["z"]).sum() / (-math.log(2) * npix)),
        bpp=L["bpp_loss"], mse=L["mse_loss"], loss=L["loss"],
        x_hat_crop=o_ref["x_hat"][0, :, 96:128, 160:192], x_hat_sum=o_ref["x_hat"].double().sum(),
        x_hat_abs_sum=o_ref["x_hat"].double().abs().sum(),
        lik_y=o_ref["likelihoods"]["y"], lik_z=o_ref["likelihoods"]["z"],
        y=d["y"], z=d["z"], y_hat=d["y_hat"], mu=d["mu"], scale=d["scale"],
        margin_y_min=margin_y.min(), margin_z_min=margin_z.min(),
        margin_y_lt_1e4=(margin_y < 1e-4).sum(), margin_y_lt_1e3=(margin_y < 1e-3).sum(),
    )
    print("  eval:", {k: float(v) for k, v in ev.items() if v.numel() == 1})
    # --- train mode with injected noise + backward
    nz = U("wacnn.noise_z", (1, 192, 4, 4), -0.5, 0.5)
    ny = U("wacnn.noise_y", (1, 320, 16, 16), -0.5, 0.5)
    model.train()
    orig = ref_em.EntropyModel.quantize
    state = {"i": 0}

    def q(self, inputs, m, means=None, _o=orig):
        if m != "noise":
            return _o(self, inputs, m, means)
        if isinstance(self, ref_em.EntropyBottleneck):
            return inputs + nz.transpose(0, 1).reshape(192, 1, -1)
        i = state["i"]
        state["i"] += 1
        return inputs + ny[:, 32 * i:32 * (i + 1)]
    ref_em.EntropyModel.quantize = q
    model.zero_grad()
    o_ref = model(x)
    ref_em.EntropyModel.quantize = orig
    Lr = O.rd_loss(x, o_ref, lmbda)
    Lr["loss"].backward()
    aux = model.aux_loss()
    s = leafify(sd)
    o = O.wacnn_forward(s, x, {"z": nz, "y": ny})
    Lo = O.rd_loss(x, o, lmbda)
    Lo["loss"].backward()
    check(o["x_hat"], o_ref["x_hat"], "wacnn.train.x_hat")
    check(Lo["loss"], Lr["loss"], "wacnn.train.loss")
    worst = 0.0
    gnorms, names = [], []
    for n, p in model.named_parameters():
        go = s[n].grad
        gr = p.grad if p.grad is not None else torch.zeros_like(p)
        go = go if go is not None else torch.zeros_like(p)
        dd = (go - gr).abs().max().item() / max(gr.abs().max().item(), 1e-20)
        worst = max(worst, dd)
        gnorms.append(gr.double().norm().item())
        names.append(n)
    assert worst <= 1e-5, f"oracle grads differ from reference: {worst}"
    print(f"  train: loss {Lr['loss'].item():.6f} bpp {Lr['bpp_loss'].item():.6f} mse {Lr['mse_loss'].item():.6f} "
          f"aux {aux.item():.4f} worst rel grad diff oracle-vs-ref {worst:.2e}")
    P = dict(model.named_parameters())
    tr = dict(
        t_bpp=Lr["bpp_loss"], t_mse=Lr["mse_loss"], t_loss=Lr["loss"], t_aux=aux,
        t_x_hat_crop=o_ref["x_hat"][0, :, 96:128, 160:192], t_lik_z=o_ref["likelihoods"]["z"],
        t_lik_y=o_ref["likelihoods"]["y"],
        t_grad_norms=np.array(gnorms), t_grad_names=np.array(names),
        t_total_grad_norm=np.sqrt(np.sum(np.square(gnorms))),
        t_g_ga0_w=P["g_a.0.weight"].grad, t_g_ga0_b=P["g_a.0.bias"].grad,
        t_g_ga1_beta=P["g_a.1.beta"].grad, t_g_gs8_b=P["g_s.8.bias"].grad,
        t_g_ha8_b=P["h_a.8.bias"].grad, t_g_ccm0_8_w=P["cc_mean_transforms.0.8.weight"].grad,
        t_g_lrp9_8_b=P["lrp_transforms.9.8.bias"].grad, t_g_ccs3_8_b=P["cc_scale_transforms.3.8.bias"].grad,
        t_g_eb_m0=P["entropy_bottleneck._matrix0"].grad, t_g_eb_b4=P["entropy_bottleneck._bias4"].grad,
        t_g_table_ga4=P["g_a.4.conv_b.0.attn.relative_position_bias_table"].grad,
        t_g_gs0_qkv_b=P["g_s.0.conv_b.0.attn.qkv.bias"].grad,
    )
    save("wacnn_e2e", lmbda=np.float32(lmbda), **ev, **tr)


class _InjectedDropPath(torch.nn.Module):
    """stands in for timm's DropPath while capturing fixtures: the i-th call multiplies by scales[i] (per sample)"""

    def __init__(self, scales):
        super().__init__()
        self.scales, self.i = scales, 0

    def forward(self, x):
        s = self.scales[self.i % self.scales.shape[0]]
        self.i += 1
        return x * s.view(-1, *([1] * (x.ndim - 1)))


def gen_stf_ops():
    print("stf ops")
    # SwinTransformerBlock dim 48 / 3 heads / ws 4 (hd 16), shifted and not, with DropPath scales; 12x8 map
    for tag, shift in (("swin_d48_shift2", 2), ("swin_d48_noshift", 0)):
        blk = stf_mod.SwinTransformerBlock(dim=48, num_heads=3, window_size=4, shift_size=shift, drop_path=0.1)
        sd = fill_module(blk, tag)
        B, H, Wd = 2, 12, 8
        x = U(tag + ".x", (B, H * Wd, 48), -1.5, 1.5).requires_grad_(True)
        g = U(tag + ".g", (B, H * Wd, 48), -1.0, 1.0)
        dp = torch.tensor([[1.0 / 0.9, 0.0], [1.0 / 0.9, 1.0 / 0.9]])   # [2 calls, B]
        blk.drop_path = _InjectedDropPath(dp)
        blk.H, blk.W = H, Wd
        mask = O.shift_mask(H, Wd, 4, shift) if shift else None
        y = blk(x, mask)
        pn = [n for n, _ in blk.named_parameters()]
        gs = torch.autograd.grad(y, [x] + [p_ for _, p_ in blk.named_parameters()], g)
        s = leafify(sd)
        y2 = S.swin_block(x, H, Wd, s, tag, 3, 4, shift, dp)
        gs2 = torch.autograd.grad(y2, [x] + [s[tag + "." + n] for n in pn], g)
        check(y2, y, tag + ".y", 1e-6)
        for a, b, k in zip(gs2, gs, ["x"] + pn):
            check(a, b, tag + ".grad." + k, 5e-6)
        save(tag, x=x, g=g, dp=dp, y=y, gx=gs[0], **{"g_" + n: a for n, a in zip(pn, gs[1:])},
             **{k[len(tag) + 1:]: v for k, v in sd.items() if v.dtype.is_floating_point})
    # PatchMerging dim 48 (-> 96) and PatchSplit dim 96 (-> 48), 8x12 map
    for tag, cls, dim in (("patch_merge_d48", stf_mod.PatchMerging, 48), ("patch_split_d96", stf_mod.PatchSplit, 96)):
        mod = cls(dim=dim)
        sd = fill_module(mod, tag)
        B, H, Wd = 2, 8, 12
        x = U(tag + ".x", (B, H * Wd, dim), -1.5, 1.5).requires_grad_(True)
        y = mod(x, H, Wd)
        g = U(tag + ".g", tuple(y.shape), -1.0, 1.0)
        pn = [n for n, _ in mod.named_parameters()]
        gs = torch.autograd.grad(y, [x] + [p_ for _, p_ in mod.named_parameters()], g)
        s = leafify(sd)
        f = S.patch_merging if cls is stf_mod.PatchMerging else S.patch_split
        y2 = f(x, H, Wd, s, tag)
        gs2 = torch.autograd.grad(y2, [x] + [s[tag + "." + n] for n in pn], g)
        check(y2, y, tag + ".y", 1e-6)
        for a, b, k in zip(gs2, gs, ["x"] + pn):
            check(a, b, tag + ".grad." + k, 5e-6)
        save(tag, x=x, g=g, y=y, gx=gs[0], **{"g_" + n: a for n, a in zip(pn, gs[1:])},
             **{k[len(tag) + 1:]: v for k, v in sd.items() if v.dtype.is_floating_point})


def gen_stf():
    print("stf end-to-end")
    sd = W.make_stf_state_dict()
    model = stf_mod.SymmetricalTransFormer()
    rsd = model.state_dict()
    assert list(rsd.keys()) == list(sd.keys()), "stf state-dict key order/list differs from reference"
    for k, v in rsd.items():
        assert tuple(v.shape) == tuple(sd[k].shape) and v.dtype == sd[k].dtype, k
        if k.rsplit(".", 1)[-1] in ("pedestal", "bound", "target", "scale_bound", "relative_position_index"):
            assert torch.equal(v, sd[k]), k
    with open(os.path.join(HERE, "stf_keys.json"), "w") as f:
        json.dump([[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in rsd.items()], f)
    # the reference's own load_state_dict override (stf.py:654-663) only resizes CDF buffers, then defers to nn.Module
    torch.nn.Module.load_state_dict(model, sd)
    rates = S.drop_path_rates()
    for name, mod in model.named_modules():
        if isinstance(mod, stf_mod.SwinTransformerBlock):
            r = rates[name]
            have = 0.0 if isinstance(mod.drop_path, torch.nn.Identity) else mod.drop_path.p
            assert abs(have - r) < 1e-7, (name, have, r)
    x = U("stf.x", (1, 3, 256, 256), 0.0, 1.0)
    lmbda = 0.0067
    model.eval()
    with torch.no_grad():
        o_ref = model(x)
        o = S.stf_forward(sd, x, keep=True)
    for k in ("y", "z"):
        check(o["likelihoods"][k], o_ref["likelihoods"][k], "stf.eval.lik." + k)
    check(o["x_hat"], o_ref["x_hat"], "stf.eval.x_hat")
    L = O.rd_loss(x, o_ref, lmbda)
    d = o["_dbg"]
    r = d["y"] - d["mu"]
    margin_y = (r - torch.floor(r) - 0.5).abs()
    npix = 256 * 256
    ev = dict(
        bpp_y=(torch.log(o_ref["likelihoods"]["y"]).sum() / (-math.log(2) * npix)),
        bpp_z=(torch.log(o_ref["likelihoods"]["z"]).sum() / (-math.log(2) * npix)),
        bpp=L["bpp_loss"], mse=L["mse_loss"], loss=L["loss"],
        x_hat_crop=o_ref["x_hat"][0, :, 96:128, 160:192], x_hat_sum=o_ref["x_hat"].double().sum(),
        lik_y=o_ref["likelihoods"]["y"], lik_z=o_ref["likelihoods"]["z"],
        y=d["y"], z=d["z"], y_hat=d["y_hat"], mu=d["mu"], scale=d["scale"],
        margin_y_min=margin_y.min(), margin_y_lt_1e4=(margin_y < 1e-4).sum(), margin_y_lt_1e3=(margin_y < 1e-3).sum(),
    )
    print("  eval:", {k: float(v) for k, v in ev.items() if v.numel() == 1})
    # --- train mode: injected quantisation noise and DropPath scales, B=2 on a 128x128 crop pair + backward
    B = 2
    xt = U("stf.xt", (B, 3, 128, 128), 0.0, 1.0)
    nz = U("stf.noise_z", (B, 192, 2, 2), -0.5, 0.5)
    ny = U("stf.noise_y", (B, 384, 8, 8), -0.5, 0.5)
    drops = {}
    for name, rate in rates.items():
        if rate > 0:
            keep = 1.0 - rate
            bern = (U("stf.dp." + name, (2, B), 0.0, 1.0) < keep).float()
            drops[name] = bern / keep
    assert any((v == 0).any() for v in drops.values()), "fixture should drop at least one branch"
    model.train()
    for name, mod in model.named_modules():
        if isinstance(mod, stf_mod.SwinTransformerBlock) and name in drops:
            mod.drop_path = _InjectedDropPath(drops[name])
    orig = ref_em.EntropyModel.quantize
    state = {"i": 0}

    def q(self, inputs, m, means=None, _o=orig):
        if m != "noise":
            return _o(self, inputs, m, means)
        if isinstance(self, ref_em.EntropyBottleneck):
            return inputs + nz.transpose(0, 1).reshape(192, 1, -1)
        i = state["i"]
        state["i"] += 1
        return inputs + ny[:, 32 * i:32 * (i + 1)]
    ref_em.EntropyModel.quantize = q
    model.zero_grad()
    o_ref = model(xt)
    ref_em.EntropyModel.quantize = orig
    Lr = O.rd_loss(xt, o_ref, lmbda)
    Lr["loss"].backward()
    aux = model.aux_loss()
    s = leafify(sd)
    o = S.stf_forward(s, xt, {"z": nz, "y": ny}, drops)
    Lo = O.rd_loss(xt, o, lmbda)
    Lo["loss"].backward()
    check(o["x_hat"], o_ref["x_hat"], "stf.train.x_hat", 1e-6)
    check(Lo["loss"], Lr["loss"], "stf.train.loss", 1e-6)
    worst = 0.0
    gnorms, names = [], []
    for n, p in model.named_parameters():
        go = s[n].grad
        gr = p.grad if p.grad is not None else torch.zeros_like(p)
        go = go if go is not None else torch.zeros_like(p)
        dd = (go - gr).abs().max().item() / max(gr.abs().max().item(), 1e-20)
        worst = max(worst, dd)
        gnorms.append(gr.double().norm().item())
        names.append(n)
    assert worst <= 1e-5, f"stf oracle grads differ from reference: {worst}"
    print(f"  train: loss {Lr['loss'].item():.6f} bpp {Lr['bpp_loss'].item():.6f} mse {Lr['mse_loss'].item():.6f} "
          f"aux {aux.item():.4f} worst rel grad diff oracle-vs-ref {worst:.2e}")
    P = dict(model.named_parameters())
    pick = ["patch_embed.proj.weight", "patch_embed.norm.bias", "layers.0.blocks.1.attn.relative_position_bias_table",
            "layers.0.blocks.0.norm1.weight", "layers.2.blocks.3.mlp.fc1.bias", "layers.1.downsample.reduction.weight",
            "layers.1.downsample.norm.weight", "syn_layers.0.downsample.reduction.weight",
            "syn_layers.3.blocks.1.attn.qkv.bias", "syn_layers.1.blocks.5.norm2.bias", "end_conv.0.bias",
            "end_conv.2.weight", "h_a.8.bias", "lrp_transforms.11.8.bias", "cc_scale_transforms.7.8.bias",
            "entropy_bottleneck._matrix0"]
    tr = dict(
        t_bpp=Lr["bpp_loss"], t_mse=Lr["mse_loss"], t_loss=Lr["loss"], t_aux=aux,
        t_x_hat_crop=o_ref["x_hat"][:, :, 32:64, 64:96], t_lik_z=o_ref["likelihoods"]["z"],
        t_lik_y=o_ref["likelihoods"]["y"], t_grad_norms=np.array(gnorms), t_grad_names=np.array(names),
        t_total_grad_norm=np.sqrt(np.sum(np.square(gnorms))),
        t_drop_names=np.array(list(drops.keys())), t_drops=torch.stack(list(drops.values())),
        **{"t_g_" + k: P[k].grad for k in pick},
    )
    save("stf_e2e", lmbda=np.float32(lmbda), **ev, **tr)


if __name__ == "__main__":
    if len(sys.argv) > 1:     # python make_golden.py gen_stf gen_stf_ops ...
        for name in sys.argv[1:]:
            globals()[name]()
        print("done")
        sys.exit(0)
    gen_ops()
    gen_gdn()
    gen_attention()
    gen_entropy()
    gen_wacnn()
    gen_stf_ops()
    gen_stf()
    print("done")
