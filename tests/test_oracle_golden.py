"""CPU: the oracle (oracle/wacnn_oracle.py) reproduces the committed reference fixtures.

The fixtures in tests/golden/*.npz were produced by tests/golden/make_golden.py from the real
reference modules; these tests pin the oracle to them with no reference present."""
import json
import math
import os

import numpy as np
import pytest
import torch

from oracle import wacnn_oracle as O
from oracle import weights as W

TOL = 2e-6  # CPU summation-order noise between thread counts (4.7e-7 observed)


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)
    return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in "fiu" else z[k]) for k in z.files}


def close(a, b, tol=TOL):
    ref = max(b.abs().max().item(), 1e-30) if b.numel() else 1.0
    d = (a - b).abs().max().item() if b.numel() else 0.0
    assert d <= tol * ref, f"maxdiff {d} vs ref max {ref}"


def grads(out, g, ts):
    gs = torch.autograd.grad(out, ts, g, allow_unused=True, retain_graph=True)
    return [torch.zeros_like(t) if a is None else a for a, t in zip(gs, ts)]


def test_ops(golden_dir):
    f = load(golden_dir, "ops")
    x = f["ste_x"].clone().requires_grad_(True)
    y = O.ste_round(x)
    assert torch.equal(y, f["ste_y"])
    assert torch.equal(grads(y, torch.arange(12, dtype=torch.float32), [x])[0], f["ste_gx"])
    x = f["lb_x"].clone().requires_grad_(True)
    y = O.lower_bound(x, 0.11)
    assert torch.equal(y, f["lb_y"])
    assert torch.equal(grads(y, f["lb_g"], [x])[0], f["lb_gx"])
    x = f["nn_x"].clone().requires_grad_(True)
    y = O.nonneg_param(x, 1e-6)
    assert torch.equal(y, f["nn_y"])
    assert torch.equal(grads(y, f["nn_g"], [x])[0], f["nn_gx"])


@pytest.mark.parametrize("name,inverse", [("gdn", False), ("igdn", True)])
def test_gdn(golden_dir, name, inverse):
    f = load(golden_dir, name)
    x, b, g = (f[k].clone().requires_grad_(True) for k in ("x", "beta", "gamma"))
    y = O.gdn(x, b, g, inverse)
    close(y, f["y"])
    gx, gb, gg = grads(y, f["g"], [x, b, g])
    close(gx, f["gx"]); close(gb, f["gbeta"]); close(gg, f["ggamma"])


@pytest.mark.parametrize("tag,ws,shift", [("wa_d64_ws8", 8, 4), ("wa_d80_ws4", 4, 2), ("wa_d64_ws8_noshift", 8, 0)])
def test_window_attention(golden_dir, tag, ws, shift):
    f = load(golden_dir, tag)
    keys = ["attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias", "attn.relative_position_bias_table"]
    sd = {tag + "." + k: f[k].clone().requires_grad_(True) for k in keys}
    x = f["x"].clone().requires_grad_(True)
    y = O.win_based_attention(x, sd, tag, 8, ws, shift)
    close(y, f["y"])
    gs = grads(y, f["g"], [x] + [sd[tag + "." + k] for k in keys])
    for a, k in zip(gs, ["gx", "g_qkv_w", "g_qkv_b", "g_proj_w", "g_proj_b", "g_table"]):
        close(a, f[k], 5e-6)


def _fill_like_generator(prefix, names_shapes):
    """Re-derive the formula parameters the generator used for module fixtures (fill_module)."""
    out = {}
    for k, shp in names_shapes:
        key = prefix + "." + k
        leaf = k.rsplit(".", 1)[-1]
        if leaf == "relative_position_bias_table":
            t = W._u(key, shp, -0.5, 0.5)
        elif leaf == "weight":
            b = 1.0 / math.sqrt(int(np.prod(shp[1:])))
            t = W._u(key, shp, -b, b) * 1.7
        elif leaf == "bias":
            t = W._u(key, shp, -0.1, 0.1)
        else:
            raise KeyError(k)
        out[key] = t.float().reshape(shp)
    return out


def gate_param_shapes(dim, ws):
    spec = {}
    W._gate_spec(spec, "g", dim, ws)
    return [(k[2:], v) for k, v in spec.items() if not isinstance(v[0], str)]


def test_gate(golden_dir):
    tag, dim, ws, shift = "gate_d64_ws8", 64, 8, 4
    f = load(golden_dir, tag)
    sd = _fill_like_generator(tag, gate_param_shapes(dim, ws))
    sd = {k: v.requires_grad_(True) for k, v in sd.items()}
    x = f["x"].clone().requires_grad_(True)
    y = O.win_attention_gate(x, sd, tag, 8, ws, shift)
    close(y, f["y"])
    names = [str(n) for n in f["grad_names"]]
    gs = grads(y, f["g"], [x] + [sd[tag + "." + n] for n in names])
    close(gs[0], f["gx"], 5e-6)
    gn = torch.stack([t.norm() for t in gs[1:]])
    close(gn, f["grad_norms"], 2e-5)
    close(gs[1 + names.index("conv_a.0.conv.0.weight")], f["g_first_conv_w"], 1e-5)
    close(gs[1 + names.index("conv_b.4.bias")], f["g_last_conv_b"], 1e-5)


def test_entropy_bottleneck(golden_dir):
    f = load(golden_dir, "entropy_bottleneck")
    pn = [k[1:] for k in f if k.startswith("p_") or k == "pquantiles"]
    for mode in ("eval", "train"):
        sd = {"entropy_bottleneck." + n: f["p" + n].clone().requires_grad_(True) for n in pn}
        z = f["z"].clone().requires_grad_(True)
        zt, lik = O.eb_likelihood(z, sd, "entropy_bottleneck", f["noise"] if mode == "train" else None)
        close(zt, f[mode + "_zt"]); close(lik, f[mode + "_lik"])
        for gname in ("g", "gpos"):
            gs = grads(lik, f[gname], [z] + [sd["entropy_bottleneck." + n] for n in pn])
            close(gs[0], f[f"{mode}_{gname}_gz"], 1e-5)
            for n, a in zip(pn, gs[1:]):
                close(a, f[f"{mode}_{gname}_grad{n}"], 1e-5)
    sd = {"entropy_bottleneck." + n: f["p" + n].clone().requires_grad_(True) for n in pn}
    aux = O.eb_aux_loss(sd)
    close(aux, f["aux"])
    close(grads(aux, torch.tensor(1.0), [sd["entropy_bottleneck.quantiles"]])[0], f["aux_gq"])


def test_gaussian_conditional(golden_dir):
    f = load(golden_dir, "gaussian_conditional")
    for mode in ("eval", "train"):
        y, mu, sc = (f[k].clone().requires_grad_(True) for k in ("y", "mu", "sc"))
        yt, lik = O.gaussian_likelihood(y, sc, mu, f["noise"] if mode == "train" else None)
        assert torch.equal(yt, f[mode + "_yt"])
        close(lik, f[mode + "_lik"])
        assert (f[mode + "_lik"] <= 1.0000001e-9).any() and (f["sc"] < 0.11).any()  # corners are exercised
        for gname in ("g", "gpos"):
            gy, gm, gs = grads(lik, f[gname], [y, mu, sc])
            close(gy, f[f"{mode}_{gname}_gy"], 1e-5)
            close(gm, f[f"{mode}_{gname}_gmu"], 1e-5)
            close(gs, f[f"{mode}_{gname}_gsc"], 1e-5)


def test_state_dict_spec(golden_dir):
    with open(os.path.join(golden_dir, "wacnn_keys.json")) as fh:
        ref = json.load(fh)
    sd = W.make_wacnn_state_dict()
    assert [k for k, _, _ in ref] == list(sd.keys())
    for k, shp, dt in ref:
        assert list(sd[k].shape) == shp and str(sd[k].dtype) == "torch." + dt, k
    assert sum(v.numel() for k, v in sd.items() if v.dtype.is_floating_point and k.rsplit(".", 1)[-1] not in
               ("pedestal", "bound", "target", "scale_bound", "scale_table")) == 75235779


def test_wacnn_end_to_end_eval(golden_dir):
    f = load(golden_dir, "wacnn_e2e")
    sd = W.make_wacnn_state_dict()
    x = W._u("wacnn.x", (1, 3, 256, 256), 0.0, 1.0)
    with torch.no_grad():
        o = O.wacnn_forward(sd, x, keep=True)
    d = o["_dbg"]
    close(d["y"], f["y"], 5e-6)
    close(d["z"], f["z"], 5e-6)
    flips = (torch.round(d["y"] - d["mu"]) != torch.round(f["y"] - f["mu"])).sum().item()
    if flips == 0:
        close(o["likelihoods"]["y"], f["lik_y"], 2e-5)
        close(o["x_hat"][0, :, 96:128, 160:192], f["x_hat_crop"], 2e-5)
    L = O.rd_loss(x, o, float(f["lmbda"]))
    assert abs(L["bpp_loss"].item() - f["bpp"].item()) <= 1e-4 * f["bpp"].item() + 3e-4 * flips
    assert abs(L["mse_loss"].item() - f["mse"].item()) <= 1e-4 * f["mse"].item() + 1e-3 * flips
    assert O.psnr(L["mse_loss"].item()) == pytest.approx(-10 * math.log10(f["mse"].item()), abs=1e-3 + 0.01 * flips)


def test_round_override_is_transparent():
    """tests/_parity.py: an oracle run that ADOPTS rounding decisions equal to its own reproduces the free run (values bit
    for bit; gradients to 1e-6 relative -- the threaded CPU conv backward does not fix its summation order between two
    runs), and a changed decision moves y_hat by exactly one step"""
    sd = W.make_wacnn_state_dict()
    x = W._u("ro.x", (1, 3, 64, 64), 0.0, 1.0)
    noise = {"z": W._u("ro.nz", (1, 192, 1, 1), -0.5, 0.5), "y": W._u("ro.ny", (1, 320, 4, 4), -0.5, 0.5)}

    def run(ro):
        s = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and v.numel() else v) for k, v in sd.items()}
        o = O.wacnn_forward(s, x, noise, keep=True, round_override=ro)
        O.rd_loss(x, o)["loss"].backward()
        return o, s
    o0, s0 = run(None)
    d = o0["_dbg"]
    med = sd["entropy_bottleneck.quantiles"][:, :, 1:2].reshape(1, -1, 1, 1)
    ro = {"y": torch.round(d["y"] - d["mu"]).detach(), "z": torch.round(d["z"] - med).detach()}
    o1, s1 = run(ro)
    assert torch.equal(o0["x_hat"], o1["x_hat"]) and torch.equal(o0["likelihoods"]["y"], o1["likelihoods"]["y"])
    for k in ("g_a.0.weight", "g_s.8.bias", "lrp_transforms.9.8.weight", "h_a.0.weight"):
        torch.testing.assert_close(s0[k].grad, s1[k].grad, rtol=1e-6, atol=1e-9 + 1e-6 * float(s0[k].grad.abs().max()), msg=k)
    ro2 = {"y": ro["y"].clone(), "z": ro["z"]}
    ro2["y"][0, 300, 1, 2] += 1.0          # slice 9: nothing downstream re-rounds
    o2, _ = run(ro2)
    dy = (o2["_dbg"]["y_hat"] - d["y_hat"]).detach()
    assert abs(dy[0, 300, 1, 2].item() - 1.0) < 0.6 and int((dy.abs() > 1e-3).sum()) <= 32 * 16
