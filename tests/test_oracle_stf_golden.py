"""CPU: the stf oracle (oracle/stf_oracle.py) reproduces the committed reference fixtures
(tests/golden/swin_*.npz, patch_*.npz, stf_e2e.npz -- produced by make_golden.py from the real reference's
SwinTransformerBlock / PatchMerging / PatchSplit / SymmetricalTransFormer)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import stf_oracle as S
from oracle import wacnn_oracle as O
from oracle import weights as W
from test_oracle_golden import close, grads, load


def _params(f, tag):
    skip = {"x", "g", "y", "gx", "dp"}
    return {f"{tag}.{k}": v.clone().requires_grad_(True) for k, v in f.items()
            if k not in skip and not k.startswith("g_") and isinstance(v, torch.Tensor)}


@pytest.mark.parametrize("tag,shift", [("swin_d48_shift2", 2), ("swin_d48_noshift", 0)])
def test_swin_block(golden_dir, tag, shift):
    f = load(golden_dir, tag)
    P = _params(f, tag)
    P[tag + ".attn.relative_position_index"] = O.relative_position_index(4)
    x = f["x"].clone().requires_grad_(True)
    y = S.swin_block(x, 12, 8, P, tag, 3, 4, shift, f["dp"])
    close(y, f["y"], 3e-6)
    names = [k for k in P if P[k].dtype.is_floating_point]
    gs = grads(y, f["g"], [x] + [P[k] for k in names])
    close(gs[0], f["gx"], 1e-5)
    for k, a in zip(names, gs[1:]):
        close(a, f["g_" + k[len(tag) + 1:]], 1e-5)


@pytest.mark.parametrize("tag,fn", [("patch_merge_d48", S.patch_merging), ("patch_split_d96", S.patch_split)])
def test_patch_resample(golden_dir, tag, fn):
    f = load(golden_dir, tag)
    P = _params(f, tag)
    x = f["x"].clone().requires_grad_(True)
    y = fn(x, 8, 12, P, tag)
    close(y, f["y"], 3e-6)
    names = list(P)
    gs = grads(y, f["g"], [x] + [P[k] for k in names])
    close(gs[0], f["gx"], 1e-5)
    for k, a in zip(names, gs[1:]):
        close(a, f["g_" + k[len(tag) + 1:]], 1e-5)


def test_stf_state_dict_spec(golden_dir):
    with open(os.path.join(golden_dir, "stf_keys.json")) as fh:
        ref = json.load(fh)
    sd = W.make_stf_state_dict()
    assert [k for k, _, _ in ref] == list(sd.keys())
    for k, shp, dt in ref:
        assert list(sd[k].shape) == shp and str(sd[k].dtype) == "torch." + dt, k


def test_drop_path_rates():
    r = S.drop_path_rates()
    assert r["layers.0.blocks.0"] == 0.0 and abs(r["layers.3.blocks.1"] - 0.2) < 1e-7
    # synthesis side re-indexes the same linspace with the reversed depths (stf.py:382-398)
    assert r["syn_layers.0.blocks.0"] == 0.0 and abs(r["syn_layers.3.blocks.1"] - 0.2) < 1e-7
    assert abs(r["syn_layers.1.blocks.0"] - r["layers.1.blocks.0"]) < 1e-7


def test_stf_end_to_end_eval(golden_dir):
    f = load(golden_dir, "stf_e2e")
    sd = W.make_stf_state_dict()
    x = W._u("stf.x", (1, 3, 256, 256), 0.0, 1.0)
    with torch.no_grad():
        o = S.stf_forward(sd, x, keep=True)
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


def stf_train_inputs(f):
    B = 2
    xt = W._u("stf.xt", (B, 3, 128, 128), 0.0, 1.0)
    noise = {"z": W._u("stf.noise_z", (B, 192, 2, 2), -0.5, 0.5), "y": W._u("stf.noise_y", (B, 384, 8, 8), -0.5, 0.5)}
    drops = {str(n): f["t_drops"][i] for i, n in enumerate(f["t_drop_names"])}
    return xt, noise, drops


def test_stf_end_to_end_train(golden_dir):
    f = load(golden_dir, "stf_e2e")
    sd = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and v.numel() else v)
          for k, v in W.make_stf_state_dict().items()}
    xt, noise, drops = stf_train_inputs(f)
    o = S.stf_forward(sd, xt, noise, drops)
    L = O.rd_loss(xt, o, float(f["lmbda"]))
    L["loss"].backward()
    assert abs(L["loss"].item() - f["t_loss"].item()) <= 2e-5 * abs(f["t_loss"].item())
    close(o["x_hat"][:, :, 32:64, 64:96], f["t_x_hat_crop"], 2e-5)
    close(o["likelihoods"]["z"], f["t_lik_z"], 2e-5)
    names = [str(n) for n in f["t_grad_names"]]
    total = float(f["t_total_grad_norm"])
    for n, gn in zip(names, f["t_grad_norms"].tolist()):
        g = sd[n].grad
        have = 0.0 if g is None else g.double().norm().item()
        assert abs(have - gn) <= 1e-4 * total, n
    for k in f:
        if k.startswith("t_g_"):
            close(sd[k[4:]].grad, f[k], 5e-5)
    assert abs(O.eb_aux_loss(sd).item() - f["t_aux"].item()) <= 1e-5 * f["t_aux"].item()
