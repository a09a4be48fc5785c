"""Emit tests/golden/stf6_e2e.npz + stf6_keys.json from the REAL ``SymmetricalTransFormer3`` (compressai/models/stf6.py;
this container only, needs /root/reference): formula weights, eval forward on [1,3,128,128], train forward + backward
on [2,3,128,128] with injected quantisation noise and DropPath scales; asserts oracle/stf6_oracle.py equals the
reference (outputs bit for bit, gradients to 1e-5 of each tensor's max).  Usage: python tests/golden/make_golden_stf6.py"""
import importlib
import json
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
from _ref_loader import load_reference  # noqa: E402
from oracle import stf6_oracle as S6  # noqa: E402
from oracle import wacnn_oracle as O  # noqa: E402
from oracle import weights as W  # noqa: E402
from make_golden import _InjectedDropPath, check, leafify, save, U  # noqa: E402


def main():
    torch.set_num_threads(8)
    load_reference()
    stf6 = importlib.import_module("compressai.models.stf6")
    ref_em = importlib.import_module("compressai.entropy_models.entropy_models")
    sd = W.make_stf6_state_dict()
    model = stf6.SymmetricalTransFormer3()
    rsd = model.state_dict()
    assert list(rsd.keys()) == list(sd.keys()), "stf6 state-dict key order/list differs from reference"
    for k, v in rsd.items():
        assert tuple(v.shape) == tuple(sd[k].shape) and v.dtype == sd[k].dtype, k
    with open(os.path.join(HERE, "stf6_keys.json"), "w") as f:
        json.dump([[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in rsd.items()], f)
    torch.nn.Module.load_state_dict(model, sd)
    rates = S6.drop_path_rates()
    nblk = 0
    for name, mod in model.named_modules():
        if isinstance(mod, stf6.SwinTransformerBlock) and name in rates:
            have = 0.0 if isinstance(mod.drop_path, torch.nn.Identity) else mod.drop_path.p
            assert abs(have - rates[name]) < 1e-7, (name, have, rates[name])
            nblk += 1
    assert nblk == 24 + 24 * 12
    lmbda = 0.0067
    x = U("stf6.x", (1, 3, 128, 128), 0.0, 1.0)
    model.eval()
    with torch.no_grad():
        o_ref = model(x)
        o = S6.stf6_forward(sd, x, keep=True)
    for k in ("y", "z"):
        check(o["likelihoods"][k], o_ref["likelihoods"][k], "stf6.eval.lik." + k)
    check(o["x_hat"], o_ref["x_hat"], "stf6.eval.x_hat")
    L = O.rd_loss(x, o_ref, lmbda)
    d = o["_dbg"]
    r = d["y_zz"] - d["mu"]
    margin = (r - torch.floor(r) - 0.5).abs()
    refined = (d["mu"].abs().mean()).item()
    ev = dict(bpp=L["bpp_loss"], mse=L["mse_loss"], loss=L["loss"], x_hat=o_ref["x_hat"],
              lik_y=o_ref["likelihoods"]["y"], lik_z=o_ref["likelihoods"]["z"], y=d["y"], z=d["z"], y_hat=d["y_hat"],
              mu=d["mu"], scale=d["scale"], margin_y_min=margin.min(), margin_y_lt_1e4=(margin < 1e-4).sum())
    print("  eval:", {k: float(v) for k, v in ev.items() if v.numel() == 1}, "mean |mu|", refined)
    # --- train mode
    B = 2
    xt = U("stf6.xt", (B, 3, 128, 128), 0.0, 1.0)
    nz = U("stf6.noise_z", (B, 192, 2, 2), -0.5, 0.5)
    ny = U("stf6.noise_y", (B, 24, 64, 4, 4), -0.5, 0.5)
    drops = {}
    for name, rate in rates.items():
        if rate > 0:
            keep = 1.0 - rate
            drops[name] = (U("stf6.dp." + name, (2, B), 0.0, 1.0) < keep).float() / keep
    model.train()
    for name, mod in model.named_modules():
        if isinstance(mod, stf6.SwinTransformerBlock) and name in drops:
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
        return inputs + ny[:, i]
    ref_em.EntropyModel.quantize = q
    model.zero_grad()
    o_ref = model(xt)
    ref_em.EntropyModel.quantize = orig
    assert state["i"] == 24
    Lr = O.rd_loss(xt, o_ref, lmbda)
    Lr["loss"].backward()
    aux = model.aux_loss()
    s = leafify(sd)
    o = S6.stf6_forward(s, xt, {"z": nz, "y": ny}, drops)
    Lo = O.rd_loss(xt, o, lmbda)
    Lo["loss"].backward()
    check(o["x_hat"], o_ref["x_hat"], "stf6.train.x_hat", 1e-6)
    check(Lo["loss"], Lr["loss"], "stf6.train.loss", 1e-6)
    worst, gnorms, names, idle = 0.0, [], [], 0
    for n, p in model.named_parameters():
        go = s[n].grad
        if p.grad is None:
            idle += 1
            assert go is None or float(go.abs().max()) == 0.0, n
        gr = p.grad if p.grad is not None else torch.zeros_like(p)
        go = go if go is not None else torch.zeros_like(p)
        worst = max(worst, (go - gr).abs().max().item() / max(gr.abs().max().item(), 1e-20))
        gnorms.append(gr.double().norm().item())
        names.append(n)
    assert worst <= 1e-5, f"stf6 oracle grads differ from reference: {worst}"
    print(f"  train: loss {Lr['loss'].item():.6f} bpp {Lr['bpp_loss'].item():.6f} mse {Lr['mse_loss'].item():.6f} "
          f"aux {aux.item():.4f} worst rel grad diff oracle-vs-ref {worst:.2e}; {idle} idle parameters (sigma/LRP_Swin)")
    P = dict(model.named_parameters())
    pick = ["patch_embed.proj.weight", "layers.0.blocks.1.attn.relative_position_bias_table",
            "syn_layers.3.blocks.1.attn.qkv.bias", "end_conv.2.weight", "h_a.8.bias",
            "mu_Swin.0.0.blocks.0.attn.qkv.weight", "mu_Swin.7.1.blocks.3.mlp.fc1.bias",
            "mu_Swin.23.3.blocks.1.attn.relative_position_bias_table", "mu_Swin.16.2.blocks.0.norm1.weight",
            "cc_mean_transforms2.0.8.weight", "cc_mean_transforms2.17.0.bias", "cc_scale_transforms2.9.8.bias",
            "lrp_transforms2.23.8.bias", "lrp_transforms2.3.6.weight", "entropy_bottleneck._matrix0"]
    tr = dict(t_bpp=Lr["bpp_loss"], t_mse=Lr["mse_loss"], t_loss=Lr["loss"], t_aux=aux,
              t_x_hat_crop=o_ref["x_hat"][:, :, 32:64, 64:96], t_lik_z=o_ref["likelihoods"]["z"],
              t_lik_y=o_ref["likelihoods"]["y"], t_grad_norms=np.array(gnorms), t_grad_names=np.array(names),
              t_total_grad_norm=np.sqrt(np.sum(np.square(gnorms))),
              t_drop_names=np.array(list(drops.keys())), t_drops=torch.stack(list(drops.values())),
              **{"t_g_" + k: P[k].grad for k in pick})
    save("stf6_e2e", lmbda=np.float32(lmbda), **ev, **tr)


if __name__ == "__main__":
    main()
