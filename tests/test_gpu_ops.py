"""GPU parity: every HIP op (through the Python mirror -> ctypes -> C ABI) against the CPU oracle.

Tolerances: all arithmetic is IEEE f32; the f32 MFMA is an exact fmaf chain, so differences to the CPU
reference are summation-order noise.  ``TOL`` is relative to the reference tensor's max magnitude.
"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import wacnn_oracle as O
from oracle import weights as W

pytestmark = pytest.mark.gpu
TOL = 3e-5


def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def U(key, shape, lo=-1.0, hi=1.0):
    return W._u(key, shape, lo, hi)


def close(a, b, tol=TOL, what=""):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    ref = max(b.abs().max().item(), 1e-30)
    d = (a - b).abs().max().item()
    assert math.isfinite(d) and d <= tol * ref, f"{what}: maxdiff {d:.3e} vs ref max {ref:.3e} (rel {d/ref:.2e})"


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)
    return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in "fiu" else z[k]) for k in z.files}


# ------------------------------------------------------------------------------------------ conv family
CONV_CASES = [
    # name, N, Cin, H, W, Cout, k, stride, transposed
    ("c5s2_3_192", 2, 3, 32, 32, 192, 5, 2, False),
    ("c5s2_192_192", 2, 192, 16, 16, 192, 5, 2, False),
    ("c5s2_40_320_odd", 1, 40, 24, 40, 320, 5, 2, False),
    ("c3s1_96_96", 2, 96, 16, 16, 96, 3, 1, False),
    ("c3s2_72_56", 3, 72, 8, 8, 56, 3, 2, False),
    ("c3s1_tiny4x4", 4, 24, 4, 4, 48, 3, 1, False),
    ("c3s1_176", 2, 64, 16, 16, 176, 3, 1, False),
    ("c1_192_96", 2, 192, 16, 16, 96, 1, 1, False),
    ("c1_20_576", 1, 20, 8, 24, 576, 1, 1, False),
    ("t5s2_320_192", 2, 320, 8, 8, 192, 5, 2, True),
    ("t5s2_64_3", 2, 64, 16, 16, 3, 5, 2, True),
    ("t5s2_24_40_odd", 1, 24, 6, 10, 40, 5, 2, True),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_fwd_bwd(case):
    from icm_amd import layers
    name, N, Cin, H, Wd, Cout, k, s, tr = case
    d = dev()
    if tr:
        m = layers.deconv(Cin, Cout, kernel_size=k, stride=s)
    else:
        m = layers.Conv2d(Cin, Cout, kernel_size=k, stride=s, padding=k // 2)
    w = U(name + ".w", m.weight.shape, -0.2, 0.2)
    b = U(name + ".b", m.bias.shape, -0.5, 0.5)
    x = U(name + ".x", (N, Cin, H, Wd), -1.0, 1.0)
    # reference
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    if tr:
        yr = F.conv_transpose2d(xr, wr, br, stride=s, padding=k // 2, output_padding=s - 1)
    else:
        yr = F.conv2d(xr, wr, br, stride=s, padding=k // 2)
    g = U(name + ".g", yr.shape, -1.0, 1.0)
    gxr, gwr, gbr = torch.autograd.grad(yr, [xr, wr, br], g)
    # HIP
    m = m.to(d)
    with torch.no_grad():
        m.weight.copy_(w)
        m.bias.copy_(b)
    xg = x.to(d).requires_grad_(True)
    y = m(xg)
    close(y, yr, what="y")
    gx, gw, gb = torch.autograd.grad(y, [xg, m.weight, m.bias], g.to(d))
    close(gx, gxr, what="dx")
    close(gw, gwr, what="dw")
    close(gb, gbr, what="db")


def test_conv_every_tile_config():
    """every template instantiation of the implicit-GEMM kernel gives the same answer"""
    import ctypes
    from icm_amd import _lib, layers
    d = dev()
    lib = _lib.lib()
    lib.icm_debug_force_conv_cfg.argtypes = [ctypes.c_int]
    lib.icm_debug_force_conv_cfg.restype = None
    m = layers.Conv2d(40, 200, kernel_size=3, stride=1, padding=1)
    w = U("cfg.w", m.weight.shape, -0.2, 0.2)
    b = U("cfg.b", m.bias.shape, -0.5, 0.5)
    x = U("cfg.x", (3, 40, 20, 36), -1.0, 1.0)
    yr = F.conv2d(x, w, b, padding=1)
    m = m.to(d)
    with torch.no_grad():
        m.weight.copy_(w)
        m.bias.copy_(b)
    try:
        for cfg in range(13):   # 10 tile shapes + 3 intra-workgroup split-K variants
            lib.icm_debug_force_conv_cfg(cfg)
            with torch.no_grad():
                y = m(x.to(d))
            close(y, yr, what=f"cfg{cfg}")
        # 1x1 and stride-2 5x5 through the split-K variants as well (odd channel counts: K tails per split differ)
        for (ci, co, k, s_) in ((13, 70, 1, 1), (21, 33, 5, 2), (8, 32, 3, 1)):
            m2 = layers.Conv2d(ci, co, kernel_size=k, stride=s_, padding=k // 2)
            w2 = U(f"cfg2.w{ci}", m2.weight.shape, -0.2, 0.2)
            b2 = U(f"cfg2.b{ci}", m2.bias.shape, -0.5, 0.5)
            x2 = U(f"cfg2.x{ci}", (2, ci, 18, 24), -1.0, 1.0)
            yr2 = F.conv2d(x2, w2, b2, stride=s_, padding=k // 2)
            m2 = m2.to(d)
            with torch.no_grad():
                m2.weight.copy_(w2)
                m2.bias.copy_(b2)
            for cfg in (7, 10, 11, 12):
                lib.icm_debug_force_conv_cfg(cfg)
                with torch.no_grad():
                    y2 = m2(x2.to(d))
                close(y2, yr2, what=f"cfg{cfg} {ci}->{co} k{k}")
    finally:
        lib.icm_debug_force_conv_cfg(-1)


def test_conv_argument_errors():
    from icm_amd import layers
    d = dev()
    m = layers.Conv2d(8, 8, kernel_size=3, padding=1).to(d)
    with pytest.raises(ValueError):
        m(torch.zeros(1, 4, 8, 8, device=d))


# ------------------------------------------------------------------------------------------ GDN
@pytest.mark.parametrize("name,inverse", [("gdn", False), ("igdn", True)])
def test_gdn_golden(golden_dir, name, inverse):
    from icm_amd import layers
    f = load(golden_dir, name)
    d = dev()
    m = layers.GDN(f["x"].shape[1], inverse=inverse).to(d)
    with torch.no_grad():
        m.beta.copy_(f["beta"])
        m.gamma.copy_(f["gamma"])
    x = f["x"].to(d).requires_grad_(True)
    y = m(x)
    close(y, f["y"], what="y")
    gx, gb, gg = torch.autograd.grad(y, [x, m.beta, m.gamma], f["g"].to(d))
    close(gx, f["gx"], what="gx")
    close(gb, f["gbeta"], what="gbeta")
    close(gg, f["ggamma"], what="ggamma")


@pytest.mark.parametrize("inverse", [False, True])
def test_gdn_192_vs_oracle(inverse):
    from icm_amd import layers
    d = dev()
    C = 192
    beta = math.sqrt(1 + W.PEDESTAL) + U("g192.beta", (C,), -0.2, 0.3)
    gamma = torch.sqrt(0.1 * torch.eye(C) + W.PEDESTAL) + U("g192.gamma", (C, C), -0.004, 0.012)
    x = U("g192.x", (2, C, 32, 32), -2, 2)
    g = U("g192.g", (2, C, 32, 32), -1, 1)
    xr, br, gr = (t.clone().requires_grad_(True) for t in (x, beta, gamma))
    yr = O.gdn(xr, br, gr, inverse)
    gxr, gbr, ggr = torch.autograd.grad(yr, [xr, br, gr], g)
    m = layers.GDN(C, inverse=inverse).to(d)
    with torch.no_grad():
        m.beta.copy_(beta)
        m.gamma.copy_(gamma)
    xg = x.to(d).requires_grad_(True)
    y = m(xg)
    close(y, yr, what="y")
    gx, gb, gg = torch.autograd.grad(y, [xg, m.beta, m.gamma], g.to(d))
    close(gx, gxr, what="gx")
    close(gb, gbr, what="gbeta")
    close(gg, ggr, what="ggamma")


# ------------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("tag,dim,ws,shift", [("wa_d64_ws8", 64, 8, 4), ("wa_d80_ws4", 80, 4, 2),
                                              ("wa_d64_ws8_noshift", 64, 8, 0)])
def test_window_attention_golden(golden_dir, tag, dim, ws, shift):
    from icm_amd import layers
    f = load(golden_dir, tag)
    d = dev()
    m = layers.WinBasedAttention(dim=dim, num_heads=8, window_size=ws, shift_size=shift).to(d)
    with torch.no_grad():
        m.attn.qkv.weight.copy_(f["attn.qkv.weight"])
        m.attn.qkv.bias.copy_(f["attn.qkv.bias"])
        m.attn.proj.weight.copy_(f["attn.proj.weight"])
        m.attn.proj.bias.copy_(f["attn.proj.bias"])
        m.attn.relative_position_bias_table.copy_(f["attn.relative_position_bias_table"])
    x = f["x"].to(d).requires_grad_(True)
    y = m(x)
    close(y, f["y"], what="y")
    ps = [m.attn.qkv.weight, m.attn.qkv.bias, m.attn.proj.weight, m.attn.proj.bias, m.attn.relative_position_bias_table]
    gs = torch.autograd.grad(y, [x] + ps, f["g"].to(d))
    for a, k in zip(gs, ["gx", "g_qkv_w", "g_qkv_b", "g_proj_w", "g_proj_b", "g_table"]):
        close(a, f[k], 5e-5, what=k)


@pytest.mark.parametrize("dim,ws,shift,hw", [(192, 8, 4, 16), (320, 4, 2, 8)])
def test_attention_gate_vs_oracle(dim, ws, shift, hw):
    from icm_amd import layers
    d = dev()
    tag = f"gate{dim}"
    m = layers.Win_noShift_Attention(dim=dim, num_heads=8, window_size=ws, shift_size=shift)
    sd = {}
    for k, v in m.state_dict().items():
        leaf = k.rsplit(".", 1)[-1]
        if not v.dtype.is_floating_point:
            sd[k] = v
        elif leaf == "relative_position_bias_table":
            sd[k] = U(tag + k, v.shape, -0.5, 0.5)
        elif leaf == "weight":
            bnd = 1.0 / math.sqrt(int(np.prod(v.shape[1:])))
            sd[k] = U(tag + k, v.shape, -bnd, bnd) * 1.7
        else:
            sd[k] = U(tag + k, v.shape, -0.1, 0.1)
    m.load_state_dict(sd)
    x = U(tag + ".x", (2, dim, hw, hw), -1.5, 1.5)
    g = U(tag + ".g", (2, dim, hw, hw), -1, 1)
    osd = {"p." + k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    yr = O.win_attention_gate(xr, osd, "p", 8, ws, shift)
    names = [n for n, _ in m.named_parameters()]
    grs = torch.autograd.grad(yr, [xr] + [osd["p." + n] for n in names], g)
    m = m.to(d)
    xg = x.to(d).requires_grad_(True)
    y = m(xg)
    close(y, yr, what="y")
    gs = torch.autograd.grad(y, [xg] + [p for _, p in m.named_parameters()], g.to(d))
    for a, b, n in zip(gs, grs, ["x"] + names):
        close(a, b, 1e-4, what="grad " + n)


# ------------------------------------------------------------------------------------------ entropy models
def test_entropy_bottleneck_golden(golden_dir):
    from icm_amd.entropy_models import EntropyBottleneck
    f = load(golden_dir, "entropy_bottleneck")
    d = dev()
    pn = [k[1:] for k in f if k.startswith("p_") or k == "pquantiles"]
    eb = EntropyBottleneck(f["z"].shape[1]).to(d)
    with torch.no_grad():
        for n in pn:
            getattr(eb, n).copy_(f["p" + n])
    for mode in ("eval", "train"):
        eb.train(mode == "train")
        eb.inject_noise(f["noise"].to(d) if mode == "train" else None)
        for gname in ("g", "gpos"):
            z = f["z"].to(d).requires_grad_(True)
            zt, lik = eb(z)
            close(zt, f[mode + "_zt"], 1e-6, what=mode + " zt")
            close(lik, f[mode + "_lik"], what=mode + " lik")
            ps = [getattr(eb, n) for n in pn]
            gs = torch.autograd.grad(lik, [z] + ps, f[gname].to(d), allow_unused=True)
            gs = [torch.zeros_like(t) if a is None else a for a, t in zip(gs, [z] + ps)]
            close(gs[0], f[f"{mode}_{gname}_gz"], 5e-5, what=f"{mode} {gname} gz")
            for n, a in zip(pn, gs[1:]):
                ref = f[f"{mode}_{gname}_grad{n}"]
                if ref.abs().max() == 0:
                    assert a.abs().max().item() == 0, n
                else:
                    close(a, ref, 1e-4, what=f"{mode} {gname} grad {n}")
    aux = eb.loss()
    assert abs(aux.item() - f["aux"].item()) <= 1e-5 * abs(f["aux"].item())
    (gq,) = torch.autograd.grad(aux, [eb.quantiles])
    close(gq, f["aux_gq"], 1e-5, what="aux gq")


def test_gaussian_conditional_golden(golden_dir):
    from icm_amd.entropy_models import GaussianConditional
    f = load(golden_dir, "gaussian_conditional")
    d = dev()
    gc = GaussianConditional(None).to(d)
    for mode in ("eval", "train"):
        gc.train(mode == "train")
        gc.inject_noise(f["noise"].to(d) if mode == "train" else None)
        for gname in ("g", "gpos"):
            y, mu, sc = (f[k].to(d).requires_grad_(True) for k in ("y", "mu", "sc"))
            yt, lik = gc(y, sc, mu)
            close(yt, f[mode + "_yt"], 1e-6, what="yt")
            ref = f[mode + "_lik"]
            dd = (lik.cpu() - ref).abs()
            assert (dd <= 2e-6 + 2e-5 * ref).all(), f"{mode} lik maxdiff {dd.max().item()}"
            gy, gm, gs = torch.autograd.grad(lik, [y, mu, sc], f[gname].to(d), allow_unused=True)
            gy = torch.zeros_like(y) if gy is None else gy
            gm = torch.zeros_like(mu) if gm is None else gm
            close(gy, f[f"{mode}_{gname}_gy"], 5e-5, what=f"{mode} {gname} gy")
            close(gm, f[f"{mode}_{gname}_gmu"], 5e-5, what=f"{mode} {gname} gmu")
            close(gs, f[f"{mode}_{gname}_gsc"], 5e-5, what=f"{mode} {gname} gsc")


def test_gaussian_conditional_errors():
    from icm_amd.entropy_models import GaussianConditional
    with pytest.raises(ValueError):
        GaussianConditional([3.0, 1.0])
    with pytest.raises(ValueError):
        GaussianConditional(None, scale_bound=0.0)
    with pytest.raises(ValueError):
        GaussianConditional("x")


def test_ops_golden(golden_dir):
    from icm_amd import ops
    f = load(golden_dir, "ops")
    d = dev()
    x = f["ste_x"].to(d).requires_grad_(True)
    y = ops.ste_round(x)
    assert torch.equal(y.cpu(), f["ste_y"])
    (gx,) = torch.autograd.grad(y, [x], torch.arange(12, dtype=torch.float32, device=d))
    assert torch.equal(gx.cpu(), f["ste_gx"])
    lb = ops.LowerBound(0.11).to(d)
    x = f["lb_x"].to(d).requires_grad_(True)
    y = lb(x)
    assert torch.equal(y.cpu(), f["lb_y"])
    (gx,) = torch.autograd.grad(y, [x], f["lb_g"].to(d))
    assert torch.equal(gx.cpu(), f["lb_gx"])
    par = ops.NonNegativeParametrizer(minimum=1e-6).to(d)
    x = f["nn_x"].to(d).requires_grad_(True)
    y = par(x)
    close(y, f["nn_y"], 1e-6, what="nonneg")
    (gx,) = torch.autograd.grad(y, [x], f["nn_g"].to(d))
    close(gx, f["nn_gx"], 1e-6, what="nonneg grad")


# ------------------------------------------------------------------------------------------ loss / optimiser
def test_rd_loss_and_adam():
    from icm_amd import _lib as L
    from icm_amd.losses import RateDistortionLoss
    d = dev()
    x = U("rd.x", (2, 3, 64, 64), 0, 1)
    xh = (x + U("rd.e", x.shape, -0.3, 0.3)).requires_grad_(True)
    ly = U("rd.ly", (2, 320, 4, 4), 1e-4, 1.0).requires_grad_(True)
    lz = U("rd.lz", (2, 192, 1, 1), 1e-3, 1.0).requires_grad_(True)
    ref = O.rd_loss(x, {"x_hat": xh, "likelihoods": {"y": ly, "z": lz}}, 0.0067)
    gr = torch.autograd.grad(ref["loss"], [xh, ly, lz])
    xg, xhg, lyg, lzg = x.to(d), xh.detach().to(d).requires_grad_(True), ly.detach().to(d).requires_grad_(True), \
        lz.detach().to(d).requires_grad_(True)
    out = RateDistortionLoss(0.0067)({"x_hat": xhg, "likelihoods": {"y": lyg, "z": lzg}}, xg)
    for k in ("loss", "bpp_loss", "mse_loss"):
        assert abs(out[k].item() - ref[k].item()) <= 2e-6 * abs(ref[k].item()), k
    gg = torch.autograd.grad(out["loss"], [xhg, lyg, lzg])
    for a, b, n in zip(gg, gr, ("dx_hat", "dlik_y", "dlik_z")):
        close(a, b, 1e-5, what=n)
    # Adam + clip against the oracle
    n = 100003
    p = U("ad.p", (n,), -1, 1)
    g = U("ad.g", (n,), -2, 2)
    m = torch.zeros(n)
    v = torch.zeros(n)
    pg, gg_, mg, vg = (t.clone().to(d) for t in (p, g, m, v))
    sq = torch.full((1,), 123.0, device=d)   # overwritten, not accumulated
    rws = torch.empty(L.REDUCE_WS_FLOATS, device=d)
    for step in (1, 2, 3):
        gs = [g.clone()]
        total = O.clip_grad_norm_(gs, 1.0)
        O.adam_step(p, gs[0], m, v, step, 1e-4)
        L.check(L.lib().icm_grad_sqnorm(L.ptr(gg_), n, L.ptr(sq), L.ptr(rws), L.stream()))
        assert abs(math.sqrt(sq.item()) - total.item()) <= 1e-5 * total.item()
        L.check(L.lib().icm_adam_step(L.ptr(pg), L.ptr(gg_), L.ptr(mg), L.ptr(vg), n, 1e-4, 0.9, 0.999, 1e-8, step,
                                      L.ptr(sq), 1.0, 1.0, L.stream()))
        close(pg, p, 1e-6, what=f"adam p step {step}")
        close(mg, m, 1e-5, what="adam m")
        close(vg, v, 1e-5, what="adam v")


# ------------------------------------------------------------------------------------------ thin-channel ends
def _tape_run(fn, inputs, g):
    """fn(tape, *device_tensors) -> y; backward seeded with g; returns (y, [grad of each input])"""
    from icm_amd import engine as E
    tape = E.Tape(need_grad=True)
    ts = [t.to(dev()) for t in inputs]
    y = fn(tape, *ts)
    tape.bind_grad(y, g.to(dev()).contiguous(), True)
    tape.backward()
    torch.cuda.synchronize()
    return y, [tape.grad_of(t) for t in ts]


@pytest.mark.parametrize("shape", [(2, 3, 20, 24, 40, 5, 2), (1, 3, 16, 16, 192, 5, 2), (2, 2, 9, 11, 33, 3, 1)])
def test_conv_thin_in_vs_torch(shape):
    """g_a.0-style Conv2d (few input channels) as im2col + 1x1 GEMM: forward, dgrad (col2im), wgrad, bias grad"""
    from icm_amd import engine as E
    N, Cin, H, Wd, Cout, k, s = shape
    x = U("thin.x", (N, Cin, H, Wd), -1.0, 1.0)
    w = U("thin.w", (Cout, Cin, k, k), -0.3, 0.3)
    b = U("thin.b", (Cout,), -0.5, 0.5)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = F.conv2d(xr, wr, br, stride=s, padding=k // 2)
    g = U("thin.g", yr.shape, -1.0, 1.0)
    gxr, gwr, gbr = torch.autograd.grad(yr, [xr, wr, br], g)
    y, (gx, gw, gb) = _tape_run(lambda t, xx, ww, bb: E.conv2d_thin_in(t, xx, ww, bb, stride=s, pad=k // 2), [x, w, b], g)
    close(y, yr, what="y")
    close(gx, gxr, what="gx")
    close(gw, gwr, what="gw", tol=1e-4)
    close(gb, gbr, what="gb", tol=1e-4)


@pytest.mark.parametrize("shape", [(2, 24, 10, 12, 3, 5, 2), (1, 192, 8, 8, 3, 5, 2), (2, 16, 7, 9, 2, 3, 1)])
def test_convT_thin_out_vs_torch(shape):
    """g_s.8-style ConvTranspose2d (few output channels) as 1x1 GEMM + col2im: forward, dgrad, wgrad, bias grad"""
    from icm_amd import engine as E
    from icm_amd.engine import VT
    N, Cin, H, Wd, Cout, k, s = shape
    x = U("thinT.x", (N, Cin, H, Wd), -1.0, 1.0)
    w = U("thinT.w", (Cin, Cout, k, k), -0.3, 0.3)
    b = U("thinT.b", (Cout,), -0.5, 0.5)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = F.conv_transpose2d(xr, wr, br, stride=s, padding=k // 2, output_padding=s - 1)
    g = U("thinT.g", yr.shape, -1.0, 1.0)
    gxr, gwr, gbr = torch.autograd.grad(yr, [xr, wr, br], g)
    y, (gx, gw, gb) = _tape_run(lambda t, xx, ww, bb: E.convT2d_thin_out(t, VT(xx), ww, bb, stride=s, pad=k // 2,
                                                                          output_padding=s - 1), [x, w, b], g)
    close(y, yr, what="y")
    close(gx, gxr, what="gx")
    close(gw, gwr, what="gw", tol=1e-4)
    close(gb, gbr, what="gb", tol=1e-4)


@pytest.mark.parametrize("shape", [(2, 48, 20, 24, 3, 3, 1), (1, 16, 9, 11, 2, 3, 0), (2, 24, 12, 12, 5, 5, 2), (1, 8, 6, 7, 1, 1, 0)])
def test_conv_thin_out_vs_torch(shape):
    """stf end_conv[2]-style stride-1 Conv2d (few OUTPUT channels; stf.py:401-404) as the ConvTranspose2d of the same
    map on the flipped / transposed weight (icm_permute_flip) through the thin-output path: forward, dgrad, wgrad carried
    back into the Conv2d weight layout, bias grad; a second call accumulates into an existing weight gradient"""
    from icm_amd import engine as E
    from icm_amd.engine import VT
    N, Cin, H, Wd, Cout, k, p = shape
    x = U("thinO.x", (N, Cin, H, Wd), -1.0, 1.0)
    w = U("thinO.w", (Cout, Cin, k, k), -0.3, 0.3)
    b = U("thinO.b", (Cout,), -0.5, 0.5)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = F.conv2d(xr, wr, br, stride=1, padding=p)
    g = U("thinO.g", yr.shape, -1.0, 1.0)
    gxr, gwr, gbr = torch.autograd.grad(yr, [xr, wr, br], g)
    y, (gx, gw, gb) = _tape_run(lambda t, xx, ww, bb: E.conv2d_thin_out(t, VT(xx), ww, bb, pad=p), [x, w, b], g)
    assert tuple(y.shape) == tuple(yr.shape)
    close(y, yr, what="y")
    close(gx, gxr, what="gx")
    close(gw, gwr, what="gw", tol=1e-4)
    close(gb, gbr, what="gb", tol=1e-4)
    # the same layer applied twice on one tape (weight sharing): the second backward ACCUMULATES into dW
    def twice(t, xx, ww, bb):
        y1 = E.conv2d_thin_out(t, VT(xx), ww, bb, pad=p)
        y2 = E.conv2d_thin_out(t, VT(xx), ww, bb, pad=p)
        out = E.new(y1)
        out.copy_(y1 + y2)          # (test plumbing only)
        def bwd():
            go = t.grad_of(out)
            for yy in (y1, y2):
                gy, acc = t.grad_for_write(yy)
                assert acc == 0
                gy.copy_(go)
        t.bw.append(bwd)
        return out
    _, (gx2, gw2, gb2) = _tape_run(twice, [x, w, b], g)
    close(gw2, 2 * gwr, what="gw x2", tol=1e-4)
    close(gx2, 2 * gxr, what="gx x2")
    close(gb2, 2 * gbr, what="gb x2", tol=1e-4)


def test_conv_group_shared_input_and_lrp_tail_vs_singles():
    """conv2d_group with members sharing one input (the fixed support of the late slices) and with the LRP tail fused
    must equal the same convolutions issued one by one (forward values, input / weight / bias / aux gradients)."""
    from icm_amd import engine as E
    from icm_amd.engine import VT
    d = dev()
    N, Cin, H, Wd, Cout, n = 2, 40, 8, 8, 32, 5
    x = U("grp.x", (N, Cin, H, Wd), -1.0, 1.0).to(d)
    ws = [U(f"grp.w{i}", (Cout, Cin, 3, 3), -0.2, 0.2).to(d) for i in range(n)]
    bs_ = [U(f"grp.b{i}", (Cout,), -0.3, 0.3).to(d) for i in range(n)]
    auxs = [U(f"grp.a{i}", (N, Cout, H, Wd), -1.0, 1.0).to(d) for i in range(n)]
    gs = [U(f"grp.g{i}", (N, Cout, H, Wd), -1.0, 1.0).to(d) for i in range(n)]

    def run(grouped):
        tape = E.Tape(need_grad=True)
        xs = x.clone()
        wl, bl, al = [w.clone() for w in ws], [b.clone() for b in bs_], [a.clone() for a in auxs]
        if grouped:
            ys = E.conv2d_group(tape, [VT(xs)] * n, wl, bl, pad=1, outs=[E.new((N, Cout, H, Wd), d) for _ in range(n)],
                                lrp_auxs=al)
        else:
            ys = [E.conv2d(tape, VT(xs), wl[i], bl[i], pad=1, lrp_aux=al[i]) for i in range(n)]
        for y, g in zip(ys, gs):
            tape.bind_grad(y, g.clone(), True)
        tape.backward()
        torch.cuda.synchronize()
        return (ys, tape.grad_of(xs), [tape.grad_of(w) for w in wl], [tape.grad_of(b) for b in bl],
                [tape.grad_of(a) for a in al])

    ya, gxa, gwa, gba, gaa = run(True)
    yb, gxb, gwb, gbb, gab = run(False)
    for i in range(n):
        close(ya[i], yb[i], what=f"y{i}")
        close(gwa[i], gwb[i], what=f"gw{i}", tol=1e-4)
        close(gba[i], gbb[i], what=f"gb{i}", tol=1e-4)
        close(gaa[i], gab[i], what=f"gaux{i}")
    close(gxa, gxb, what="gx (sum over the members sharing the input)", tol=1e-4)
    # and against torch for member 0: y = aux + 0.5 * tanh(conv(x))
    ref = auxs[0].cpu() + 0.5 * torch.tanh(F.conv2d(x.cpu(), ws[0].cpu(), bs_[0].cpu(), padding=1))
    close(ya[0], ref, what="lrp tail vs torch")


# ------------------------------------------------------------------------------------------ wgrad kernel variants
def _wgrad_ref(x, w, b, g, k, s, tr):
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    if tr:
        yr = F.conv_transpose2d(xr, wr, br, stride=s, padding=k // 2, output_padding=s - 1)
    else:
        yr = F.conv2d(xr, wr, br, stride=s, padding=k // 2)
    return torch.autograd.grad(yr, [wr, br], g)


WG_CASES = [
    # name, N, Cin, H, W, Cout, k, stride, transposed, variants to force (conv_wgrad.hip: 0-2 general kernel <2,1,7> /
    # <2,2,9> / <4,4,4>; 3-6 the 3x3-tiles-per-wave kernel <3,3,1,1> / <6,6,2,2> / <3,6,1,2> / <6,3,2,1>; 7 its
    # tap-per-wave form; 8 / 9 the nine-tap single-staging DMA kernel tap9<3,3> / tap9<2,2>; 10 its 25-tap 5x5 form;
    # 11-14 the DMA-only 1x1 kernel dma1<6,6> / <3,6> / <6,3> / <3,3>)
    ("wg_c5s2", 3, 40, 36, 20, 72, 5, 2, False, (0, 1, 3, 7, 10)),   # <4,4,4> and the 192-wide tiles need > 160 KB of LDS here
    ("wg_t5s2", 2, 48, 10, 12, 40, 5, 2, True, (0, 1, 10)),
    ("wg_c5s2_192_100", 2, 192, 24, 40, 100, 5, 2, False, (10,)),    # 25 taps from one staging: several blocks, channel tails
    ("wg_c5s1", 2, 48, 12, 20, 72, 5, 1, False, (0, 10)),
    ("wg_c3s1", 3, 48, 20, 12, 80, 3, 1, False, (0, 1, 2, 3, 5, 6, 7, 8, 9)),
    ("wg_c3s1_96_96", 2, 96, 16, 32, 96, 3, 1, False, (-1, 1, 7, 8, 9)),    # automatic choice: nine taps per workgroup (8)
    ("wg_c3s1_192_100", 2, 192, 16, 16, 100, 3, 1, False, (7, 8, 9)),
    ("wg_c3s1_224_176_ragged", 3, 224, 12, 10, 176, 3, 1, False, (8, 9)),   # several (a, b) blocks, ragged tiles, channel tails
    ("wg_c1_96_192", 2, 96, 16, 16, 192, 1, 1, False, (0, 1, 2, 3, 4, 5, 6, 11, 12, 13, 14)),
    ("wg_c1_130_70", 3, 130, 12, 20, 70, 1, 1, False, (0, 1, 2, 3, 4, 5, 6, 11, 12, 13, 14)),
    ("wg_c1_200_400", 2, 200, 8, 24, 400, 1, 1, False, (3, 4, 5, 6, 11, 12, 13, 14)),   # several (a, b) blocks per variant
    ("wg_c1_192_192_big", 4, 192, 64, 32, 192, 1, 1, False, (-1, 4, 11)),  # the automatic choice (DMA <6,6>, 32-pixel tiles)
    ("wg_c3_tiny", 4, 24, 4, 4, 48, 3, 1, False, (0, 1, 2, 3, 7, 8, 9)),
]


@pytest.mark.parametrize("case", WG_CASES, ids=[c[0] for c in WG_CASES])
def test_wgrad_every_variant(case):
    """each of wgrad_kernel<2,1,7>, <2,2,9>, <4,4,4> and <3,3,9,wave-split>, with the plain and the XCD-aware
    workgroup order, against torch.autograd.grad of F.conv2d / F.conv_transpose2d on CPU (weight AND fused bias grads)"""
    from icm_amd import _lib, layers
    name, N, Cin, H, Wd, Cout, k, s, tr, variants = case
    d = dev()
    lib = _lib.lib()
    m = layers.deconv(Cin, Cout, kernel_size=k, stride=s) if tr else layers.Conv2d(Cin, Cout, k, stride=s, padding=k // 2)
    w = U(name + ".w", m.weight.shape, -0.2, 0.2)
    b = U(name + ".b", m.bias.shape, -0.5, 0.5)
    x = U(name + ".x", (N, Cin, H, Wd), -1.0, 1.0)
    with torch.no_grad():
        yshape = (F.conv_transpose2d(x, w, b, stride=s, padding=k // 2, output_padding=s - 1) if tr
                  else F.conv2d(x, w, b, stride=s, padding=k // 2)).shape
    g = U(name + ".g", yshape, -1.0, 1.0)
    gwr, gbr = _wgrad_ref(x, w, b, g, k, s, tr)
    m = m.to(d)
    with torch.no_grad():
        m.weight.copy_(w)
        m.bias.copy_(b)
    try:
        for v in variants:
            for xcd in (0, 1):
                lib.icm_debug_force_wgrad_cfg(v, xcd)
                y = m(x.to(d))
                gw, gb = torch.autograd.grad(y, [m.weight, m.bias], g.to(d))
                close(gw, gwr, what=f"dw variant {v} xcd {xcd}")
                close(gb, gbr, what=f"db variant {v} xcd {xcd}")
    finally:
        lib.icm_debug_force_wgrad_cfg(-1, -1)


# ------------------------------------------------------------------------------------------ bench-size launches
BIG_CASES = [
    # the shapes bench.py's step is made of, at batch 16: name, N, Cin, H, W, Cout, k, stride, transposed
    ("g_a.2", 16, 192, 128, 128, 192, 5, 2, False),
    ("g_s.6", 16, 192, 64, 64, 192, 5, 2, True),
    ("g_a.7", 16, 192, 32, 32, 320, 5, 2, False),
    ("RU c1", 16, 192, 64, 64, 96, 1, 1, False),
    ("RU c3", 16, 96, 64, 64, 96, 3, 1, False),
    ("cc.0", 16, 480, 16, 16, 224, 3, 1, False),
    ("cc.6", 16, 128, 16, 16, 64, 3, 1, False),
    ("cc.8", 16, 64, 16, 16, 32, 3, 1, False),
]


@pytest.mark.parametrize("case", BIG_CASES, ids=[c[0] for c in BIG_CASES])
def test_conv_bench_size_vs_torch(case):
    """forward, input gradient, weight and bias gradient of the bench's own launch shapes (B=16: the single-pass
    tilings, XCD-aware orders and pixel-split counts chosen there) against F.conv2d / F.conv_transpose2d on CPU"""
    test_conv_fwd_bwd(case)


def test_conv_group_bench_size_vs_torch():
    """a 10-member grouped launch of the slice-chain head (5 mean + 5 scale chains of the independent tail slices share
    two inputs) at batch 16: every member's forward, shared-input gradient, weight and bias gradient vs torch"""
    from icm_amd import engine as E
    from icm_amd.engine import VT
    d = dev()
    N, Cin, Cout, H = 16, 480, 224, 16
    xs = [U(f"grp.x{i}", (N, Cin, H, H), -1.0, 1.0) for i in range(2)]
    ws = [U(f"grp.w{i}", (Cout, Cin, 3, 3), -0.05, 0.05) for i in range(10)]
    bs_ = [U(f"grp.b{i}", (Cout,), -0.5, 0.5) for i in range(10)]
    gs = [U(f"grp.g{i}", (N, Cout, H, H), -1.0, 1.0) for i in range(10)]
    xd = [t.to(d) for t in xs]
    wd = [t.to(d) for t in ws]
    bd = [t.to(d) for t in bs_]
    tape = E.Tape(need_grad=True)
    ys = E.conv2d_group(tape, [VT(xd[i // 5]) for i in range(10)], wd, bd, pad=1)
    for y, g in zip(ys, gs):
        tape.bind_grad(y, g.to(d), True)
    tape.backward()
    torch.cuda.synchronize()
    gx_ref = [torch.zeros_like(xs[0]), torch.zeros_like(xs[1])]
    for i in range(10):
        xr, wr, br = xs[i // 5].clone().requires_grad_(True), ws[i].clone().requires_grad_(True), bs_[i].clone().requires_grad_(True)
        yr = F.conv2d(xr, wr, br, padding=1)
        gx, gw, gb = torch.autograd.grad(yr, [xr, wr, br], gs[i])
        gx_ref[i // 5] += gx
        close(ys[i], yr, what=f"y{i}")
        close(tape.grad_of(wd[i]), gw, what=f"dw{i}")
        close(tape.grad_of(bd[i]), gb, what=f"db{i}")
    for j in range(2):
        close(tape.grad_of(xd[j]), gx_ref[j], what=f"dx{j}")


def test_gate_golden(golden_dir):
    """Win_noShift_Attention against the fixture produced by the REAL reference module (gate_d64_ws8.npz)"""
    from icm_amd import layers
    tag, dim, ws, shift = "gate_d64_ws8", 64, 8, 4
    f = load(golden_dir, tag)
    d = dev()
    m = layers.Win_noShift_Attention(dim=dim, num_heads=8, window_size=ws, shift_size=shift)
    sd = dict(m.state_dict())
    for k, v in m.state_dict().items():
        if not v.dtype.is_floating_point:
            continue
        key = tag + "." + k
        leaf = k.rsplit(".", 1)[-1]
        if leaf == "relative_position_bias_table":
            sd[k] = U(key, v.shape, -0.5, 0.5)
        elif leaf == "weight":
            bnd = 1.0 / math.sqrt(int(np.prod(v.shape[1:])))
            sd[k] = U(key, v.shape, -bnd, bnd) * 1.7
        else:
            sd[k] = U(key, v.shape, -0.1, 0.1)
    m.load_state_dict(sd)
    m = m.to(d)
    x = f["x"].to(d).requires_grad_(True)
    y = m(x)
    close(y, f["y"], what="y")
    names = [str(n) for n in f["grad_names"]]
    P = dict(m.named_parameters())
    gs = torch.autograd.grad(y, [x] + [P[n] for n in names], f["g"].to(d))
    close(gs[0], f["gx"], 5e-5, what="gx")
    gn = torch.stack([t.norm() for t in gs[1:]]).cpu()
    close(gn, f["grad_norms"], 5e-5, what="grad norms")
    close(gs[1 + names.index("conv_a.0.conv.0.weight")], f["g_first_conv_w"], 5e-5, what="g_first_conv_w")
    close(gs[1 + names.index("conv_b.4.bias")], f["g_last_conv_b"], 5e-5, what="g_last_conv_b")


# ------------------------------------------------------------------------------------------ window attention: MFMA vs VALU
@pytest.mark.parametrize("dim,heads,ws,shift,hw", [(192, 8, 8, 4, (16, 16)), (192, 8, 8, 0, (24, 24)), (64, 8, 8, 4, (16, 16)),
                                                   (128, 8, 8, 3, (8, 8)), (256, 8, 8, 4, (16, 16)), (384, 8, 8, 4, (8, 8)),
                                                   (320, 8, 4, 2, (16, 16)), (48, 3, 4, 2, (32, 24)), (96, 6, 4, 0, (8, 40)),
                                                   (192, 12, 4, 1, (12, 20)), (64, 8, 4, 2, (8, 8))])
def test_window_attention_mfma_vs_valu_and_oracle(dim, heads, ws, shift, hw):
    """the matrix-core kernels (winattn_mfma.hip: QK^T / PV and all five backward contractions on
    v_mfma_f32_32x32x2_f32 for 8x8 windows, v_mfma_f32_16x16x4_f32 for 4x4 windows incl. rows whose window count is not a
    multiple of the four a wave takes) against the generic VALU kernels on the same inputs, and both against the CPU
    oracle (win_attention.py:84-115,153-207): output, input gradient, all parameter gradients incl. the bias table"""
    from icm_amd import _lib, layers
    d = dev()
    lib = _lib.lib()
    tag = f"wm{dim}_{ws}_{shift}"
    m = layers.WinBasedAttention(dim=dim, num_heads=heads, window_size=ws, shift_size=shift)
    sd = {}
    for k, v in m.state_dict().items():
        leaf = k.rsplit(".", 1)[-1]
        if not v.dtype.is_floating_point:
            sd[k] = v
        elif leaf == "relative_position_bias_table":
            sd[k] = U(tag + k, v.shape, -0.5, 0.5)
        elif leaf == "weight":
            bnd = 1.0 / math.sqrt(int(np.prod(v.shape[1:])))
            sd[k] = U(tag + k, v.shape, -bnd, bnd) * 1.7
        else:
            sd[k] = U(tag + k, v.shape, -0.1, 0.1)
    m.load_state_dict(sd)
    x = U(tag + ".x", (2, dim, hw[0], hw[1]), -1.5, 1.5)
    g = U(tag + ".g", (2, dim, hw[0], hw[1]), -1, 1)
    osd = {"p." + k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    yr = O.win_based_attention(xr, osd, "p", heads, ws, shift)
    names = [n for n, _ in m.named_parameters()]
    grs = torch.autograd.grad(yr, [xr] + [osd["p." + n] for n in names], g)
    m = m.to(d)
    res = {}
    modes = (0, 1) if dim // heads <= 40 else (0,)   # the generic kernel's LDS layout stops at head dim 40 for 8x8 windows
    try:
        for mode in modes:
            lib.icm_debug_force_winattn_valu(mode)
            xg = x.to(d).requires_grad_(True)
            y = m(xg)
            gs = torch.autograd.grad(y, [xg] + [p for _, p in m.named_parameters()], g.to(d))
            res[mode] = [y] + list(gs)
    finally:
        lib.icm_debug_force_winattn_valu(0)
    for mode in modes:
        close(res[mode][0], yr, what=f"y mode {mode}")
        for a, b, n in zip(res[mode][1:], grs, ["x"] + names):
            close(a, b, 1e-4, what=f"grad {n} mode {mode}")
    if len(modes) == 2:
        for a, b, n in zip(res[0], res[1], ["y", "x"] + names):
            close(a, b, 2e-5, what=f"mfma vs valu {n}")


# ------------------------------------------------------------------------------------------ pointwise (1x1) direct kernel
P1_CASES = [
    # name, N, Cin, H, W, Cout  (Cin % 8 == 0; ragged pixel counts, co tails, every co-tile configuration)
    ("p1_192_192", 2, 192, 16, 16, 192),     # <6,1>
    ("p1_96_160", 3, 96, 9, 7, 160),         # <5,1>, N*H*W = 189: ragged last strip
    ("p1_48_128", 2, 48, 8, 8, 128),         # <4,1>
    ("p1_192_96", 2, 192, 12, 20, 96),       # <3,2>
    ("p1_8_48", 5, 8, 5, 5, 48),             # <2,2>, one chunk, co tail (48 = 32 + 16)
    ("p1_24_20", 1, 24, 6, 11, 20),          # <1,2>, three chunks (not a multiple of the prefetch depth)
    ("p1_320_576", 1, 320, 8, 8, 576),       # <6,1> x 3 co blocks, 40 chunks
]


@pytest.mark.parametrize("case", P1_CASES, ids=[c[0] for c in P1_CASES])
def test_conv1x1_direct_kernel(case):
    """the barrier-free pointwise kernel (forced on: these sizes are below its automatic threshold) against torch
    and against the LDS-staged kernel: forward, input gradient (runs as a transposed 1x1), weight / bias gradient"""
    from icm_amd import _lib, layers
    name, N, Cin, H, Wd, Cout = case
    d = dev()
    lib = _lib.lib()
    w = U(name + ".w", (Cout, Cin, 1, 1), -0.2, 0.2)
    b = U(name + ".b", (Cout,), -0.5, 0.5)
    x = U(name + ".x", (N, Cin, H, Wd), -1.0, 1.0)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = F.conv2d(xr, wr, br)
    g = U(name + ".g", yr.shape, -1.0, 1.0)
    gxr, gwr, gbr = torch.autograd.grad(yr, [xr, wr, br], g)
    out = {}
    try:
        for mode in (1, 0):
            lib.icm_debug_force_conv1x1(mode)
            m = layers.Conv2d(Cin, Cout, kernel_size=1).to(d)
            with torch.no_grad():
                m.weight.copy_(w)
                m.bias.copy_(b)
            xg = x.to(d).requires_grad_(True)
            y = m(xg)
            gx, gw, gb = torch.autograd.grad(y, [xg, m.weight, m.bias], g.to(d))
            out[mode] = (y, gx)
            close(y, yr, what=f"y mode{mode}")
            close(gx, gxr, what=f"dx mode{mode}")
            close(gw, gwr, what=f"dw mode{mode}")
            close(gb, gbr, what=f"db mode{mode}")
    finally:
        lib.icm_debug_force_conv1x1(-1)
    close(out[1][0], out[0][0].cpu(), what="direct vs staged y", tol=2e-6)
    close(out[1][1], out[0][1].cpu(), what="direct vs staged dx", tol=2e-6)


@pytest.mark.parametrize("which", ["gate192", "gate320", "gdn", "igdn", "gate_golden"])
def test_conv1x1_direct_kernel_fused_neighbours(which, golden_dir):
    """every fused prologue / epilogue the 1x1 convolutions of the model use (virtual GELU operand, x^2 operand,
    residual (+GELU), GDN / IGDN normalisation, GELU' and AXPY2 backward epilogues, gradient accumulation) through the
    direct kernel: the oracle / golden comparisons of the gate and GDN tests, re-run with the kernel forced on"""
    from icm_amd import _lib
    lib = _lib.lib()
    try:
        lib.icm_debug_force_conv1x1(1)
        if which == "gate192":
            test_attention_gate_vs_oracle(192, 8, 4, 16)
        elif which == "gate320":
            test_attention_gate_vs_oracle(320, 4, 2, 8)
        elif which == "gdn":
            test_gdn_192_vs_oracle(False)
        elif which == "igdn":
            test_gdn_192_vs_oracle(True)
        else:
            test_gate_golden(golden_dir)
    finally:
        lib.icm_debug_force_conv1x1(-1)


# ------------------------------------------------------------------------------------------ 8-wave K-split conv kernel
KS8_CASES = [
    # name, N, Cin, H, W, Cout, k, stride, transposed: stride-1 sampling problems of the slice-chain / hyper-path kind
    ("ks8_c3_224_176", 2, 224, 16, 16, 176, 3, 1, False),
    ("ks8_c3_64_32", 3, 64, 12, 20, 32, 3, 1, False),      # one co tile: the 32 x 128 block; ragged pixel tiles
    ("ks8_c3_40_200_odd", 2, 40, 9, 7, 200, 3, 1, False),  # channel tails, fewer sub-steps than waves in the last chunk
    ("ks8_c5s1_48_72", 2, 48, 12, 10, 72, 5, 1, False),    # 25 taps: two 8-channel groups per staged chunk
    ("ks8_c3_8_64_tiny", 4, 8, 4, 4, 64, 3, 1, False),     # 9 sub-steps in total: one wave gets two, the others one
]


@pytest.mark.parametrize("case", KS8_CASES, ids=[c[0] for c in KS8_CASES])
@pytest.mark.parametrize("cfg", [100, 101])
def test_conv_ks8_kernel(case, cfg):
    """the 8-wave K-split kernel forced on (100: 64 co x 64 px blocks, 101: 32 co x 128 px): forward and input gradient
    (the dgrad of a stride-1 conv is the same kernel on the transposed weights) against torch, and the automatic path"""
    from icm_amd import _lib
    lib = _lib.lib()
    try:
        lib.icm_debug_force_conv_cfg(cfg)
        test_conv_fwd_bwd(case)
    finally:
        lib.icm_debug_force_conv_cfg(-1)


def test_conv_ks8_fused_neighbours():
    """residual / LRP / GELU' epilogues and grouped launches through the K-split kernel: the grouped slice-chain test
    and the dim-320 gate against the oracle with the kernel forced on"""
    from icm_amd import _lib
    lib = _lib.lib()
    try:
        lib.icm_debug_force_conv_cfg(100)
        test_conv_group_shared_input_and_lrp_tail_vs_singles()
        test_attention_gate_vs_oracle(320, 4, 2, 8)
        lib.icm_debug_force_conv_cfg(101)
        test_conv_group_shared_input_and_lrp_tail_vs_singles()
        # fused PixelShuffle store (subpel_conv3x3 of the hyper synthesis) and the materialised-GELU second output
        from icm_amd import engine as E
        from icm_amd.engine import VT
        x = U("ks8.ps.x", (2, 40, 8, 12), -1.0, 1.0)
        w = U("ks8.ps.w", (96, 40, 3, 3), -0.2, 0.2)
        b = U("ks8.ps.b", (96,), -0.3, 0.3)
        ref = F.pixel_shuffle(F.conv2d(x, w, b, padding=1), 2)
        min_px, E._MAT_MIN_PIXELS = E._MAT_MIN_PIXELS, 0      # materialise even this small launch
        inplace = E.EVAL_INPLACE_ACT
        try:
            for cfg in (100, 101):
                lib.icm_debug_force_conv_cfg(cfg)
                E.EVAL_INPLACE_ACT = False      # two outputs: the pre-activation and its GELU
                tape = E.Tape(need_grad=False)
                y = E.conv2d(tape, VT(x.to(dev())), w.to(dev()), b.to(dev()), pad=1, pixel_shuffle=2, act_out=True)
                close(y, ref, what=f"pixel-shuffle store cfg {cfg}")
                close(tape.mat[E._key(y)], F.gelu(ref), what=f"materialised gelu cfg {cfg}")
                E.EVAL_INPLACE_ACT = True       # inference default: only gelu(y) is stored, in y itself
                tape = E.Tape(need_grad=False)
                y = E.conv2d(tape, VT(x.to(dev())), w.to(dev()), b.to(dev()), pad=1, pixel_shuffle=2, act_out=True)
                assert tape.mat[E._key(y)] is y
                close(y, F.gelu(ref), what=f"activated-only store cfg {cfg}")
                # ... and never when gradients are recorded (the backward pass needs the pre-activation)
                tape = E.Tape(need_grad=True)
                y = E.conv2d(tape, VT(x.to(dev())), w.to(dev()), b.to(dev()), pad=1, pixel_shuffle=2, act_out=True)
                assert tape.mat[E._key(y)] is not y
                close(y, ref, what=f"pre-activation kept for the backward pass cfg {cfg}")
        finally:
            E.EVAL_INPLACE_ACT = inplace
        E._MAT_MIN_PIXELS = min_px
    finally:
        lib.icm_debug_force_conv_cfg(-1)


def test_entropy_model_outputs_are_differentiable():
    """GaussianConditional.forward / EntropyBottleneck.forward return (outputs, likelihood) with outputs =
    quantize(inputs, "noise" | "dequantize", means) INCLUDING its gradient (entropy_models.py:126-150,468-472,655):
    identity to the inputs in noise mode; zero to the inputs and one to the means in dequantize mode."""
    import torch
    from icm_amd.entropy_models import EntropyBottleneck, GaussianConditional
    from oracle import weights as W
    dev = "cuda:0"
    y = W._u("qo.y", (2, 6, 4, 4), -4.0, 4.0).to(dev).requires_grad_(True)
    mu = W._u("qo.mu", (2, 6, 4, 4), -1.0, 1.0).to(dev).requires_grad_(True)
    sc = W._u("qo.sc", (2, 6, 4, 4), 0.2, 2.0).to(dev).requires_grad_(True)
    w = W._u("qo.w", (2, 6, 4, 4), -1.0, 1.0).to(dev)
    gc = GaussianConditional(None).to(dev)
    # eval: outputs = round(y - mu) + mu
    out, lik = gc(y, sc, mu, training=False)
    assert torch.equal(out.detach(), torch.round(y.detach() - mu.detach()) + mu.detach())
    (gy, gm) = torch.autograd.grad((out * w).sum(), [y, mu], allow_unused=True)
    assert gy is None or float(gy.abs().max()) == 0.0
    assert torch.equal(gm, w)
    # train: outputs = y + noise
    noise = W._u("qo.n", (2, 6, 4, 4), -0.5, 0.5)
    gc.inject_noise(noise)
    out, lik = gc(y, sc, mu, training=True)
    assert torch.allclose(out.detach(), y.detach() + noise.to(dev), atol=1e-7)
    gy, gm, gs = torch.autograd.grad((out * w).sum() + lik.sum(), [y, mu, sc])
    out2, lik2 = gc(y, sc, mu, training=True)
    gy2, gm2, gs2 = torch.autograd.grad(lik2.sum(), [y, mu, sc])
    assert torch.allclose(gy - gy2, w, atol=1e-6) and torch.equal(gm, gm2) and torch.equal(gs, gs2)
    # EntropyBottleneck: the same through the fused tape op
    eb = EntropyBottleneck(6).to(dev)
    z = W._u("qo.z", (2, 6, 3, 3), -3.0, 3.0).to(dev).requires_grad_(True)
    wz = W._u("qo.wz", (2, 6, 3, 3), -1.0, 1.0).to(dev)
    nz = W._u("qo.nz", (2, 6, 3, 3), -0.5, 0.5)
    eb.inject_noise(nz)
    zt, zl = eb(z, training=True)
    (g1,) = torch.autograd.grad((zt * wz).sum() + zl.sum(), [z])
    zt, zl = eb(z, training=True)
    (g2,) = torch.autograd.grad(zl.sum(), [z])
    assert torch.allclose(g1 - g2, wz, atol=1e-6)
    zt, zl = eb(z, training=False)
    med = eb.quantiles[:, 0, 1].reshape(1, -1, 1, 1).detach()
    assert torch.equal(zt.detach(), torch.round(z.detach() - med) + med)
    gz, gq = torch.autograd.grad((zt * wz).sum(), [z, eb.quantiles], allow_unused=True)
    assert gz is None or float(gz.abs().max()) == 0.0
    assert torch.allclose(gq[:, 0, 1], wz.sum(dim=(0, 2, 3)), atol=1e-5)


# ------------------------------------------------------------------------------------------ first-layer split pieces
@pytest.mark.parametrize("N,h,w", [(2, 8, 8), (16, 16, 16)])
def test_split_first_layer_pieces_vs_torch(N, h, w):
    """The building blocks of icm_amd/slices.py against torch on the CPU: (a) packed weights that are column blocks of
    wider canonical weights, concatenated along GEMM-M (forward of several chains' latent blocks in one launch);
    (b) a second input-channel block accumulated in place with the materialised GELU written after the add; (c) the
    input gradient of (a) as ONE contraction over the K-concatenated channels, also through the blocked channel map
    (runs of channels lying apart in a wider buffer); (d) weight gradients written into column blocks of the wider
    tensors (dw_ld)."""
    from icm_amd import engine as E
    from icm_amd import _lib as L
    d = dev()
    C1, C2, D0 = 40, 24, 64
    w1, w2 = U("sp.w1", (D0, C1 + C2, 3, 3), -0.2, 0.2), U("sp.w2", (D0, C1 + 8, 3, 3), -0.2, 0.2)
    b1 = U("sp.b1", (D0,))
    x1, x2 = U("sp.x1", (N, C1, h, w)), U("sp.x2", (N, C2, h, w))
    W1, W2, X1, X2, B1 = (t.to(d) for t in (w1, w2, x1, x2, b1))
    tape = E.Tape(need_grad=False)
    conv = dict(KH=3, KW=3, stride=1, pad=1, OH=h, OW=w)
    # (a) latent blocks of both weights side by side: PRE[:, :64] = conv(x1, w1[:, :C1]), PRE[:, 64:] = conv(x1, w2[:, :C1])
    PRE = torch.full((N, 3 * D0, h, w), 7.0, device=d)       # a third, untouched run in the middle (blocked map below)
    wpA = tape.pack_cat([(W1, 0), (W2, 0)], D0, C1, 3, 3, 1, 0, 1, 1, "M")
    tmp = torch.empty((N, 2 * D0, h, w), device=d)
    E.conv_launch(tape, X1, wpA, None, tmp, Cin=C1, Cout=2 * D0, transposed=0, **conv)
    close(tmp[:, :D0], F.conv2d(x1, w1[:, :C1], None, padding=1), what="A/w1")
    close(tmp[:, D0:], F.conv2d(x1, w2[:, :C1], None, padding=1), what="A/w2")
    PRE[:, :D0] = tmp[:, :D0]
    PRE[:, 2 * D0:] = tmp[:, D0:]
    # (b) the support block of w1 accumulated in place (+ bias here), GELU of the SUM materialised
    G = torch.zeros((N, D0, h, w), device=d)
    wpB = tape.pack_cat([(W1, C1)], D0, C2, 3, 3, 1, 0, 1, 1, "M")
    E.conv_launch_grouped(tape, [X2], [wpB], [B1], [PRE[:, :D0]], Cin=C2, Cout=D0, transposed=0, y2s=[G], accum=1, **conv)
    full = F.conv2d(torch.cat([x1, x2], 1), w1, b1, padding=1)
    close(PRE[:, :D0], full, what="A+B")
    close(G, F.gelu(full), what="gelu(A+B)")
    assert float((PRE[:, D0:2 * D0] - 7.0).abs().max()) == 0.0
    # (c) d x1 = sum over both chains: one contraction over K = 2*D0, contiguous and through the blocked map
    g = U("sp.g", (N, 2 * D0, h, w))
    ref_dx = (torch.nn.grad.conv2d_input(x1.shape, w1[:, :C1].contiguous(), g[:, :D0].contiguous(), padding=1) +
              torch.nn.grad.conv2d_input(x1.shape, w2[:, :C1].contiguous(), g[:, D0:].contiguous(), padding=1))
    wpK = tape.pack_cat([(W1, 0), (W2, 0)], C1, D0, 3, 3, 0, 1, 1, 1, "K")
    Gd = g.to(d)
    dx = torch.empty((N, C1, h, w), device=d)
    E.conv_launch(tape, Gd, wpK, None, dx, Cin=2 * D0, Cout=C1, transposed=1, **conv)
    close(dx, ref_dx, tol=5e-5, what="dgrad K-concat")
    Gw = torch.full((N, 3 * D0, h, w), float("nan"), device=d)     # runs 0 and 2 of three: the middle one must not be read
    Gw[:, :D0] = Gd[:, :D0]
    Gw[:, 2 * D0:] = Gd[:, D0:]
    dx2 = torch.zeros((N, C1, h, w), device=d)
    E.conv_launch(tape, Gw, wpK, None, dx2, Cin=2 * D0, Cout=C1, transposed=1, seg=(D0, D0), **conv)
    assert torch.equal(dx2, dx)
    # (d) weight gradients into column blocks of the canonical tensors
    gw1 = torch.zeros_like(W1)
    gb1 = torch.zeros_like(B1)
    t2 = E.Tape(need_grad=True)
    E.wgrad_defer(t2, Gd[:, :D0], X1, gw1[:, :C1], Ca=D0, Cb=C1, KH=3, KW=3, stride=1, pad=1, accum=1, dbias=gb1,
                  accum_bias=0, dw_ld=C1 + C2)
    E.wgrad_defer(t2, Gd[:, :D0], X2, gw1[:, C1:], Ca=D0, Cb=C2, KH=3, KW=3, stride=1, pad=1, accum=1, dw_ld=C1 + C2)
    E.flush_wgrads(t2)
    ref_w = torch.nn.grad.conv2d_weight(torch.cat([x1, x2], 1), w1.shape, g[:, :D0].contiguous(), padding=1)
    close(gw1, ref_w, tol=5e-5, what="wgrad blocks")
    close(gb1, g[:, :D0].sum(dim=(0, 2, 3)), tol=5e-5, what="bias grad")
    torch.cuda.synchronize()


def test_grouped_conv_rejects_mismatching_members():
    """C ABI: the members of a grouped launch must agree with member 0 in everything but their pointers"""
    import ctypes as C
    from icm_amd import _lib as L
    d = dev()
    x = torch.zeros((2, 8, 8, 8), device=d)
    y = torch.zeros((2, 32, 8, 8), device=d)
    wp = torch.zeros(L.lib().icm_packed_weight_floats(32, 8, 3, 3), device=d)

    def arg(**over):
        a = L.ConvArgs()
        a.x, a.x_bs, a.N, a.Cin, a.H, a.W = x.data_ptr(), 8 * 64, 2, 8, 8, 8
        a.wp, a.y, a.y_bs, a.Cout, a.OH, a.OW = wp.data_ptr(), y.data_ptr(), 32 * 64, 32, 8, 8
        a.KH, a.KW, a.stride, a.pad = 3, 3, 1, 1
        for k, v in over.items():
            setattr(a, k, v)
        return a
    ok = (L.ConvArgs * 2)(arg(), arg())
    assert L.lib().icm_conv_run_grouped(ok, 2, L.stream()) == 0
    for over in (dict(y_bs=16 * 64), dict(Cin=4), dict(Cout=16), dict(accum=1), dict(epi=L.EPI_MUL_DGELU),
                 dict(x_seg_len=4, x_seg_gap=4), dict(pro_act=L.ACT_GELU)):
        bad = (L.ConvArgs * 2)(arg(), arg(**over))
        assert L.lib().icm_conv_run_grouped(bad, 2, L.stream()) == 1, over
    torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------ Winograd F(2x2, 3x3)
WINO_CASES = [
    # name, N, Cin, H, W, Cout
    ("w_chain_224_176", 16, 224, 16, 16, 176),
    ("w_ru_96_96", 2, 96, 64, 64, 96),
    ("w_odd_40_72", 3, 40, 11, 13, 72),          # odd sizes: partial 2x2 tiles, channels not a multiple of 16 / 32
    ("w_tiny_192_192", 16, 192, 4, 4, 192),
    ("w_64_32", 16, 64, 16, 16, 32),
    ("w_24_200", 2, 24, 20, 36, 200),
]


@pytest.mark.parametrize("case", WINO_CASES, ids=[c[0] for c in WINO_CASES])
def test_winograd_conv_vs_torch_and_direct(case, monkeypatch):
    """csrc/conv_wino.hip through engine.conv2d: forward (bias, materialised GELU) and input gradient (GELU' epilogue,
    accumulation) of 3x3 stride-1 pad-1 convolutions against torch on the CPU, and against the direct implicit-GEMM
    kernel on the same inputs (the two algorithms must agree to summation-order noise)."""
    from icm_amd import engine as E
    from icm_amd.engine import VT
    _, N, Cin, H, Wd, Cout = case
    d = dev()
    x = U(case[0] + ".x", (N, Cin, H, Wd), -1.5, 1.5)
    w = U(case[0] + ".w", (Cout, Cin, 3, 3), -0.2, 0.2)
    b = U(case[0] + ".b", (Cout,), -0.5, 0.5)
    g = U(case[0] + ".g", (N, Cout, H, Wd), -1.0, 1.0)
    xr = x.clone().requires_grad_(True)
    yr = F.conv2d(F.gelu(xr), w, b, padding=1)
    yr.backward(g)
    outs = {}
    monkeypatch.setattr(E, "_MAT_MIN_PIXELS", 0)
    monkeypatch.setattr(E, "_WINO_MIN_WORK", 0.0)     # (by default small / shallow launches stay on the direct kernels)
    monkeypatch.setattr(E, "_WINO_MIN_CIN", 16)
    for name, on in (("wino", True), ("direct", False)):
        monkeypatch.setattr(E, "USE_WINO", on)
        assert E.wino_ok(3, 3, 1, 1, Cin) == on
        monkeypatch.setattr(E, "WINO_PRE", N % 2 == 0)    # both placements of the input transform across the cases
        tape = E.Tape(need_grad=True)
        xd = x.to(d)
        wd_, bd = w.to(d), b.to(d)
        y = E.conv2d(tape, VT(xd, E.ACT_GELU), wd_, bd, pad=1, act_out=True)       # virtual-GELU operand
        y2 = tape.mat[E._key(y)]
        tape.bind_grad(y, g.to(d), True)
        gx0 = torch.full_like(xd, 0.25)
        tape.bind_grad(xd, gx0, True)                                             # dgrad accumulates onto 0.25
        tape.backward()
        torch.cuda.synchronize()
        outs[name] = (y, y2, tape.grad_of(xd), tape.grad_of(wd_), tape.grad_of(bd))
        close(y, yr, what=name + " fwd")
        close(y2, F.gelu(yr), what=name + " gelu(fwd)")
        close(tape.grad_of(xd) - 0.25, xr.grad, tol=5e-5, what=name + " dgrad")
    # algorithm against algorithm: tighter than either against the CPU
    close(outs["wino"][0], outs["direct"][0], tol=1e-5, what="wino vs direct fwd")
    close(outs["wino"][2], outs["direct"][2], tol=2e-5, what="wino vs direct dgrad")
    assert torch.equal(outs["wino"][3], outs["direct"][3])    # the weight gradient is the same (direct) kernel


@pytest.mark.parametrize("case", [("ip.igemm", 2, 24, 12, 12, 40, 3, False), ("ip.1x1", 4, 96, 64, 64, 192, 1, False),
                                  ("ip.wino", 4, 64, 16, 16, 64, 3, True), ("ip.group", 2, 40, 16, 16, 32, 3, False)])
def test_inference_stores_only_the_activated_value(case, monkeypatch):
    """need_grad=False + act_out: y2 == y is passed to the kernels and y holds gelu(conv(x)) (+ residual), for the staged
    implicit-GEMM kernel, the independent-wave 1x1 kernel, the Winograd kernel and grouped launches; consumers that
    wrap the result as VT(y, GELU) read it unchanged"""
    from icm_amd import engine as E
    from icm_amd.engine import VT
    d = dev()
    name, N, Cin, H, Wd, Cout, k, wino = case
    monkeypatch.setattr(E, "_MAT_MIN_PIXELS", 0)
    monkeypatch.setattr(E, "EVAL_INPLACE_ACT", True)
    if wino:
        monkeypatch.setattr(E, "_WINO_MIN_WORK", 0.0)
        monkeypatch.setattr(E, "_WINO_MIN_CIN", 16)
    else:
        monkeypatch.setattr(E, "USE_WINO", False)
    x = U(name + ".x", (N, Cin, H, Wd), -1.0, 1.0)
    w = U(name + ".w", (Cout, Cin, k, k), -0.2, 0.2)
    b = U(name + ".b", (Cout,), -0.3, 0.3)
    r = U(name + ".r", (N, Cout, H, Wd), -1.0, 1.0)
    ref = F.gelu(F.conv2d(x, w, b, padding=k // 2) + r)
    tape = E.Tape(need_grad=False)
    if name == "ip.group":
        ys = E.conv2d_group(tape, [VT(x.to(d)), VT(x.to(d))], [w.to(d), w.to(d)], [b.to(d), b.to(d)], pad=k // 2,
                            ress=[VT(r.to(d)), VT(r.to(d))], act_out=True)
    else:
        ys = [E.conv2d(tape, VT(x.to(d)), w.to(d), b.to(d), pad=k // 2, res=VT(r.to(d)), act_out=True)]
    for y in ys:
        assert tape.mat[E._key(y)] is y
        close(y, ref, what=name + " activated-only store")
        t, a = E._operand(tape, VT(y, E.ACT_GELU))
        assert t is y and a == E.ACT_NONE


def test_winograd_grouped_lrp_residual_and_blocked_map(monkeypatch):
    """fused neighbours of the Winograd epilogue (residual, LRP tanh with its second output, in-place accumulation +
    materialisation) in grouped launches, and the blocked input-channel map, against the direct kernel"""
    from icm_amd import engine as E
    d = dev()
    N, Cin, H, Wd, Cout = 4, 64, 16, 16, 32
    xs = [U(f"wg.x{i}", (N, Cin, H, Wd)).to(d) for i in range(3)]
    ws = [U(f"wg.w{i}", (Cout, Cin, 3, 3), -0.2, 0.2).to(d) for i in range(3)]
    bs_ = [U(f"wg.b{i}", (Cout,)).to(d) for i in range(3)]
    aux = [U(f"wg.a{i}", (N, Cout, H, Wd)).to(d) for i in range(3)]
    res = {}
    for on in (True, False):
        monkeypatch.setattr(E, "USE_WINO", on)
        tape = E.Tape(need_grad=False)
        wino = 1 if on else 0
        wps = [tape.pack(w, Cout, Cin, 3, 3, 1, 0, 1, 1, wino=wino) for w in ws]
        kw = dict(Cin=Cin, Cout=Cout, KH=3, KW=3, stride=1, pad=1, transposed=0, OH=H, OW=Wd, algo=wino)
        y_lrp = [torch.empty((N, Cout, H, Wd), device=d) for _ in range(3)]
        t_lrp = [torch.empty((N, Cout, H, Wd), device=d) for _ in range(3)]
        E.conv_launch_grouped(tape, xs, wps, bs_, y_lrp, epi=E.EPI_LRP, auxs=aux, y2s=t_lrp, **kw)
        y_res = [torch.empty((N, Cout, H, Wd), device=d) for _ in range(3)]
        E.conv_launch_grouped(tape, xs, wps, bs_, y_res, epi=E.EPI_RES_GELU, ress=aux, **kw)
        y_acc = [a.clone() for a in aux]
        g_acc = [torch.empty((N, Cout, H, Wd), device=d) for _ in range(3)]
        E.conv_launch_grouped(tape, xs, wps, None, y_acc, y2s=g_acc, accum=1, **kw)
        # blocked map: the 64 input channels as two runs of 32 lying 96 planes apart
        wide = torch.full((N, 128 + 32, H, Wd), float("nan"), device=d)
        wide[:, :32] = xs[0][:, :32]
        wide[:, 128:] = xs[0][:, 32:]
        y_seg = torch.empty((N, Cout, H, Wd), device=d)
        E.conv_launch(tape, wide, wps[0], bs_[0], y_seg, seg=(32, 96), **kw)
        torch.cuda.synchronize()
        res[on] = (y_lrp, t_lrp, y_res, y_acc, g_acc, y_seg)
    for i in range(3):
        ref = F.conv2d(xs[i].cpu(), ws[i].cpu(), bs_[i].cpu(), padding=1)
        close(res[True][0][i], aux[i].cpu() + 0.5 * torch.tanh(ref), what="lrp")
        close(res[True][1][i], torch.tanh(ref), what="lrp tanh")
        close(res[True][2][i], ref + F.gelu(aux[i].cpu()), what="res_gelu")
        acc_ref = F.conv2d(xs[i].cpu(), ws[i].cpu(), None, padding=1) + aux[i].cpu()
        close(res[True][3][i], acc_ref, what="accumulate")
        close(res[True][4][i], F.gelu(acc_ref), what="gelu(accumulate)")
        for k in range(5):
            close(res[True][k][i], res[False][k][i], tol=1e-5, what=f"wino vs direct {k}")
    close(res[True][5], F.conv2d(xs[0].cpu(), ws[0].cpu(), bs_[0].cpu(), padding=1), what="blocked map")
    close(res[True][5], res[False][5], tol=1e-5, what="blocked map wino vs direct")


WINO_WG_CASES = [
    # name, N, Cb (in), H, W, Ca (out), problems in the batch
    ("ww_chain_224_176", 16, 224, 16, 16, 176, 3),
    ("ww_first_320_224", 16, 320, 16, 16, 224, 2),
    ("ww_ru_96_96", 2, 96, 64, 64, 96, 1),
    ("ww_odd_40_72", 3, 40, 11, 13, 72, 2),
    ("ww_small_24_200", 2, 24, 6, 10, 200, 1),
]


@pytest.mark.parametrize("case", WINO_WG_CASES, ids=[c[0] for c in WINO_WG_CASES])
def test_winograd_wgrad_vs_torch_and_direct(case):
    """csrc/wgrad_wino.hip (dU = (A dY A^T)(.)(B^T d B) over tiles, dW = G^T dU G in the reduction) through the
    deferred / batched weight-gradient path: weight and fused bias gradients against torch.nn.grad.conv2d_weight on the
    CPU and against the direct kernel; accumulation into an existing gradient; a column block of a wider weight (dw_ld)."""
    from icm_amd import engine as E
    _, N, Cb, H, Wd, Ca, nprob = case
    d = dev()
    res = {}
    xs = [U(f"{case[0]}.x{i}", (N, Cb, H, Wd), -1.0, 1.0) for i in range(nprob)]
    gs = [U(f"{case[0]}.g{i}", (N, Ca, H, Wd), -1.0, 1.0) for i in range(nprob)]
    for algo in (1, 0):
        tape = E.Tape(need_grad=True)
        dws = [torch.full((Ca, Cb + 8, 3, 3), 0.5, device=d) for _ in range(nprob)]     # wider weight: columns [8, 8 + Cb)
        dbs = [torch.zeros(Ca, device=d) for _ in range(nprob)]
        keep = []
        for i in range(nprob):
            g_, x_ = gs[i].to(d), xs[i].to(d)
            keep.append((g_, x_))
            E.wgrad_defer(tape, g_, x_, dws[i][:, 8:], Ca=Ca, Cb=Cb, KH=3, KW=3, stride=1, pad=1, accum=1, dbias=dbs[i],
                          accum_bias=0, dw_ld=Cb + 8, algo=algo)
        E.flush_wgrads(tape)
        torch.cuda.synchronize()
        res[algo] = (dws, dbs)
    for i in range(nprob):
        ref = torch.nn.grad.conv2d_weight(xs[i], (Ca, Cb, 3, 3), gs[i], padding=1)
        for algo in (1, 0):
            dw = res[algo][0][i].cpu()
            assert float((dw[:, :8] - 0.5).abs().max()) == 0.0, "columns outside the block were touched"
            close(dw[:, 8:] - 0.5, ref, tol=5e-5, what=f"algo {algo} dW[{i}]")
            close(res[algo][1][i], gs[i].sum(dim=(0, 2, 3)), tol=5e-5, what=f"algo {algo} dbias[{i}]")
        close(res[1][0][i], res[0][0][i], tol=2e-5, what="wino vs direct")
