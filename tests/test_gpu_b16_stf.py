"""GPU parity of the `stf` and `stf6` models at the BENCH geometry (batch 16 x 256x256), the way tests/test_gpu_b16.py
does it for `cnn`: tile / split-K / 1x1 / DMA / weight-gradient variant selection depends on pixel and workgroup
counts, so the kernel configurations `bench.py --model stf` (and the `stf` sub-object of the default line) time are
only exercised at this size.  Workload = bench.make_workload (default-initialised weights under manual_seed(0),
x = rand(16,3,256,256) seed 1234); the oracle (oracle/stf_oracle.py, stf6_oracle.py) runs the same batch on the host
cores with the HIP path's rounding decisions adopted (tests/_parity.py) and flips counted."""
import math

import pytest
import torch

import _parity as PT
from oracle import stf6_oracle as S6
from oracle import stf_oracle as S
from oracle import wacnn_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GRAD_L2_TOL, GRAD_ELEM_TOL = 5e-5, 2e-4   # same bounds as the cnn test at this geometry


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def _drops(rates, B, dev, seed):
    """DropPath scales (0 or 1 / keep_prob per sample and branch) from a fixed generator"""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, rate in rates.items():
        if rate > 0:
            out[name] = ((torch.rand((2, B), generator=g) < 1.0 - rate).float() / (1.0 - rate))
    return out


def test_b16_stf_eval_forward_vs_oracle():
    from icm_amd import engine as E
    from icm_amd.models import stf_forward
    import bench
    tr, x, sd = bench.make_workload("stf", torch.device(DEV))
    P = tr.params()
    ro, keep = PT.hip_round_decisions(stf_forward, P, x, None, None)
    with torch.no_grad():
        x_hat, y_lik, z_lik = stf_forward(E.Tape(need_grad=False), P, x)
        ref = S.stf_forward(sd, x.cpu(), None, None, keep=True, round_override=ro)
    d = ref["_dbg"]
    fy, fz = PT.count_flips(ro, d, sd)
    print("flips y/z:", fy, fz, "| near-half:", PT.near_half(d))
    assert fy <= PT.near_half(d) + 2 and fz == 0
    for k in ("y", "z", "mu", "scale"):
        r = rel(keep[k], d[k])
        print(f"  {k}: rel {r:.2e}")
        assert r < 1e-4, k
    assert rel(x_hat, ref["x_hat"]) < 1e-4
    assert rel(y_lik, ref["likelihoods"]["y"]) < 1e-4 and rel(z_lik, ref["likelihoods"]["z"]) < 1e-4
    Lh = O.rd_loss(x.cpu(), {"x_hat": x_hat.cpu(), "likelihoods": {"y": y_lik.cpu(), "z": z_lik.cpu()}})
    Lr = O.rd_loss(x.cpu(), ref)
    for k in ("bpp_loss", "mse_loss", "loss"):
        e = abs(Lh[k].item() - Lr[k].item()) / abs(Lr[k].item())
        print(f"  {k}: {Lh[k].item():.6f} vs {Lr[k].item():.6f} rel {e:.2e}")
        assert e < 5e-6, k


def test_b16_stf_trainer_step_vs_oracle():
    """one native training step at B=16 (injected noise and DropPath scales) against the oracle loop: loss terms, ALL
    parameter gradients, the parameter update"""
    from icm_amd.models import stf_forward
    import bench
    dev = torch.device(DEV)
    tr, x, sd0 = bench.make_workload("stf", dev)
    B = x.shape[0]
    g = torch.Generator().manual_seed(99)
    noise = {"z": torch.rand((B, 192, 4, 4), generator=g) - 0.5, "y": torch.rand((B, 384, 16, 16), generator=g) - 0.5}
    drops = _drops(S.drop_path_rates(), B, dev, 17)
    s, pnames, main, st = PT.trainable(sd0)
    dd = {k: v.to(dev).contiguous() for k, v in drops.items()}
    ro, _ = PT.hip_round_decisions(stf_forward, tr.params(), x, noise["z"].to(dev), noise["y"].to(dev), drops=dd)
    scal = tr.step(x, noise, drops).tolist()
    Lr = PT.oracle_train_step(S.stf_forward, s, x.cpu(), noise, 1, st, pnames, main, drops=drops, keep=True,
                              round_override=ro)
    fy, fz = PT.count_flips(ro, Lr["out"]["_dbg"], s)
    print(f"flips y/z {fy} {fz}")
    assert fy <= PT.near_half(Lr["out"]["_dbg"]) + 2 and fz == 0
    for k, i in (("bpp_loss", 0), ("mse_loss", 1), ("loss", 2)):
        e = abs(scal[i] - Lr[k].item()) / abs(Lr[k].item())
        print(f"  {k}: {scal[i]:.6f} vs {Lr[k].item():.6f} rel {e:.2e}")
        assert e < 5e-6, k
    hip = {n: tr.flat.gviews[n] for n in main}
    tot, worst_l2, worst_elem, rows = PT.grad_errors(hip, Lr["raw_grads"], main)
    rows.sort(key=lambda r: -r[3])
    print(f"  all {len(rows)} gradients: worst ||d||/total {worst_l2:.2e}, worst element-wise rel {worst_elem:.2e}; "
          f"top: {[(n, f'{e:.1e}') for n, _, _, e in rows[:4]]}")
    assert len(rows) == len(main)
    assert worst_l2 < GRAD_L2_TOL and worst_elem < GRAD_ELEM_TOL
    assert abs(math.sqrt(scal[5]) - tot) <= 1e-4 * tot
    l2 = PT.update_l2(dict(tr.model.named_parameters()), s, sd0, pnames)
    print(f"  relative L2 error of the update: {l2:.2e}")
    assert l2 < 5e-4


def test_b16_stf6_eval_forward_vs_oracle():
    """the zigzag variant at the batch its bench uses: eval forward only (the 24-block oracle backward at B=16 takes
    minutes on the host cores; gradients are covered at B=2 by tests/test_gpu_stf6.py)"""
    from icm_amd import engine as E
    from icm_amd.models import stf6_forward
    import bench
    tr, x, sd = bench.make_workload("stf6", torch.device(DEV))
    P = tr.params()
    keep = {}
    with torch.no_grad():
        stf6_forward(E.Tape(need_grad=False), P, x, None, None, None, keep=keep)
        med = P["entropy_bottleneck.quantiles"][:, 0, 1].reshape(1, -1, 1, 1)
        ro = {"y": torch.round(keep["y_zz"] - keep["mu"]).cpu(), "z": torch.round(keep["z"] - med).cpu()}
        x_hat, y_lik, z_lik = stf6_forward(E.Tape(need_grad=False), P, x)
        ref = S6.stf6_forward(sd, x.cpu(), None, None, keep=True, round_override=ro)
    d = ref["_dbg"]
    t = (d["y_zz"] - d["mu"]).detach()
    fy = int((torch.round(t) != ro["y"]).sum().item())
    near = int(((t - torch.floor(t) - 0.5).abs() < 1e-4).sum().item())
    print("flips y:", fy, "near-half:", near)
    assert fy <= near + 2
    assert rel(x_hat, ref["x_hat"]) < 1e-4
    assert rel(y_lik, ref["likelihoods"]["y"]) < 1e-4 and rel(z_lik, ref["likelihoods"]["z"]) < 1e-4
    Lh = O.rd_loss(x.cpu(), {"x_hat": x_hat.cpu(), "likelihoods": {"y": y_lik.cpu(), "z": z_lik.cpu()}})
    Lr = O.rd_loss(x.cpu(), ref)
    for k in ("bpp_loss", "mse_loss", "loss"):
        e = abs(Lh[k].item() - Lr[k].item()) / abs(Lr[k].item())
        print(f"  {k}: {Lh[k].item():.6f} vs {Lr[k].item():.6f} rel {e:.2e}")
        assert e < 5e-6, k
