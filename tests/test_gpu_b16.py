"""GPU parity at the BENCH geometry: batch 16 x 256x256 (BASELINE.json configs[1]).

The tile choices of the implicit-GEMM and weight-gradient kernels depend on the pixel count (single-pass 192-co
tiling from 16 384 pixels, XCD-aware wgrad order, pixel-split counts, 10-12-member grouped launches, side-stream
weight gradients), so the paths `bench.py` times are only exercised at this size.  These tests rebuild exactly
`bench.make_workload` (same default-initialised weights, same x, same per-rank noise generator) and compare

  * the eval forward (y, z, mu, likelihoods, x_hat, bpp, mse),
  * two native training steps (loss / bpp / mse, all 585 gradients of step 1, the parameters after each step)

against the CPU oracle (fwd + bwd at B=16: a few seconds per call on the box's host cores).  Rounding decisions of
the HIP path are adopted by the oracle and flips are counted (tests/_parity.py), so every assertion is unconditional.
"""
import math

import pytest
import torch

import _parity as PT
from oracle import wacnn_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
# all-gradient bounds (measured on MI355X: see DESIGN.md 2): L2 error of any tensor relative to the total gradient
# norm, and the largest element-wise error of any tensor relative to that tensor's largest entry
GRAD_L2_TOL, GRAD_ELEM_TOL = 5e-5, 2e-4   # measured 5.0e-6 / 2.0e-5


def _bench():
    import bench
    return bench


def _noise(B, dev, seed):
    """what Trainer.step draws from its generator on the first step (trainer.py: nz first, then ny)"""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    nz = torch.rand((B, 192, 4, 4), dtype=torch.float32, device=dev, generator=g) - 0.5
    ny = torch.rand((B, 320, 16, 16), dtype=torch.float32, device=dev, generator=g) - 0.5
    nz2 = torch.rand((B, 192, 4, 4), dtype=torch.float32, device=dev, generator=g) - 0.5
    ny2 = torch.rand((B, 320, 16, 16), dtype=torch.float32, device=dev, generator=g) - 0.5
    return [{"z": nz, "y": ny}, {"z": nz2, "y": ny2}]


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def test_b16_eval_forward_vs_oracle():
    from icm_amd import engine as E
    from icm_amd.models import wacnn_forward
    bench = _bench()
    tr, x, sd = bench.make_workload("cnn", torch.device(DEV))
    P = tr.params()
    ro, keep = PT.hip_round_decisions(wacnn_forward, P, x, None, None)
    with torch.no_grad():
        x_hat, y_lik, z_lik = wacnn_forward(E.Tape(need_grad=False), P, x)
        ref = O.wacnn_forward(sd, x.cpu(), None, keep=True, round_override=ro)
    d = ref["_dbg"]
    fy, fz = PT.count_flips(ro, d, sd)
    print("flips y/z:", fy, fz, "| oracle latents within 1e-4 of a half:", PT.near_half(d))
    assert fy <= PT.near_half(d) + 2 and fz == 0
    for k in ("y", "z", "mu", "scale", "lat_means"):
        r = rel(keep[k], d[k])
        print(f"  {k}: rel {r:.2e}")
        assert r < 1e-4, k
    assert rel(x_hat, ref["x_hat"]) < 1e-4
    assert rel(y_lik, ref["likelihoods"]["y"]) < 1e-4
    assert rel(z_lik, ref["likelihoods"]["z"]) < 1e-4
    Lh = O.rd_loss(x.cpu(), {"x_hat": x_hat.cpu(), "likelihoods": {"y": y_lik.cpu(), "z": z_lik.cpu()}})
    Lr = O.rd_loss(x.cpu(), ref)
    for k in ("bpp_loss", "mse_loss", "loss"):
        e = abs(Lh[k].item() - Lr[k].item()) / abs(Lr[k].item())
        print(f"  {k}: {Lh[k].item():.6f} vs {Lr[k].item():.6f} rel {e:.2e}")
        assert e < 5e-6, k   # measured 1.2e-7 (north_star bound: 1e-4 relative)


@pytest.mark.parametrize("slice_split", [False, True], ids=["default", "split_slices"])
def test_b16_trainer_steps_vs_oracle(slice_split, monkeypatch):
    """the bench's own first two steps (noise=None: drawn from the trainer's generator) against the oracle loop; once
    more with the slice section in its split form (icm_amd/slices.py: wide latent-block launches, K-split input
    gradients, column-block weight gradients -- paths that only exist at this geometry)"""
    from icm_amd import models as M_
    monkeypatch.setattr(M_, "SLICE_SPLIT", slice_split)
    from icm_amd.models import wacnn_forward
    bench = _bench()
    dev = torch.device(DEV)
    tr, x, sd0 = bench.make_workload("cnn", dev)
    noises = _noise(16, dev, bench.SEED_NOISE)
    s, pnames, main, st = PT.trainable(sd0)
    xc = x.cpu()
    for it in (1, 2):
        nz, ny = noises[it - 1]["z"], noises[it - 1]["y"]
        ro, _ = PT.hip_round_decisions(wacnn_forward, tr.params(), x, nz, ny)
        scal = tr.step(x).tolist()           # the bench path: noise drawn inside
        cpu_noise = {"z": nz.cpu(), "y": ny.cpu()}
        Lr = PT.oracle_train_step(O.wacnn_forward, s, xc, cpu_noise, it, st, pnames, main, keep=True,
                                  round_override=ro)
        fy, fz = PT.count_flips(ro, Lr["out"]["_dbg"], s)
        print(f"step {it}: flips y/z {fy} {fz}")
        assert fy <= PT.near_half(Lr["out"]["_dbg"]) + 2 and fz == 0
        for k, i in (("bpp_loss", 0), ("mse_loss", 1), ("loss", 2)):
            e = abs(scal[i] - Lr[k].item()) / abs(Lr[k].item())
            print(f"  step {it} {k}: {scal[i]:.6f} vs {Lr[k].item():.6f} rel {e:.2e}")
            assert e < 5e-6, (it, k)     # measured 1.2e-7
        if it == 1:
            hip = {n: tr.flat.gviews[n] for n in main}
            tot, worst_l2, worst_elem, rows = PT.grad_errors(hip, Lr["raw_grads"], main)
            print(f"  all {len(rows)} gradients: worst ||d||/total {worst_l2:.2e}, worst element-wise rel {worst_elem:.2e}")
            rows.sort(key=lambda r: -r[3])
            for n, dd, rn, er in rows[:5]:
                print(f"    {n}: elem rel {er:.2e} ||d|| {dd:.3e} ||ref|| {rn:.3e}")
            assert len(rows) == len(main)
            assert worst_l2 < GRAD_L2_TOL and worst_elem < GRAD_ELEM_TOL
            total_h = math.sqrt(scal[5])
            assert abs(total_h - tot) <= 1e-4 * tot      # the norm the clip coefficient is computed from
            # sensitivity of this check: a 1e-3 perturbation of ONE 32x32 tile of one weight gradient must trip it
            pert = dict(hip)
            g = hip["g_a.2.weight"].clone()
            g[:32, :32] *= 1.0 + 1e-3
            pert["g_a.2.weight"] = g
            _, _, pe, _ = PT.grad_errors(pert, Lr["raw_grads"], main)
            print(f"  perturbed-tile element-wise rel: {pe:.2e}")
            assert pe > GRAD_ELEM_TOL, "a 1e-3 fault in one wgrad tile would go unnoticed"
        P = dict(tr.model.named_parameters())
        l2 = PT.update_l2(P, s, sd0, pnames)
        print(f"  step {it}: relative L2 error of the accumulated update: {l2:.2e}")
        assert l2 < 5e-4


def test_b16_bench_step_is_reproducible():
    """no float atomics anywhere: two trainers built the bench's way produce bit-identical step outputs and parameters"""
    bench = _bench()
    dev = torch.device(DEV)
    outs = []
    for _ in range(2):
        tr, x, _ = bench.make_workload("cnn", dev)
        sc = [tr.step(x).clone() for _ in range(2)]
        outs.append((sc, tr.flat.p.clone(), tr.flat.ap.clone()))
        del tr
    assert torch.equal(outs[0][0][0], outs[1][0][0]) and torch.equal(outs[0][0][1], outs[1][0][1])
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
