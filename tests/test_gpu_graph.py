"""MI355X: hipGraph replay of the eval forward (icm_amd/graphs.py) equals the eager launch sequence bit for bit, picks
up in-place parameter updates, and reports the host-side speed-up at batch 1."""
import os
import sys
import time

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "image-compression-for-machine_amd"))
from oracle import weights as W  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["cnn", "stf"])
def test_graphed_forward_matches_eager(name):
    from icm_amd.graphs import GraphedForward
    from icm_amd.zoo import models
    net = models[name]()
    net.load_state_dict(W.make_wacnn_state_dict() if name == "cnn" else W.make_stf_state_dict())
    net = net.cuda().eval()
    fwd = GraphedForward(net)
    xs = [W._u(f"graph.{name}.x{i}", (1, 3, 128, 192), 0.0, 1.0).cuda() for i in range(3)]
    with torch.no_grad():
        eager = [net(x) for x in xs]
    for x, e in zip(xs, eager):
        out = fwd(x)
        assert torch.equal(out["x_hat"], e["x_hat"])
        assert torch.equal(out["likelihoods"]["y"], e["likelihoods"]["y"])
        assert torch.equal(out["likelihoods"]["z"], e["likelihoods"]["z"])
    assert len(fwd._graphs) == 1                      # one capture, two replays
    # another input shape -> a second graph; a padded (non multiple of 64) size goes through the same path
    x2 = W._u(f"graph.{name}.x2", (2, 3, 100, 72), 0.0, 1.0).cuda()
    with torch.no_grad():
        e2 = net(x2)
    o2 = fwd(x2)
    assert tuple(o2["x_hat"].shape) == (2, 3, 100, 72) and torch.equal(o2["x_hat"], e2["x_hat"])
    assert len(fwd._graphs) == 2
    # parameters changed in place (what an optimiser step does): the replay repacks and uses them
    with torch.no_grad():
        p = dict(net.named_parameters())["g_s.8.bias" if name == "cnn" else "end_conv.2.bias"]
        p.add_(0.25)
        e3 = net(xs[0])
    o3 = fwd(xs[0])
    assert torch.equal(o3["x_hat"], e3["x_hat"]) and not torch.equal(e3["x_hat"], eager[0]["x_hat"])
    net.train()
    with pytest.raises(RuntimeError):
        fwd(xs[0])


def test_graphed_forward_batch1_speedup():
    from icm_amd.graphs import GraphedForward
    from icm_amd.zoo import models
    torch.manual_seed(0)
    net = models["cnn"]().cuda().eval()
    x = torch.rand(1, 3, 256, 256, device="cuda")
    fwd = GraphedForward(net)
    fwd(x)

    def timeit(fn, n=20):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    with torch.no_grad():
        t_eager = timeit(lambda: net(x))
    t_graph = timeit(lambda: fwd(x))
    print(f"cnn eval forward, batch 1, 256x256: eager {t_eager:.2f} ms, graph replay {t_graph:.2f} ms "
          f"({t_eager / t_graph:.2f}x)")
    # at batch 1 both are bound by ~340 dependent few-microsecond kernels; the replay only removes the host's share.
    # (Timing assertions on a shared box are kept loose: this guards against a pathological replay, not a ratio.)
    assert t_graph < 1.5 * t_eager


@pytest.mark.parametrize("name,shape", [("cnn", (2, 3, 64, 64)), ("stf", (2, 3, 64, 64)), ("stf6", (1, 3, 128, 128))])
def test_step_graphed_equals_eager_steps(name, shape):
    """Trainer.step_graphed (two eager warm-up steps, then hipGraph replays with noise / DropPath scales / Adam scalars
    fed through device memory) leaves the SAME parameters as the same number of eager steps, bit for bit"""
    from icm_amd.trainer import Trainer
    from icm_amd.zoo import models
    sd = {"cnn": W.make_wacnn_state_dict, "stf": W.make_stf_state_dict, "stf6": W.make_stf6_state_dict}[name]()
    xs = [W._u(f"tg.{name}.x{i}", shape, 0.0, 1.0).cuda() for i in range(5)]
    outs = []
    for graphed in (False, True):
        net = models[name]()
        net.load_state_dict(sd)
        tr = Trainer(net, device="cuda:0", seed=5)
        losses = []
        for x in xs:
            s = tr.step_graphed(x) if graphed else tr.step(x)
            losses.append(s.clone())
        torch.cuda.synchronize()
        outs.append((torch.stack(losses).cpu(), {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()}))
        if graphed:
            assert tr.step_no == 5 and tr._graph is not None
    assert torch.equal(outs[0][0], outs[1][0]), (outs[0][0][:, 2], outs[1][0][:, 2])
    for k, v in outs[0][1].items():
        assert torch.equal(v, outs[1][1][k]), k


def test_eval_pack_cache_follows_weight_updates():
    """eval-mode module calls keep their packed weight copies between calls and drop them when a parameter changes,
    whether through torch (version counter) or through the HIP trainer (engine weight generation)"""
    from icm_amd import engine as E
    from icm_amd.trainer import Trainer
    from icm_amd.zoo import models
    net = models["cnn"]()
    net.load_state_dict(W.make_wacnn_state_dict())
    net = net.cuda().eval()
    x = W._u("pc.x", (1, 3, 64, 64), 0.0, 1.0).cuda()
    with torch.no_grad():
        a = net(x)["x_hat"].clone()
        n_packed = len(net._pack_eval)
        assert n_packed > 100
        b = net(x)["x_hat"].clone()
        assert torch.equal(a, b) and len(net._pack_eval) == n_packed          # second call: cache hits only
        dict(net.named_parameters())["g_s.8.bias"].add_(0.5)                  # torch in-place update
        c = net(x)["x_hat"].clone()
        assert not torch.equal(a, c) and len(net._pack_eval) == n_packed      # dropped and rebuilt, not grown
    net.train()
    assert net._pack_cache() is None
    tr = Trainer(net, device="cuda:0")
    tr.step(torch.rand(1, 3, 64, 64, device="cuda"))
    net.eval()
    with torch.no_grad():
        d = net(x)["x_hat"].clone()
        ref = None
        net._eval_cache_off = True
        ref = net(x)["x_hat"].clone()
        net._eval_cache_off = False
    assert torch.equal(d, ref) and not torch.equal(d, c)                      # the step's update was picked up
