"""GPU: two data-parallel ranks of the native trainer (fresh child processes sharing cuda:0, gloo) against one rank on
the whole batch.  Covers what tests/test_ddp_gloo.py cannot without kernels: a real Trainer.step per rank, bucket
markers firing during backward, parameter broadcast from rank 0, mean-of-shards == full-batch semantics, and replicas
that stay BIT-identical (fixed-order reductions: every rank derives the same clip coefficient)."""
import os
import socket
import subprocess
import sys
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_equal_one_rank_on_the_full_batch():
    sys.path.insert(0, HERE)
    import ddp_worker
    from oracle import weights as W
    from icm_amd.trainer import Trainer
    from icm_amd.zoo import models
    port = str(_free_port())
    with tempfile.TemporaryDirectory() as out:
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "ddp_worker.py"), str(r), "2", port, out], env=env)
                 for r in range(2)]
        rcs = [p.wait(timeout=900) for p in procs]
        assert rcs == [0, 0], rcs
        r0 = torch.load(os.path.join(out, "rank0.pt"), weights_only=True)
        r1 = torch.load(os.path.join(out, "rank1.pt"), weights_only=True)
    # parameters were broadcast from rank 0 at construction (rank 1 was built from other weights)
    assert torch.equal(r0["p_init"], r1["p_init"])
    # the four bucket all-reduces fire in completion order: synthesis | slice chains | hyper path | analysis
    assert r0["fired"] == [0, 1, 2, 3] * 2 and r1["fired"] == r0["fired"]
    # the replicas never drift: same all-reduced gradient, same clip coefficient, same update -- bit for bit
    assert torch.equal(r0["g"], r1["g"])
    assert torch.equal(r0["p"], r1["p"]) and torch.equal(r0["ap"], r1["ap"])
    # one rank on the whole batch
    net = models["cnn"]()
    sd = W.make_wacnn_state_dict()
    net.load_state_dict(sd)
    tr = Trainer(net, lr=1e-4, aux_lr=1e-4, lmbda=0.0067, clip_max_norm=1.0, device="cuda:0")
    p0 = tr.flat.p.clone().cpu()
    assert torch.equal(p0, r0["p_init"])
    x, noises = ddp_worker.inputs()
    scal = [tr.step(x.cuda(), noises[it]).cpu() for it in range(2)]
    for it in range(2):
        for i, name in ((0, "bpp"), (1, "mse"), (2, "loss")):
            mean = 0.5 * (r0["scal"][it][i].item() + r1["scal"][it][i].item())
            full = scal[it][i].item()
            assert abs(mean - full) <= 2e-5 * abs(full), (it, name, mean, full)
        # gradient norm of the averaged gradient == full-batch gradient norm
        n2, nf = r0["scal"][it][5].item() / 4.0, scal[it][5].item()   # ranks hold the SUM over 2 shards: (2 g)^2 = 4 g^2
        assert abs(n2 - nf) <= 1e-4 * nf, (it, n2, nf)
    p1 = tr.flat.p.cpu()
    upd = (p1 - p0).double()
    err = (r0["p"].double() - p1.double())
    l2 = err.norm().item() / upd.norm().item()
    print(f"2 ranks vs 1 rank: relative L2 error of the 2-step update {l2:.2e}, max abs parameter diff {err.abs().max().item():.2e}")
    assert l2 < 3e-4 and err.abs().max().item() < 2e-5     # measured 3.2e-5 / 5.2e-6
    assert (r0["ap"] - tr.flat.ap.cpu()).abs().max().item() < 1e-6


def test_rccl_one_rank_collective_path_is_bit_identical():
    """The RCCL branch of GradReducer (side-stream async all_reduce, event ordering against the weight-gradient side
    stream, Work.wait() in finish(), device-tensor broadcast) on a one-GPU box: a fresh child process initialises a
    1-rank `nccl` process group with ICM_FORCE_COLLECTIVES=1, so every bucket goes through dist.all_reduce on the
    reducer's stream.  A sum over one rank is the identity: parameters, gradients and loss scalars after two steps must
    equal the group-less run BIT FOR BIT."""
    sys.path.insert(0, HERE)
    import ddp_worker
    from oracle import weights as W
    from icm_amd.trainer import Trainer
    from icm_amd.zoo import models
    port = str(_free_port())
    with tempfile.TemporaryDirectory() as out:
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", ICM_FORCE_COLLECTIVES="1")
        p = subprocess.Popen([sys.executable, os.path.join(HERE, "ddp_worker.py"), "0", "1", port, out, "nccl"], env=env)
        assert p.wait(timeout=900) == 0
        r0 = torch.load(os.path.join(out, "rank0.pt"), weights_only=True)
    assert r0["backend"] == "nccl" and r0["world"] == 1
    assert r0["side_stream"] and r0["collectives"] == 8 and r0["fired"] == [0, 1, 2, 3] * 2
    net = models["cnn"]()
    net.load_state_dict(W.make_wacnn_state_dict())
    tr = Trainer(net, lr=1e-4, aux_lr=1e-4, lmbda=0.0067, clip_max_norm=1.0, device="cuda:0")
    assert not tr.reducer.active
    x, noises = ddp_worker.inputs()
    scal = [tr.step(x.cuda(), noises[it]).cpu() for it in range(2)]
    torch.cuda.synchronize()
    assert torch.equal(r0["p"], tr.flat.p.cpu()) and torch.equal(r0["ap"], tr.flat.ap.cpu())
    assert torch.equal(r0["g"], tr.flat.g.cpu())
    for it in range(2):
        assert torch.equal(r0["scal"][it], scal[it])
