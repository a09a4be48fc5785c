"""CPU, world_size=2 over gloo: the data-parallel plumbing of icm_amd.trainer (bucket layout of the flat
gradient buffer, bucketed all-reduce, mean-of-shards semantics).  Kernels are not involved here."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "image-compression-for-machine_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from icm_amd.trainer import BUCKETS, FlatParams, GradReducer
        from icm_amd.zoo import models
        torch.manual_seed(0)
        net = models["cnn"]()
        flat = FlatParams(net, torch.device("cpu"))
        # every main parameter lives in exactly one bucket range; ranges tile the flat buffer
        assert flat.bucket_ranges[0][0] == 0 and flat.bucket_ranges[-1][1] == flat.n_main
        for (a0, b0), (a1, b1) in zip(flat.bucket_ranges, flat.bucket_ranges[1:]):
            assert b0 == a1
        assert sum(p.numel() for _, p in flat.main) == 75235779 - 192 * 3
        names0 = {n.split(".")[0] for n, _ in flat.main}
        assert names0 <= {p for b in BUCKETS for p in b} and len(flat.bucket_ranges) == len(BUCKETS)
        # the stf model maps onto the same four buckets (synthesis | slice chains | hyper | analysis), none left over
        sflat = FlatParams(models["stf"](), torch.device("cpu"))
        assert len(sflat.bucket_ranges) == len(BUCKETS) and sflat.bucket_ranges[-1][1] == sflat.n_main
        assert {n.split(".")[0] for n, _ in sflat.main} <= {p for b in BUCKETS for p in b}
        assert all(b > a for a, b in sflat.bucket_ranges)
        # parameters are views of the flat buffer
        flat.p.fill_(1.5)
        assert all(float(p.data.flatten()[0]) == 1.5 for _, p in flat.main)
        red = GradReducer(flat.g, flat.bucket_ranges)
        assert red.world == world
        flat.g.copy_(torch.arange(flat.n_main, dtype=torch.float32) % 97 * (rank + 1))
        expect = torch.arange(flat.n_main, dtype=torch.float32) % 97 * sum(r + 1 for r in range(world))
        for b in range(len(flat.bucket_ranges)):   # launched bucket by bucket as backward would
            red.launch(b)
        red.finish()
        assert torch.equal(flat.g, expect)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_flat_buckets_and_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def _step_worker(rank, world, port, q):
    """a full Trainer.step per rank on CPU with kernel launches stubbed (host logic + gloo collectives are real)"""
    for p in (ROOT, os.path.join(ROOT, "image-compression-for-machine_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from icm_amd import _lib, engine as E
        real = _lib.lib()

        class Fake:
            def __getattr__(self, name):
                if name in ("icm_packed_weight_floats", "icm_wgrad_workspace_floats", "icm_wgrad_workspace_floats_grouped",
                            "icm_winattn_bwd_workspace_floats", "icm_strerror"):
                    return getattr(real, name)
                return lambda *a: 0
        fake = Fake()
        _lib.lib = lambda: fake
        _lib.stream = lambda: 0

        def bs(t):
            if t is None:
                return 0
            return t.stride()[0] if t.dim() == 4 else t[0].numel()
        _lib.bs = bs
        E.bs = bs
        from icm_amd.trainer import Trainer
        from icm_amd.zoo import models
        torch.manual_seed(100 + rank)          # ranks start from DIFFERENT weights ...
        tr = Trainer(models["cnn"](), device="cpu", seed=5)
        assert tr.world == world and tr.rank == rank and tr.side is None
        ref = [torch.zeros_like(tr.flat.p) for _ in range(world)]
        dist.all_gather(ref, tr.flat.p)
        assert torch.equal(ref[0], ref[1]), "parameters were not broadcast from rank 0"   # ... and end up with rank 0's
        fired = []
        launch = tr.reducer.launch

        def spy(b, after=()):
            # stand-in for the kernels: this rank's gradient of bucket b is (rank + 1) everywhere
            a, e = tr.flat.bucket_ranges[b]
            tr.flat.g[a:e] = float(rank + 1)
            fired.append(b)
            launch(b, after)
        tr.reducer.launch = spy
        n1 = torch.rand(1, generator=tr.gen)
        tr.step(torch.rand(2, 3, 64, 64))
        assert fired == [0, 1, 2, 3], fired
        # every bucket was sum-all-reduced: 1 + 2 on every element, on both ranks
        assert torch.equal(tr.flat.g, torch.full_like(tr.flat.g, 3.0))
        # per-rank noise generators differ (seed + rank)
        ns = [torch.zeros(1) for _ in range(world)]
        dist.all_gather(ns, n1)
        assert ns[0].item() != ns[1].item()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_trainer_step_world2_dry_run():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_step_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
