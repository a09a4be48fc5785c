"""Child process of tests/test_gpu_ddp.py (not a test module): one data-parallel rank of the native trainer.

    python ddp_worker.py RANK WORLD PORT OUTDIR

Ranks share cuda:0 and talk over gloo (the one-GPU rehearsal path of GradReducer: buckets are staged through the
host), which exercises exactly the trainer code the RCCL path runs -- bucket markers during backward, parameter
broadcast, 1/world scaling, clip on the averaged gradient -- on a box with a single GPU.  Rank r trains on image r of
the 2-image batch; rank 1 deliberately starts from DIFFERENT weights (the trainer must broadcast rank 0's)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "image-compression-for-machine_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def inputs():
    from oracle import weights as W
    x = W._u("ddp.x", (2, 3, 64, 64), 0.0, 1.0)
    noises = [{"z": W._u(f"ddp.nz{i}", (2, 192, 1, 1), -0.5, 0.5), "y": W._u(f"ddp.ny{i}", (2, 320, 4, 4), -0.5, 0.5)}
              for i in range(2)]
    return x, noises


def main():
    rank, world, port, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    backend = sys.argv[5] if len(sys.argv) > 5 else "gloo"   # "nccl" = RCCL: one rank per GPU (1-rank group on a 1-GPU box)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    import torch
    import torch.distributed as dist
    from oracle import weights as W
    from icm_amd.trainer import Trainer
    from icm_amd.zoo import models
    if backend == "nccl":
        torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        net = models["cnn"]()
        net.load_state_dict(W.make_wacnn_state_dict(salt=0 if rank == 0 else 7))
        tr = Trainer(net, lr=1e-4, aux_lr=1e-4, lmbda=0.0067, clip_max_norm=1.0, device="cuda:0")
        assert tr.world == world and tr.rank == rank
        p_init = tr.flat.p.clone().cpu()
        fired = []
        launch = tr.reducer.launch
        tr.reducer.launch = lambda b, after=(): (fired.append(b), launch(b, after))[1]
        x, noises = inputs()
        scal = []
        for it in range(2):
            lo, hi = (rank, rank + 1) if world > 1 else (0, 2)   # a 1-rank group trains on the whole batch
            nz = {k: v[lo:hi] for k, v in noises[it].items()}
            scal.append(tr.step(x[lo:hi].cuda(), nz).cpu())
        torch.cuda.synchronize()
        torch.save({"p": tr.flat.p.cpu(), "ap": tr.flat.ap.cpu(), "g": tr.flat.g.cpu(), "scal": scal, "p_init": p_init,
                    "fired": fired, "backend": dist.get_backend(), "world": dist.get_world_size(),
                    "collectives": tr.reducer.launched, "side_stream": tr.reducer.stream is not None},
                   os.path.join(outdir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
