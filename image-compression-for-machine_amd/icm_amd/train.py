"""``python -m icm_amd.train`` -- the reference's training script (train.py:170-530) driving the HIP hot path.

Same flags (-m/-d/-e/-lr/-n/--lambda/--batch-size/--test-batch-size/--aux-learning-rate/--patch-size/--save/
--save_path/--seed/--clip_max_norm/--checkpoint), same epoch structure: ``train_one_epoch`` (train.py:170-232),
``test_epoch`` every fifth epoch (:235-283, :505-507), ``ReduceLROnPlateau(min, factor 0.6, patience 6)`` on the test
loss (:438), best-only checkpoints ``<save_path><epoch>.ckpt`` holding epoch / state_dict / loss / optimizer state
(:512-527).  What differs, on purpose:

* one optimisation step is ``icm_amd.trainer.Trainer.step`` -- forward, R-D loss, backward, clip, Adam, aux loss and aux
  Adam as fused HIP launches on flat buffers -- instead of five Python-level calls (identical arithmetic; parity with the
  reference loop is a GPU test); the learning rate of ReduceLROnPlateau is applied through ``Trainer.lr``;
* ``--split``/``--test-split`` replace the hard-coded ``val2017`` folder name (:404-405); ``--random-crop`` selects the
  ``RandomCrop(pad_if_needed)`` transform the reference has commented out (:393-395) instead of ``CenterCrop``;
* the loss is the published ``lmbda * 255^2 * mse + bpp`` (train_czigzag.py:63,71; default 0.0067);
* under ``torch.distributed.run`` (WORLD_SIZE > 1) every rank trains on its shard of each batch (DistributedSampler)
  and gradients are all-reduced over RCCL by the Trainer; the reference is single-process;
* checkpoints are read with ``weights_only=True``."""
from __future__ import annotations

import argparse
import os
import random
import sys
import time

import torch
from torch.utils.data import DataLoader

from .datasets import CenterCrop, Compose, ImageFolder, RandomCrop, ToTensor
from .losses import RateDistortionLoss
from .trainer import Trainer
from .zoo import models


class AverageMeter:
    """train.py:79-92"""

    def __init__(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


class PlateauLR:
    """optim.lr_scheduler.ReduceLROnPlateau(mode="min", factor, patience) with torch's defaults (relative threshold 1e-4,
    no cooldown, min_lr 0, eps 1e-8) acting on ``Trainer.lr`` (train.py:438,507)"""

    def __init__(self, trainer: Trainer, factor: float = 0.6, patience: int = 6, threshold: float = 1e-4):
        self.trainer, self.factor, self.patience, self.threshold = trainer, factor, patience, threshold
        self.best, self.num_bad = float("inf"), 0

    def step(self, metric: float) -> None:
        if metric < self.best * (1.0 - self.threshold):
            self.best, self.num_bad = metric, 0
        else:
            self.num_bad += 1
        if self.num_bad > self.patience:
            new_lr = self.trainer.lr * self.factor
            if self.trainer.lr - new_lr > 1e-8:
                self.trainer.lr = new_lr
            self.num_bad = 0

    def state_dict(self):
        return {"best": self.best, "num_bad_epochs": self.num_bad, "factor": self.factor, "patience": self.patience}

    def load_state_dict(self, sd):
        self.best, self.num_bad = float(sd["best"]), int(sd["num_bad_epochs"])


def train_one_epoch(trainer: Trainer, train_dataloader, epoch: int, log_every: int = 600) -> None:
    """train.py:170-232"""
    start = time.time()
    for i, d in enumerate(train_dataloader):
        x = d.to(trainer.device, non_blocking=True).contiguous()
        s = trainer.step(x)
        if i % log_every == 0 and trainer.rank == 0:
            v = s.tolist()   # the only host sync of the loop
            dt, start = time.time() - start, time.time()
            print(f"Train epoch {epoch}: [{i * len(x)}/{len(train_dataloader.dataset)} "
                  f"({100. * i / len(train_dataloader):.0f}%)]\tLoss: {v[2]:.3f} |\tMSE loss: {v[1]:.3f} |"
                  f"\tBpp loss: {v[0]:.2f} |\tAux loss: {v[6]:.2f} |\ttime: {dt:.1f}", flush=True)


@torch.no_grad()
def test_epoch(epoch: int, test_dataloader, model, criterion, verbose: bool = True) -> float:
    """train.py:235-283"""
    model.eval()
    device = next(model.parameters()).device
    loss, bpp_loss, mse_loss, aux_loss = AverageMeter(), AverageMeter(), AverageMeter(), AverageMeter()
    for d in test_dataloader:
        x = d.to(device)
        out = criterion(model(x), x)
        aux_loss.update(float(model.aux_loss()))
        bpp_loss.update(float(out["bpp_loss"]))
        loss.update(float(out["loss"]))
        mse_loss.update(float(out["mse_loss"]))
    model.train()
    if verbose:
        print(f"Test epoch {epoch}: Average losses:\tLoss: {loss.avg:.3f} |\tMSE loss: {mse_loss.avg * 255 ** 2:.3f} |"
              f"\tBpp loss: {bpp_loss.avg:.2f} |\tAux loss: {aux_loss.avg:.2f}\n", flush=True)
    return loss.avg


def save_checkpoint(state, filename: str) -> None:
    """train.py:286-289"""
    os.makedirs(os.path.dirname(filename) or ".", exist_ok=True)
    torch.save(state, filename)


def parse_args(argv):
    p = argparse.ArgumentParser(description="Training script (HIP path).")
    p.add_argument("-m", "--model", default="cnn", choices=sorted(models.keys()), help="Model architecture (default: %(default)s)")
    p.add_argument("-d", "--dataset", type=str, required=True, help="Training dataset (root of the split folders)")
    p.add_argument("--split", type=str, default="train", help="training split folder (default: %(default)s)")
    p.add_argument("--test-split", type=str, default="test", help="test split folder (default: %(default)s)")
    p.add_argument("-e", "--epochs", default=100, type=int, help="Number of epochs (default: %(default)s)")
    p.add_argument("-lr", "--learning-rate", default=1e-4, type=float, help="Learning rate (default: %(default)s)")
    p.add_argument("-n", "--num-workers", type=int, default=4, help="Dataloaders threads (default: %(default)s)")
    p.add_argument("--lambda", dest="lmbda", type=float, default=0.0067, help="Bit-rate distortion parameter (default: %(default)s)")
    p.add_argument("--batch-size", type=int, default=16, help="Batch size per GPU (default: %(default)s)")
    p.add_argument("--test-batch-size", type=int, default=16, help="Test batch size (default: %(default)s)")
    p.add_argument("--aux-learning-rate", default=1e-4, type=float, help="Auxiliary loss learning rate (default: %(default)s)")
    p.add_argument("--patch-size", type=int, nargs=2, default=(256, 256), help="Size of the patches to be cropped (default: %(default)s)")
    p.add_argument("--random-crop", action="store_true", help="RandomCrop(pad_if_needed) instead of CenterCrop for training")
    p.add_argument("--save", action="store_true", default=False, help="Save model to disk")
    p.add_argument("--save_path", type=str, default="./checkpoints/", help="Where to Save model")
    p.add_argument("--seed", type=float, help="Set random seed for reproducibility")
    p.add_argument("--clip_max_norm", default=1.0, type=float, help="gradient clipping max norm (default: %(default)s)")
    p.add_argument("--checkpoint", type=str, default=None, help="Path to a checkpoint")
    p.add_argument("--test-every", type=int, default=5, help="test / checkpoint every N epochs (train.py:505; default: %(default)s)")
    p.add_argument("--max-steps", type=int, default=0, help="stop each epoch after N iterations (0 = full epoch)")
    return p.parse_args(argv)


class _Limited:
    def __init__(self, loader, n):
        self.loader, self.n, self.dataset = loader, n, loader.dataset

    def __len__(self):
        return min(self.n, len(self.loader))

    def __iter__(self):
        for i, d in enumerate(self.loader):
            if i >= self.n:
                return
            yield d


def main(argv) -> int:
    args = parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if rank == 0:
        print(args)
    if not torch.cuda.is_available():
        print("Error: no GPU (the HIP path has no CPU fallback).", file=sys.stderr)
        return 3
    if args.seed is not None:
        torch.manual_seed(args.seed)
        random.seed(args.seed)
    device = f"cuda:{local}"
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device(device))

    crop = RandomCrop(args.patch_size, pad_if_needed=True) if args.random_crop else CenterCrop(args.patch_size)
    train_dataset = ImageFolder(args.dataset, split=args.split, transform=Compose([crop, ToTensor()]))
    test_dataset = ImageFolder(args.dataset, split=args.test_split, transform=Compose([CenterCrop(args.patch_size), ToTensor()]))
    sampler = None
    if world > 1:
        from torch.utils.data.distributed import DistributedSampler
        sampler = DistributedSampler(train_dataset, num_replicas=world, rank=rank, shuffle=True, drop_last=True)
    train_dataloader = DataLoader(train_dataset, batch_size=args.batch_size, num_workers=args.num_workers,
                                  shuffle=sampler is None, sampler=sampler, pin_memory=True, drop_last=world > 1)
    test_dataloader = DataLoader(test_dataset, batch_size=args.test_batch_size, num_workers=args.num_workers,
                                 shuffle=False, pin_memory=True)
    if args.max_steps > 0:
        train_dataloader = _Limited(train_dataloader, args.max_steps)

    net = models[args.model]()
    last_epoch = 0
    ck = None
    if args.checkpoint:   # train.py:453-485
        if rank == 0:
            print("Loading", args.checkpoint)
        ck = torch.load(args.checkpoint, map_location="cpu", weights_only=True)
        last_epoch = int(ck["epoch"]) + 1
        net.load_state_dict(ck["state_dict"])
    trainer = Trainer(net, lr=args.learning_rate, aux_lr=args.aux_learning_rate, lmbda=args.lmbda,
                      clip_max_norm=args.clip_max_norm, device=device, seed=int(args.seed or 0))
    lr_scheduler = PlateauLR(trainer, factor=0.6, patience=6)
    criterion = RateDistortionLoss(lmbda=args.lmbda)
    if ck is not None and "optimizer" in ck and isinstance(ck["optimizer"], dict) and "m" in ck["optimizer"]:
        trainer.load_optimizer_state(ck["optimizer"], ck.get("aux_optimizer"))
        if "lr_scheduler" in ck:
            lr_scheduler.load_state_dict(ck["lr_scheduler"])

    best_loss = float("inf")
    for epoch in range(last_epoch, args.epochs):
        if rank == 0:
            print(f"Learning rate: {trainer.lr}")
        if sampler is not None:
            sampler.set_epoch(epoch)
        train_one_epoch(trainer, train_dataloader, epoch)
        if epoch % args.test_every == 0:
            loss = test_epoch(epoch, test_dataloader, trainer.model, criterion, verbose=rank == 0)
            lr_scheduler.step(loss)
            is_best = loss < best_loss
            best_loss = min(loss, best_loss)
            if args.save and is_best and rank == 0:
                opt, aux = trainer.optimizer_state()
                save_checkpoint({"epoch": epoch, "state_dict": trainer.model.state_dict(), "loss": loss, "optimizer": opt,
                                 "aux_optimizer": aux, "lr_scheduler": lr_scheduler.state_dict()},
                                os.path.join(args.save_path, f"{epoch}.ckpt"))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
