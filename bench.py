#!/usr/bin/env python
"""Headline benchmark: images/s of the `cnn` (WACNN) training step on synthetic 256x256 batches.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

One step = the reference training iteration (train.py:188-214): forward, R-D loss (lambda=0.0067),
backward, clip_grad_norm_(1.0), Adam on the 75.2 M parameters, aux loss + Adam on the quantiles; batch 16
per GPU (BASELINE.json configs[1]; configs[2] = 8 GPUs x 16).  Prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "image-compression-for-machine_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch
import torch.distributed as dist

FWD_GFLOP_PER_IMG = 68.87       # BASELINE.md section 2 (34.435 GMAC)
STEP_GFLOP_PER_IMG = 206.6      # fwd + dgrad + wgrad
PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
BATCH_PER_GPU = 16


def dominant_kernel_roofline(dev, iters=10):
    """Time the single heaviest kernel shape live with HIP events on the launch stream: g_a.2 forward,
    conv5x5 s2 192->192 on [16,192,128,128] (3.77 GMAC/img: SURVEY.md 8 a3)."""
    from icm_amd import engine as E
    from icm_amd.engine import VT
    x = torch.randn(BATCH_PER_GPU, 192, 128, 128, device=dev)
    w = torch.randn(192, 192, 5, 5, device=dev) * 0.02
    b = torch.zeros(192, device=dev)
    tape = E.Tape(need_grad=False)
    y = E.conv2d(tape, VT(x), w, b, stride=2, pad=2)
    torch.cuda.synchronize()
    s = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(iters):
        E.conv2d(tape, VT(x), w, b, stride=2, pad=2, out=y)
    e1.record(s)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flop = 2.0 * BATCH_PER_GPU * 192 * 192 * 25 * 64 * 64
    # HBM bytes per launch of this kernel from the committed rocprofv3 PMC passes (separate FETCH_SIZE / WRITE_SIZE
    # runs, FETCH_SIZE doubled per the gfx950 correction): tools/pmc_dominant.sh + tools/pmc_dominant_summary.py ->
    # profiles/r01_v9_pmc_dominant.json.
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_v9_pmc_dominant.json")) as fh:
            traffic = json.load(fh)["hbm_bytes_per_launch"]
    except Exception:
        pass
    return {"bound": "mfma", "kernel": "conv_igemm_kernel (g_a.2 fwd: conv5x5 s2 192->192 @ [16,192,128,128])",
            "achieved": flop / (ms * 1e-3) / 1e12, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": flop / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
            "traffic_note": "HBM bytes/launch from committed PMC passes (not re-measured in this run)",
            "avg_launch_ms": ms, "algorithmic_flop_per_launch": flop}


def cpu_baseline(sd_cpu, threads):
    """The CPU oracle (port of the reference forward/backward) timed on the host cores: B=2 sample of the
    same workload (fwd + loss + bwd), a few iterations, bounded to ~20 s."""
    from oracle import wacnn_oracle as O
    torch.set_num_threads(threads)
    B = 2
    x = torch.rand(B, 3, 256, 256)
    noise = {"z": torch.rand(B, 192, 4, 4) - 0.5, "y": torch.rand(B, 320, 16, 16) - 0.5}
    s = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and v.numel() else v) for k, v in sd_cpu.items()}

    def it():
        out = O.wacnn_forward(s, x, noise)
        O.rd_loss(x, out, 0.0067)["loss"].backward()
    it()
    t0 = time.time()
    n = 0
    while n < 3 or (time.time() - t0 < 12 and n < 20):
        it()
        n += 1
    dt = (time.time() - t0) / n
    return {"value": B / dt, "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"oracle fwd+loss+bwd, B={B} 256x256, {n} iterations, {dt*1e3:.0f} ms/iter"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fwd-only", action="store_true", help="time eval-mode forward only (reported separately)")
    ap.add_argument("--model", default="cnn", choices=["cnn", "stf"], help="cnn = BASELINE.json headline (default); "
                    "stf = configs[3], reported as an extra line")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal on a one-GPU box: ICM_BENCH_SHARE_GPU=1 puts every rank on cuda:0 with the gloo backend
    share = os.environ.get("ICM_BENCH_SHARE_GPU", "0") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if share else "nccl", rank=rank, world_size=world)
    from icm_amd.zoo import models
    from icm_amd.trainer import Trainer
    torch.manual_seed(0)
    net = models[args.model]()
    sd_cpu = {k: v.clone() for k, v in net.state_dict().items()}
    tr = Trainer(net, lr=1e-4, aux_lr=1e-4, lmbda=0.0067, clip_max_norm=1.0, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    x = torch.rand(BATCH_PER_GPU, 3, 256, 256, generator=g, device=dev)

    packed = {}   # eval forward: weights are constant, the MFMA-order copies are packed once (inference serving path)

    def one():
        if args.fwd_only:
            from icm_amd import engine as E
            from icm_amd.models import stf_forward, wacnn_forward
            (stf_forward if args.model == "stf" else wacnn_forward)(E.Tape(need_grad=False, packed_cache=packed),
                                                                    tr.params(), x)
            return None
        return tr.step(x)

    for _ in range(args.warmup):
        one()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        scal = one()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cpu" if share else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    if rank == 0:
        ips = world * BATCH_PER_GPU * args.steps / dt
        gflop = FWD_GFLOP_PER_IMG if args.fwd_only else STEP_GFLOP_PER_IMG
        if args.model == "stf":   # stf forward 33.498 GMAC = 67.0 GFLOP/image
            gflop = 67.0 if args.fwd_only else 3 * 67.0
        line = {
            "metric": ("images/sec (256x256) cnn-hyperprior (WACNN) " if args.model == "cnn" else
                       "images/sec (256x256) stf (SymmetricalTransFormer) ") + ("forward" if args.fwd_only else "training step"),
            "value": ips, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": (("cnn (WACNN N=192 M=320)" if args.model == "cnn" else "stf (Swin, embed 48, 12 slices)") +
                                    (" train step, lambda=0.0067 MSE, batch 16/GPU synthetic 256x256, Adam lr 1e-4 + "
                                     "aux Adam, clip 1.0" if not args.fwd_only else
                                     " eval forward, batch 16/GPU synthetic 256x256")),
                       "global_batch": world * BATCH_PER_GPU, "parallelism": f"dp{world}"},
            "roofline_step": {"bound": "mfma", "achieved": ips / world * gflop / 1e3, "peak": PEAK_F32_MFMA_TFLOPS,
                              "unit": "TFLOP/s", "frac": ips / world * gflop / 1e3 / PEAK_F32_MFMA_TFLOPS,
                              "algorithmic_gflop_per_image": gflop},
        }
        if scal is not None:
            v = scal.tolist()
            line["last_step"] = {"bpp": v[0], "mse": v[1], "loss": v[2], "aux": v[6]}
        line["roofline"] = dominant_kernel_roofline(dev)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sd_cpu, min(os.cpu_count() or 1, 16))
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
