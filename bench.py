#!/usr/bin/env python
"""Headline benchmark: images/s of the `cnn` (WACNN) training step on synthetic 256x256 batches.

    python bench.py --gpus N --steps K --warmup W

N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (the
driver's way: WORLD_SIZE is set) or started plainly, in which case this script starts that launcher itself as a
child process BEFORE anything touches the GPU and exits with its code.

One step = the reference training iteration (train.py:188-214): forward, R-D loss (lambda=0.0067),
backward, clip_grad_norm_(1.0), Adam on the 75.2 M parameters, aux loss + Adam on the quantiles; batch 16
per GPU (BASELINE.json configs[1]; configs[2] = 8 GPUs x 16).  Prints ONE JSON line:

  value / ms_per_step   whole-job images/s over the timed region (barrier + synchronize on both sides, max over ranks)
  forward / stf         (1 GPU, cnn line only) sub-objects: eval forward of the same model on the same batch, and the stf
                        model's training step + eval forward, each {value, ms_per_step, roofline_step}
  dist                  backend and world size the collective layer saw (RCCL = "nccl")
  roofline_step         the whole step against the f32-MFMA roofline (206.6 GFLOP / image)
  roofline              the kernel FAMILY (implicit-GEMM conv = forward + input gradients | weight gradients) that is
                        furthest below the roofline among those with >= 15 % of the step's kernel time: algorithmic
                        FLOP of its launches / their HIP-event time, measured live in one extra, serialised step
  roofline_shapes       the same per launch shape (top shapes by time share)
  cpu_baseline          the CPU oracle (port of the reference) on the host cores, same batch
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "image-compression-for-machine_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

FWD_GFLOP_PER_IMG = 68.87       # BASELINE.md section 2 (34.435 GMAC)
STEP_GFLOP_PER_IMG = 206.6      # fwd + dgrad + wgrad
PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
BATCH_PER_GPU = 16
SEED_X, SEED_NOISE = 1234, 4321


def make_workload(model_name, dev, rank=0, group=None, batch=BATCH_PER_GPU):
    """(trainer, x, cpu state-dict): default-initialised model under torch.manual_seed(0) (identical on every rank),
    x = rand(batch,3,256,256) from a per-rank generator (seed 1234 + rank), training noise from the trainer's own
    per-rank generator (seed 4321 + rank).  tests/test_gpu_b16.py rebuilds exactly this and checks it against the
    CPU oracle."""
    import torch
    from icm_amd.zoo import models
    from icm_amd.trainer import Trainer
    torch.manual_seed(0)
    net = models[model_name]()
    sd_cpu = {k: v.clone() for k, v in net.state_dict().items()}
    tr = Trainer(net, lr=1e-4, aux_lr=1e-4, lmbda=0.0067, clip_max_norm=1.0, device=dev, group=group, seed=SEED_NOISE)
    g = torch.Generator(device=dev)
    g.manual_seed(SEED_X + rank)
    x = torch.rand(batch, 3, 256, 256, generator=g, device=dev)
    return tr, x, sd_cpu


def family_of(label):
    if label.startswith("wgrad"):
        return "wgrad_* kernels (direct + wgrad_wino_kernel) + wgrad_reduce_kernel (weight gradients)"
    if label.startswith("winattn"):
        return "winattn kernels (window attention core)"
    if label.startswith("pack"):
        return "pack_weights kernels"
    return "conv_igemm_kernel + conv1x1_kernel + conv_ks8_kernel + conv_wino_kernel (forward + input gradients)"


def shape_table(run_step, top=24):
    """One extra step with every MFMA-family launch bracketed by HIP events on its launch stream (engine.PROFILE),
    weight gradients serialised on the main stream so that each launch is timed with the chip to itself."""
    import torch
    from icm_amd import engine as E
    E.PROFILE = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    recs, E.PROFILE = E.PROFILE, None
    agg, fam = {}, {}
    for label, flop, e0, e1 in recs:
        ms = e0.elapsed_time(e1)
        a = agg.setdefault(label, [0.0, 0.0, 0])
        a[0] += ms; a[1] += flop; a[2] += 1
        f = fam.setdefault(family_of(label), [0.0, 0.0, 0])
        f[0] += ms; f[1] += flop; f[2] += 1
    tot = sum(v[0] for v in agg.values())
    rows = []
    dump = os.environ.get("ICM_SHAPE_TABLE")     # full table (every shape) as JSON lines, for profiles/
    if dump:
        with open(dump, "w") as fh:
            for label, (ms, flop, cnt) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
                fh.write(json.dumps({"shape": label, "launches": cnt, "ms": round(ms, 4), "share": round(ms / tot, 4),
                                     "gflop": round(flop / 1e9, 3), "tflops": round(flop / ms / 1e9, 2) if ms > 0 else 0.0})
                         + "\n")
    for label, (ms, flop, cnt) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
        tf = flop / ms / 1e9 if ms > 0 else 0.0
        rows.append({"shape": label, "launches": cnt, "ms": round(ms, 4), "share": round(ms / tot, 4),
                     "gflop": round(flop / 1e9, 3), "tflops": round(tf, 2), "frac": round(tf / PEAK_F32_MFMA_TFLOPS, 4)})
    fams = {}
    for name, (ms, flop, cnt) in fam.items():
        tf = flop / ms / 1e9 if ms > 0 else 0.0
        fams[name] = {"launches": cnt, "ms": round(ms, 3), "share": round(ms / tot, 4), "gflop": round(flop / 1e9, 2),
                      "tflops": round(tf, 2), "frac": round(tf / PEAK_F32_MFMA_TFLOPS, 4)}
    return rows, fams, tot, wall


def pmc_traffic(kind):
    """HBM bytes per launch of the family's heaviest launch from the committed rocprofv3 PMC passes of this round
    (separate FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE doubled per the gfx950 correction; tools/pmc_family.sh)."""
    try:
        with open(os.path.join(ROOT, "profiles", "r03_pmc_family.json")) as fh:
            return json.load(fh).get(kind)
    except Exception:
        return None


def cpu_baseline(sd_cpu, threads, x_cpu):
    """The CPU oracle (port of the reference forward / backward) timed on the host cores on the SAME batch
    (B=16, 256x256): forward + R-D loss + backward, 1 warm-up + 3 timed iterations."""
    import torch
    from oracle import wacnn_oracle as O
    torch.set_num_threads(threads)
    B = x_cpu.shape[0]
    g = torch.Generator().manual_seed(7)
    noise = {"z": torch.rand(B, 192, 4, 4, generator=g) - 0.5, "y": torch.rand(B, 320, 16, 16, generator=g) - 0.5}
    s = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and v.numel() else v) for k, v in sd_cpu.items()}

    def it():
        for v in s.values():
            if isinstance(v, torch.Tensor) and v.requires_grad:
                v.grad = None
        out = O.wacnn_forward(s, x_cpu, noise)
        O.rd_loss(x_cpu, out, 0.0067)["loss"].backward()
    it()
    t0 = time.time()
    n = 0
    while n < 3:
        it()
        n += 1
    dt = (time.time() - t0) / n
    return {"value": B / dt, "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"oracle fwd+loss+bwd on the bench batch, B={B} 256x256, 1 warm-up + {n} iterations, "
                      f"{dt*1e3:.0f} ms/iter"}


def timed(fn, warmup, steps):
    """ms per call of fn(): warm-up calls, then `steps` calls bracketed by device synchronisation"""
    import torch
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def roofline_of(ips, gflop_per_image):
    tf = ips * gflop_per_image / 1e3
    return {"bound": "mfma", "achieved": tf, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": tf / PEAK_F32_MFMA_TFLOPS, "algorithmic_gflop_per_image": gflop_per_image}


def forward_bench(model_name, params, x, warmup=2, steps=10):
    """eval-mode forward (cnn.py:141-189 / stf.py:582-645) on the resident batch x with the constant-weight packed
    cache of the inference path: {value img/s, ms_per_step, roofline_step}"""
    from icm_amd import engine as E
    from icm_amd.models import stf6_forward, stf_forward, wacnn_forward
    fwd = {"cnn": wacnn_forward, "stf": stf_forward, "stf6": stf6_forward}[model_name]
    packed = {}
    ms = timed(lambda: fwd(E.Tape(need_grad=False, packed_cache=packed), params, x), warmup, steps)
    ips = x.shape[0] / ms * 1e3
    gf = {"cnn": FWD_GFLOP_PER_IMG, "stf": 67.0}.get(model_name)
    out = {"value": ips, "unit": "images/s", "ms_per_step": ms, "steps": steps, "warmup": warmup,
           "batch": int(x.shape[0])}
    if gf is not None:
        out["roofline_step"] = roofline_of(ips, gf)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-shape-table", action="store_true")
    ap.add_argument("--graph", action="store_true", help="train steps replayed from a captured hipGraph "
                    "(Trainer.step_graphed; single GPU): reported as an extra line")
    ap.add_argument("--fwd-only", action="store_true", help="time eval-mode forward only (reported separately)")
    ap.add_argument("--model", default="cnn", choices=["cnn", "stf", "stf6"], help="cnn = BASELINE.json headline (default); "
                    "stf = configs[3], stf6 = the zigzag variant (SURVEY 8 f3): reported as extra lines")
    ap.add_argument("--no-extras", action="store_true", help="skip the forward-only and stf sub-objects of the default "
                    "1-GPU cnn line")
    args = ap.parse_args()
    if args.gpus < 1:
        sys.exit("--gpus must be >= 1")
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        # plain `python bench.py --gpus N`: start one rank per GPU as CHILD processes (nothing here has touched the GPU
        # yet; a GPU-initialised process must never exec) and pass their exit code on
        port = os.environ.get("MASTER_PORT", "29541")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    world = int(env_world or "1")
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    import torch.distributed as dist
    # rehearsal on a one-GPU box: ICM_BENCH_SHARE_GPU=1 puts every rank on cuda:0 with the gloo backend
    share = os.environ.get("ICM_BENCH_SHARE_GPU", "0") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if share else "nccl", rank=rank, world_size=world)
    tr, x, sd_cpu = make_workload(args.model, dev, rank)

    packed = {}   # eval forward: weights are constant, the MFMA-order copies are packed once (inference serving path)

    def one():
        if args.fwd_only:
            from icm_amd import engine as E
            from icm_amd.models import stf6_forward, stf_forward, wacnn_forward
            {"cnn": wacnn_forward, "stf": stf_forward, "stf6": stf6_forward}[args.model](
                E.Tape(need_grad=False, packed_cache=packed), tr.params(), x)
            return None
        return tr.step_graphed(x) if args.graph else tr.step(x)

    first = None
    for i in range(args.warmup):
        r = one()
        if i == 0 and r is not None:
            first = r.tolist()
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
    last = scal.tolist() if scal is not None else None

    # ---- per-shape / per-family roofline: one extra step outside the timed region (every rank runs it: it contains
    # the same collectives), weight gradients serialised on the main stream
    rows = fams = None
    if not args.no_shape_table:
        side = tr.side
        tr.side = None
        rows, fams, ktot, wall = shape_table(one)
        tr.side = side

    if rank == 0:
        ips = world * BATCH_PER_GPU * args.steps / dt
        gflop = FWD_GFLOP_PER_IMG if args.fwd_only else STEP_GFLOP_PER_IMG
        if args.model == "stf":   # stf forward 33.498 GMAC = 67.0 GFLOP/image
            gflop = 67.0 if args.fwd_only else 3 * 67.0
        if args.model == "stf6":  # no survey figure: algorithmic FLOP of the GEMM / attention launches of the profiled step
            gflop = (sum(v["gflop"] for v in fams.values()) / BATCH_PER_GPU) if fams else None
        line = {
            "metric": {"cnn": "images/sec (256x256) cnn-hyperprior (WACNN) ",
                       "stf": "images/sec (256x256) stf (SymmetricalTransFormer) ",
                       "stf6": "images/sec (256x256) stf6 (SymmetricalTransFormer3, zigzag) "}[args.model] +
                      ("forward" if args.fwd_only else "training step"),
            "value": ips, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ({"cnn": "cnn (WACNN N=192 M=320)", "stf": "stf (Swin, embed 48, 12 slices)",
                                     "stf6": "stf6 (Swin, embed 48, 24 zigzag blocks, mu_Swin refinement)"}[args.model] +
                                    (" train step, lambda=0.0067 MSE, batch 16/GPU synthetic 256x256, Adam lr 1e-4 + "
                                     "aux Adam, clip 1.0" if not args.fwd_only else
                                     " eval forward, batch 16/GPU synthetic 256x256")),
                       "global_batch": world * BATCH_PER_GPU, "parallelism": f"dp{world}"},
            # what the collective layer saw (RCCL = backend "nccl" on ROCm); None / 1 for a single process
            "dist": {"backend": dist.get_backend() if world > 1 else None,
                     "world_size": dist.get_world_size() if world > 1 else 1},
            "roofline_step": roofline_of(ips / world, gflop) if gflop is not None else None,
        }
        if args.graph:
            line["config"]["workload"] += " [hipGraph replay: Trainer.step_graphed]"
        if first is not None:
            line["first_step"] = {"bpp": first[0], "mse": first[1], "loss": first[2], "aux": first[6]}
        if last is not None:
            line["last_step"] = {"bpp": last[0], "mse": last[1], "loss": last[2], "aux": last[6]}
        if fams:
            cand = {k: v for k, v in fams.items() if v["share"] >= 0.15 and v["gflop"] > 0}
            name = min(cand, key=lambda k: cand[k]["frac"]) if cand else max(fams, key=lambda k: fams[k]["share"])
            f = fams[name]
            kind = "wgrad" if name.startswith("wgrad") else "conv"
            line["roofline"] = {
                "bound": "mfma", "kernel": name, "achieved": f["tflops"], "peak": PEAK_F32_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": f["frac"], "traffic": pmc_traffic(kind),
                "share_of_step_kernel_time": f["share"], "launches": f["launches"],
                "algorithmic_gflop": f["gflop"], "family_ms": f["ms"],
                "avg_launch_ms": f["ms"] / max(1, f["launches"]),
                "note": "family aggregate over one serialised step: sum of algorithmic FLOP of its launches / sum of "
                        "their HIP-event durations on the launch stream; traffic = HBM bytes of the family's heaviest "
                        "launch from the committed PMC passes (profiles/r03_pmc_family.json), null if absent"}
            line["roofline_families"] = fams
            line["roofline_shapes"] = rows
            line["profiled_step"] = {"kernel_ms_sum": round(ktot, 3), "wall_ms": round(wall, 3)}
        # ---- sub-objects of the default driver line (1 GPU, cnn train step): the eval forward of the same model on the
        # same batch (the north-star target is stated on the forward) and the stf model's train step + forward
        if world == 1 and args.model == "cnn" and not args.fwd_only and not args.graph and not args.no_extras:
            line["forward"] = forward_bench("cnn", tr.params(), x)
            try:
                tr2, x2, _ = make_workload("stf", dev, rank)
                ms2 = timed(lambda: tr2.step(x2), 5, 10)   # (step 1 records the packing sequence, step 2 replays it: 5 warm-ups)
                ips2 = BATCH_PER_GPU / ms2 * 1e3
                line["stf"] = {"train": {"value": ips2, "unit": "images/s", "ms_per_step": ms2, "steps": 10, "warmup": 5,
                                         "batch": BATCH_PER_GPU, "roofline_step": roofline_of(ips2, 3 * 67.0)},
                               "forward": forward_bench("stf", tr2.params(), x2)}
                del tr2, x2
            except Exception as e:   # the headline line must survive a failure of the extra model
                line["stf"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline and args.model == "cnn":
            line["cpu_baseline"] = cpu_baseline(sd_cpu, min(os.cpu_count() or 1, 16), x.cpu())
        print(json.dumps(line, allow_nan=False))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
