"""Data-parallel training step for the `cnn` codec -- the semantics of the reference loop
(train.py:188-214, configure_optimizers train.py:105-169) on the HIP engine, one process per GPU.

    optimizer.zero_grad(); aux_optimizer.zero_grad()
    out = model(x); loss = lmbda*255^2*mse + bpp; loss.backward()
    clip_grad_norm_(model.parameters(), clip); optimizer.step()          # Adam on non-".quantiles"
    aux = model.aux_loss(); aux.backward(); aux_optimizer.step()         # Adam(1e-4) on ".quantiles"

The reference has no distributed code; data parallelism is added here the MI355X
way: the batch is sharded over ranks, every parameter lives in ONE flat f32 buffer (so gradients are a
single contiguous buffer too), gradient ranges are sum-all-reduced over RCCL/xGMI on a side stream as soon as
the backward tape has passed the ops that produce them (4 large buckets, reverse topological order:
g_s -> slice chains -> hyper path -> g_a), and clip + Adam run as two fused kernels over the flat buffers
with the clip coefficient computed on the device (no host sync anywhere in the step).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from . import _lib as L
from . import engine as E
from ._lib import check, ptr
from .models import SymmetricalTransFormer, SymmetricalTransFormer3, stf6_forward, stf_forward, wacnn_forward

# bucket -> parameter-name prefixes, in the order their gradients complete during backward
# (synthesis transform | slice chains | hyper path | analysis transform); names of the cnn and the stf model
BUCKETS = [("g_s", "syn_layers", "end_conv"),
           ("cc_mean_transforms", "cc_scale_transforms", "lrp_transforms", "cc_mean_transforms2", "cc_scale_transforms2",
            "lrp_transforms2", "mu_Swin", "sigma_Swin", "LRP_Swin"),
           ("h_a", "h_mean_s", "h_scale_s", "entropy_bottleneck"), ("g_a", "patch_embed", "layers")]


class FlatParams:
    """All trainable tensors of a module re-homed into one flat buffer (views keep names/shapes)."""

    def __init__(self, model: torch.nn.Module, device):
        items = [(n, p) for n, p in model.named_parameters()]
        self.main = [(n, p) for n, p in items if not n.endswith(".quantiles")]
        self.aux = [(n, p) for n, p in items if n.endswith(".quantiles")]
        # order the main parameters bucket by bucket so every bucket is one contiguous range
        order: List[Tuple[str, torch.nn.Parameter]] = []
        self.bucket_ranges: List[Tuple[int, int]] = []
        off = 0
        used = set()
        for prefixes in BUCKETS:
            start = off
            for n, p in self.main:
                if n.split(".")[0] in prefixes:
                    order.append((n, p))
                    used.add(n)
                    off += (p.numel() + 3) // 4 * 4
            self.bucket_ranges.append((start, off))
        rest = [(n, p) for n, p in self.main if n not in used]
        if rest:
            start = off
            for n, p in rest:
                order.append((n, p))
                off += (p.numel() + 3) // 4 * 4
            self.bucket_ranges.append((start, off))
        self.main = order
        self.n_main = off
        self.p = torch.zeros(off, dtype=torch.float32, device=device)
        self.g = torch.zeros(off, dtype=torch.float32, device=device)
        self.m = torch.zeros(off, dtype=torch.float32, device=device)
        self.v = torch.zeros(off, dtype=torch.float32, device=device)
        self.views: Dict[str, torch.Tensor] = {}
        self.gviews: Dict[str, torch.Tensor] = {}
        o = 0
        for n, prm in self.main:
            k = prm.numel()
            v = self.p[o:o + k].view(prm.shape)
            v.copy_(prm.data.to(device))
            prm.data = v
            self.views[n] = v
            self.gviews[n] = self.g[o:o + k].view(prm.shape)
            o += (k + 3) // 4 * 4
        na = sum(p.numel() for _, p in self.aux)
        self.ap = torch.zeros(na, dtype=torch.float32, device=device)
        self.ag = torch.zeros(na, dtype=torch.float32, device=device)
        self.am = torch.zeros(na, dtype=torch.float32, device=device)
        self.av = torch.zeros(na, dtype=torch.float32, device=device)
        o = 0
        for n, prm in self.aux:
            k = prm.numel()
            v = self.ap[o:o + k].view(prm.shape)
            v.copy_(prm.data.to(device))
            prm.data = v
            self.views[n] = v
            self.gviews[n] = self.ag[o:o + k].view(prm.shape)
            o += k


class GradReducer:
    """Sum-all-reduce contiguous ranges of the flat gradient buffer on a side stream (RCCL over xGMI when
    the process group is NCCL; gloo works for CPU tests)."""

    def __init__(self, flat_g: torch.Tensor, ranges, group=None):
        self.g, self.ranges, self.group = flat_g, list(ranges), group
        inited = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if inited else 1
        # ICM_FORCE_COLLECTIVES=1 (tests): run the collective path even in a 1-rank group, so that the side-stream
        # async all-reduce / event ordering / Work.wait() sequence executes on a single-GPU box under RCCL
        self.active = self.world > 1 or (inited and os.environ.get("ICM_FORCE_COLLECTIVES", "0") == "1")
        self.cuda = flat_g.is_cuda
        self.stream = torch.cuda.Stream() if (self.cuda and self.active) else None
        self.works = []
        self.launched = 0     # collectives issued (tests)

    def launch(self, bucket: int, after=()):
        """all-reduce bucket `bucket`; after: extra events (recorded on other streams, e.g. the weight-gradient side
        stream) the collective must wait for besides everything issued so far on the current stream"""
        if not self.active:
            return
        a, b = self.ranges[bucket]
        if b <= a:
            return
        t = self.g[a:b]
        self.launched += 1
        if self.cuda and dist.get_backend(self.group) == "gloo":
            # rehearsal path (several ranks sharing one GPU): stage through the host
            for e in after:
                e.synchronize()
            torch.cuda.current_stream().synchronize()
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
            return
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                for e in after:
                    self.stream.wait_event(e)
                self.works.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self.works.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        for w in self.works:
            w.wait()
        self.works = []
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)


class Trainer:
    def __init__(self, model, lr: float = 1e-4, aux_lr: float = 1e-4, lmbda: float = 0.0067,
                 clip_max_norm: float = 1.0, device="cuda:0", group=None, seed: int = 0):
        self.model = model.to(device).train()
        self.device = torch.device(device)
        self.flat = FlatParams(self.model, self.device)
        self.lr, self.aux_lr, self.lmbda, self.clip = lr, aux_lr, lmbda, clip_max_norm
        self.reducer = GradReducer(self.flat.g, self.flat.bucket_ranges, group)
        self.world = self.reducer.world
        self.rank = dist.get_rank(group) if self.reducer.active else 0
        # replicas start from rank 0's parameters (what DistributedDataParallel does at construction) ...
        if self.reducer.active:
            for buf in (self.flat.p, self.flat.ap):
                _broadcast0(buf, group)
        # ... but draw their OWN quantisation noise / DropPath masks: one generator per rank, seeded seed + rank
        # (identical noise on every rank would correlate the shards' gradients)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(int(seed) + self.rank)
        self.step_no = 0
        self.names = [n for n, _ in self.flat.main] + [n for n, _ in self.flat.aux]
        self.scal = torch.zeros(8, dtype=torch.float32, device=self.device)  # [0:5] rd loss, [5] sqnorm, [6] aux
        self.red_ws = torch.zeros(L.REDUCE_WS_FLOATS, dtype=torch.float32, device=self.device)  # fixed-order partial sums
        self.side = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self._side_ws = None
        import os as _os
        self.wg_every = int(_os.environ.get("ICM_WG_EVERY", "8"))
        self.wg_min = int(_os.environ.get("ICM_WG_MIN", "8"))
        if self.wg_every < 0:
            self.side = None
        self._eb = None
        self.hold_chain_wgrads = _os.environ.get("ICM_HOLD_CHAIN_WGRADS", "0") == "1"
        self._pack_seq = None     # weight-packing miss sequence of the first step (windowed batch packing afterwards)
        self.pack_window = int(_os.environ.get("ICM_PACK_WINDOW", "24"))
        self.is_stf = isinstance(model, SymmetricalTransFormer)
        self.is_stf6 = isinstance(model, SymmetricalTransFormer3)
        self.lat_ch = 384 if self.is_stf else 320

    def params(self) -> Dict[str, torch.Tensor]:
        return self.flat.views

    def step_graphed(self, x: torch.Tensor) -> torch.Tensor:
        """step(x) replayed from a captured hipGraph: one host call per iteration instead of ~1 000 (cnn) to ~8 000
        (stf6) launches issued through Python.  The first two calls per input shape run eagerly (they record the
        weight-packing sequence and set kernel attributes); the third captures.  What changes from step to step lives
        in device memory the graph reads: the input batch, the quantisation noise and DropPath scales (drawn outside the
        graph from this rank's generator, in the same order as step() draws them) and Adam's step-dependent scalars
        (icm_adam_step_hyper) -- so a replayed step equals the eager step bit for bit.  Single process only (the RCCL
        reductions are not captured); the side stream of the weight gradients is forked and joined inside the capture."""
        if self.world > 1:
            raise RuntimeError("step_graphed: single-process training only")
        key = tuple(x.shape)
        g = getattr(self, "_graph", None)
        if g is None or g["key"] != key:
            if getattr(self, "_graph_warm", None) != key:
                self._graph_warm, self._graph_warm_n = key, 0
            if self._graph_warm_n < 2:
                self._graph_warm_n += 1
                return self.step(x)
            g = self._capture(x)
        B, _, H, W = x.shape
        dev = self.device
        self.step_no += 1
        g["hyper"][0].copy_(torch.tensor(_adam_hyper(self.lr, self.step_no), dtype=torch.float32))
        g["hyper"][1].copy_(torch.tensor(_adam_hyper(self.aux_lr, self.step_no), dtype=torch.float32))
        # the same draws, in the same order, as step() makes (z noise, y noise, then the DropPath scales)
        g["nz"].copy_(torch.rand(g["nz"].shape, dtype=torch.float32, device=dev, generator=self.gen) - 0.5)
        g["ny"].copy_((torch.rand((B, self.lat_ch, H // 16, W // 16), dtype=torch.float32, device=dev,
                                  generator=self.gen) - 0.5).reshape(g["ny"].shape))
        if g["drops"] is not None:
            fresh = self.model.draw_drops(B, dev, generator=self.gen)
            for k, v in g["drops"].items():
                v.copy_(fresh[k])
        g["x"].copy_(x)
        g["graph"].replay()
        E.bump_weight_generation()
        return self.scal

    def _capture(self, x: torch.Tensor):
        dev = self.device
        B, _, H, W = x.shape
        side = self.side
        if os.environ.get("ICM_GRAPH_SIDE", "1") == "0":
            self.side = None                       # weight gradients on the capture stream
        ny_shape = (B, 24, 64, H // 32, W // 32) if self.is_stf6 else (B, self.lat_ch, H // 16, W // 16)
        g = {"key": tuple(x.shape), "x": x.detach().clone().contiguous(),
             "nz": torch.zeros((B, 192, H // 64, W // 64), dtype=torch.float32, device=dev),
             "ny": torch.zeros(ny_shape, dtype=torch.float32, device=dev),
             "drops": None,
             "hyper": (torch.ones(2, dtype=torch.float32, device=dev), torch.ones(2, dtype=torch.float32, device=dev))}
        if self.is_stf:
            gen_state = self.gen.get_state()       # shapes / keys only: the capture must not consume random numbers
            g["drops"] = {k: v.clone() for k, v in self.model.draw_drops(B, dev, generator=self.gen).items()}
            self.gen.set_state(gen_state)
        step_no = self.step_no
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self.step(g["x"], noise={"z": g["nz"], "y": g["ny"]}, drops=g["drops"], _hyper=g["hyper"])
        self.step_no = step_no                      # nothing ran during capture
        self.side = side
        g["graph"] = graph
        self._graph = g
        return g

    def optimizer_state(self):
        """(optimizer, aux_optimizer) checkpoint entries (train.py:521-523): Adam moments per parameter NAME (independent
        of the flat-buffer layout), the step count and the learning rates"""
        f = self.flat

        def pack(items, m, v, pad):
            out_m, out_v, o = {}, {}, 0
            for n, prm in items:
                k = prm.numel()
                out_m[n] = m[o:o + k].view(prm.shape).detach().cpu().clone()
                out_v[n] = v[o:o + k].view(prm.shape).detach().cpu().clone()
                o += (k + 3) // 4 * 4 if pad else k
            return out_m, out_v
        m, v = pack(f.main, f.m, f.v, True)
        am, av = pack(f.aux, f.am, f.av, False)
        return ({"m": m, "v": v, "step": self.step_no, "lr": self.lr},
                {"m": am, "v": av, "step": self.step_no, "lr": self.aux_lr})

    def load_optimizer_state(self, opt, aux=None) -> None:
        f = self.flat

        def unpack(items, m, v, sd, pad):
            o = 0
            for n, prm in items:
                k = prm.numel()
                if n not in sd["m"] or tuple(sd["m"][n].shape) != tuple(prm.shape):
                    raise ValueError(f"optimizer state: missing or mis-shaped entry for {n}")
                m[o:o + k].view(prm.shape).copy_(sd["m"][n])
                v[o:o + k].view(prm.shape).copy_(sd["v"][n])
                o += (k + 3) // 4 * 4 if pad else k
        unpack(f.main, f.m, f.v, opt, True)
        self.step_no = int(opt["step"])
        self.lr = float(opt.get("lr", self.lr))
        if aux is not None:
            unpack(f.aux, f.am, f.av, aux, False)
            self.aux_lr = float(aux.get("lr", self.aux_lr))

    def step(self, x: torch.Tensor, noise: Optional[dict] = None, drops: Optional[dict] = None,
             _hyper: Optional[tuple] = None) -> torch.Tensor:
        """one training iteration on this rank's shard x [B,3,H,W]; returns the device tensor
        [bpp, mse, loss, sumlog_y, sumlog_z, grad_sqnorm, aux_loss, -] (no host sync).
        _hyper (step_graphed only): device tensors holding Adam's step-dependent scalars for the two optimisers."""
        f, dev = self.flat, self.device
        st = L.stream()
        lib = L.lib()
        if _hyper is None:
            self.step_no += 1
        B, _, H, W = x.shape
        if noise is None:
            nz = torch.rand((B, 192, H // 64, W // 64), dtype=torch.float32, device=dev, generator=self.gen) - 0.5
            ny = torch.rand((B, self.lat_ch, H // 16, W // 16), dtype=torch.float32, device=dev,
                            generator=self.gen) - 0.5
        else:
            nz, ny = noise["z"].to(dev).contiguous(), noise["y"].to(dev).contiguous()
        P = f.views
        tape = E.Tape(need_grad=True)
        tape.side, tape._side_ws, tape.progress_every = self.side, self._side_ws, self.wg_every
        tape.min_jobs = self.wg_min
        tape.stop(x)
        if self.pack_window > 0:
            if self._pack_seq is None:
                tape.pack_log = []
            else:
                tape.use_pack_sequence(self._pack_seq, self.pack_window)
        # Weights are packed just in time (Tape.pack), right before the GEMM that reads them: measured on MI355X,
        # packing all 600 MB up front (icm_pack_weights_batch) is 8 % slower end to end -- the packed fragments
        # fall out of the 256 MB Infinity Cache before they are used and the MFMA waves then wait on HBM.
        # gradients accumulate into the zeroed flat buffer: one 300 MB memset per step replaces the ~150 small
        # first-writer fills (attention tables, LayerNorm parameter sums, ...) that the stf model otherwise issues
        f.g.zero_()
        for n, _ in f.main:   # (the fused EntropyBottleneck backward overwrites its 13 small parameter gradients)
            tape.bind_grad(P[n], f.gviews[n], not n.startswith("entropy_bottleneck."))
        marks = {}
        if self.is_stf:
            if drops is None:   # stochastic depth, drawn per step like timm's DropPath (stf.py:145)
                drops = self.model.draw_drops(B, dev, generator=self.gen)
            drops = {k: v.to(dev, torch.float32).contiguous() for k, v in drops.items()}
            if self.is_stf6:   # zigzag blocks: iid noise, so any layout of the same draw is the same distribution
                ny6 = ny if ny.dim() == 5 else ny.reshape(B, 24, 64, H // 32, W // 32)
                x_hat, y_lik, z_lik = stf6_forward(tape, P, x, nz, ny6, drops, bucket_marks=marks)
            else:
                x_hat, y_lik, z_lik = stf_forward(tape, P, x, nz, ny, drops, bucket_marks=marks)
        else:
            x_hat, y_lik, z_lik = wacnn_forward(tape, P, x, nz, ny, bucket_marks=marks)
        # ---- R-D loss forward + seeds (train.py:53-76)
        check(lib.icm_rd_loss_fwd(ptr(x), ptr(x_hat), x.numel(), ptr(y_lik), y_lik.numel(), ptr(z_lik), z_lik.numel(),
                                  B * H * W, self.lmbda, ptr(self.scal), ptr(self.red_ws), st), "rd_loss_fwd")
        dxh, dly, dlz = E.new(x_hat), E.new(y_lik), E.new(z_lik)
        check(lib.icm_rd_loss_bwd(ptr(x), ptr(x_hat), x.numel(), ptr(y_lik), y_lik.numel(), ptr(z_lik), z_lik.numel(),
                                  B * H * W, self.lmbda, 1.0, ptr(dxh), ptr(dly), ptr(dlz), st), "rd_loss_bwd")
        tape.bind_grad(x_hat, dxh, True)
        tape.bind_grad(y_lik, dly, True)
        tape.bind_grad(z_lik, dlz, True)
        # ---- backward with bucketed all-reduce overlapped (markers fire as the tape unwinds)
        # bucket markers fire as the tape unwinds: flush the queued weight gradients, all-reduce the finished bucket.
        # ICM_HOLD_CHAIN_WGRADS=1 queues the 150 small slice-chain weight gradients between marker 0 (synthesis done)
        # and marker 1 (slice loop done) and issues them in ~15 launches of up to 32 same-geometry problems instead of
        # ~120 launches of 1-4: measured 0.7 % SLOWER on MI355X (same-box A/B, 258.9 vs 260.7 img/s) -- the early,
        # small launches fill CUs next to the latency-bound dgrad chain -- so the default keeps the periodic flush.
        def mark(b):
            if b == 1:
                tape.hold_wgrads = False
            if self.reducer.active:
                # the bucket is complete once the weight gradients queued so far have run on the side stream: the
                # COLLECTIVE waits for them (event on the side stream); the main stream's input-gradient chain does not
                # (round 2 joined the streams here, stalling backward at every bucket boundary)
                E.flush_wgrads(tape)
                after = ()
                if tape.side is not None:
                    ev = torch.cuda.Event()
                    ev.record(tape.side)
                    after = (ev,)
                self.reducer.launch(b, after)
            if b == 0 and self.hold_chain_wgrads:
                tape.hold_wgrads = True
        for name, idx in sorted(marks.items(), key=lambda kv: -kv[1]):
            tape.bw.insert(idx, (lambda b=name: mark(b)))
        tape.backward()
        self._side_ws = tape._side_ws
        if tape.pack_log is not None and self._pack_seq is None:
            self._pack_seq = tape.pack_log
        self.reducer.launch(len(BUCKETS) - 1)   # g_a: last gradients to complete (tape.backward() joined the side stream)
        if len(f.bucket_ranges) > len(BUCKETS):
            self.reducer.launch(len(BUCKETS))
        self.reducer.finish()
        # ---- clip (global L2 norm of the averaged gradients) + Adam, fused over the flat buffers
        gscale = 1.0 / self.world
        sq = self.scal[5:6]
        check(lib.icm_grad_sqnorm(ptr(f.g), f.n_main, ptr(sq), ptr(self.red_ws), st), "grad_sqnorm")
        if _hyper is None:
            check(lib.icm_adam_step(ptr(f.p), ptr(f.g), ptr(f.m), ptr(f.v), f.n_main, self.lr, 0.9, 0.999, 1e-8,
                                    self.step_no, ptr(sq) if self.clip > 0 else 0, float(self.clip), gscale, st), "adam")
        else:
            check(lib.icm_adam_step_hyper(ptr(f.p), ptr(f.g), ptr(f.m), ptr(f.v), f.n_main, 0.9, 0.999, 1e-8,
                                          ptr(_hyper[0]), ptr(sq) if self.clip > 0 else 0, float(self.clip), gscale, st),
                  "adam")
        # ---- aux loss on the UPDATED bottleneck weights, gradient to quantiles only (train.py:212-214)
        prm = E._eb_params(P, "entropy_bottleneck")
        t = math_target()
        check(lib.icm_eb_aux_loss(C.byref(prm), ptr(self.scal[6:7]), ptr(f.ag), 192, t, st), "eb_aux")
        if _hyper is None:
            check(lib.icm_adam_step(ptr(f.ap), ptr(f.ag), ptr(f.am), ptr(f.av), f.ap.numel(), self.aux_lr, 0.9, 0.999,
                                    1e-8, self.step_no, 0, 0.0, 1.0, st), "adam_aux")
        else:
            check(lib.icm_adam_step_hyper(ptr(f.ap), ptr(f.ag), ptr(f.am), ptr(f.av), f.ap.numel(), 0.9, 0.999, 1e-8,
                                          ptr(_hyper[1]), 0, 0.0, 1.0, st), "adam_aux")
        E.bump_weight_generation()   # packed-weight caches keyed before this step are stale now
        return self.scal


def _adam_hyper(lr: float, step: int):
    """the two step-dependent scalars of icm_adam_step, formed in double like torch.optim.Adam forms them"""
    import math
    return [lr / (1.0 - 0.9 ** step), math.sqrt(1.0 - 0.999 ** step)]


def _broadcast0(t: torch.Tensor, group=None):
    """in-place broadcast from rank 0 (RCCL for device tensors; through the host for the gloo rehearsal path)"""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.cpu()
        dist.broadcast(h, src=0, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=0, group=group)


def math_target(tail_mass: float = 1e-9) -> float:
    import math
    return math.log(2 / tail_mass - 1)
