"""Execution engine: a tape of HIP-kernel launches with hand-scheduled backward closures.

Every numeric op of the hot path is a call into libicm_hip.so (``_lib``).  PyTorch supplies device memory
(the caching allocator), streams and the autograd *bridge* (``tape_function``) so that reference-style
training loops (``loss.backward()``) work unchanged; the gradient math itself is the closures below.

Conventions
  * tensors are float32 NCHW device tensors; channel slices of a larger buffer are legal everywhere;
  * ``VT(t, act)`` is a *virtual* tensor: the value is ``act(t)`` but only the pre-activation ``t`` is ever
    stored -- consumers apply ``act`` while staging their operand into LDS, and backward multiplies by
    ``act'(t)`` in the producing dgrad's epilogue;
  * gradient buffers: first writer overwrites, later writers accumulate (``Tape.grad_for_write``).
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib as L
from ._lib import (ACT_GELU, ACT_NONE, ACT_SQUARE, EPI_AXPY2, EPI_GDN, EPI_IGDN, EPI_LRP, EPI_MUL_DGELU, EPI_NONE,
                   EPI_RES, EPI_RES_GELU, EPI_RES_MUL_DGELU, bs, check, ptr)

PEDESTAL = 2.0 ** -36
import os as _os
_SKIP_WGRAD = _os.environ.get("ICM_DEBUG_SKIP_WGRAD", "0") == "1"
PAIR_GATE_BRANCHES = _os.environ.get("ICM_PAIR_GATE_BRANCHES", "1") == "1"


MAX_GROUP = 12   # ICM_MAX_GROUPS of conv_common.h


# Winograd F(2x2, 3x3) for the 3x3 stride-1 pad-1 convolutions (forward and input gradient; csrc/conv_wino.hip):
# ICM_WINO=0 keeps the direct implicit-GEMM form everywhere (same-box A/B, and the reference point of the parity tests)
USE_WINO = _os.environ.get("ICM_WINO", "1") != "0"
# contraction depth below which the direct kernels win even on large launches (ResidualUnit 3x3 96 -> 96 @64x64 x2:
# 206 us direct, 255 us Winograd: six K steps do not amortise the transforms and the 16-point epilogue)
_WINO_MIN_CIN = int(_os.environ.get("ICM_WINO_MIN_CIN", "96"))
# Small launches are latency-bound (prologue gather + transform, LDS round trip of the output transform): below this
# many multiply-adds of the DIRECT form per launch / 9 (N * H * W * Cin * Cout * members) the direct kernels win
# (measured on MI355X, profiles/r03_wino_vs_direct.txt)
_WINO_MIN_WORK = float(_os.environ.get("ICM_WINO_MIN_WORK", "2.0e8"))
# weight gradients are issued in batches of same-geometry problems (flush_wgrads): a single problem counts this many times
_WINO_WG_BATCH = float(_os.environ.get("ICM_WINO_WG_BATCH", "4"))
USE_WINO_WGRAD = _os.environ.get("ICM_WINO_WGRAD", "1") != "0"
# stride-1 Conv2d layers with <= 8 output channels (stf end_conv[2]) as a dense (channel, tap)-row GEMM + col2im
THIN_OUT = _os.environ.get("ICM_THIN_OUT", "1") != "0"
# inference (no gradients): a layer whose consumers all read gelu(y) stores ONLY gelu(y) (the kernels take y2 == y as
# "store the activated value"): the pre-activation is needed by the backward pass alone
EVAL_INPLACE_ACT = _os.environ.get("ICM_EVAL_INPLACE_ACT", "1") != "0"
# input transform of the Winograd convolutions as its own launch (one per distinct input tensor of a launch) + LDS-DMA
# staging in the convolution kernel, instead of gather + transform by the kernel's loader waves per co-block
WINO_PRE = _os.environ.get("ICM_WINO_PRE", "1") != "0"
# (until the gathers of the Winograd weight-gradient kernel were vectorised and interleaved with its multiplies, problems
# with many pixels kept the direct nine-tap kernel: 96 -> 96 @64x64 x6 was 716 us direct, 832 us Winograd; now 595 us)
_WINO_WG_MAX_PIXELS = int(_os.environ.get("ICM_WINO_WG_MAX_PIXELS", "100000"))
_WINO_EPIS = (EPI_NONE, EPI_RES, EPI_RES_GELU, EPI_MUL_DGELU, EPI_RES_MUL_DGELU, EPI_LRP)


# the weight-gradient kernel has its own depth threshold (its contraction runs over the tiles, not the channels)
_WINO_WG_MIN_C = int(_os.environ.get("ICM_WINO_WG_MIN_C", "64"))


def wino_ok(KH, KW, stride, pad, Cin, ps=0, conv_transposed=False, work: float = 1e30, min_c=None) -> bool:
    """should this convolution launch (forward or input gradient) run on the Winograd kernel? (the epilogue kinds of both
    directions of a 3x3 stride-1 layer are all supported: icm_conv_winograd_ok).  work = N * H * W * Cin * Cout * members."""
    return (USE_WINO and KH == 3 and KW == 3 and stride == 1 and pad == 1 and not ps and not conv_transposed and
            Cin >= (_WINO_MIN_CIN if min_c is None else min_c) and work >= _WINO_MIN_WORK and
            L.lib().icm_debug_forced_conv_cfg() < 0)


# In-place HIP updates (icm_adam_step) do not bump torch's tensor version counters: the trainer bumps this generation
# after every optimiser step and every packed-weight cache key carries it, so a caller-owned ``packed_cache`` that
# outlives a training step misses (re-packs) instead of silently serving stale weights.
_weight_gen = [0]


def bump_weight_generation():
    _weight_gen[0] += 1


# ---- per-launch profiling (bench.py's per-shape roofline table): when PROFILE is a list, every MFMA-family launch is
# bracketed by HIP events on its launch stream and recorded as (label, algorithmic FLOP, start, end)
PROFILE: Optional[list] = None


def _prof_begin(stream=None):
    if PROFILE is None:
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record(stream if stream is not None else torch.cuda.current_stream())
    return e


def _prof_end(e0, label: str, flop: float, stream=None):
    if e0 is None:
        return
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record(stream if stream is not None else torch.cuda.current_stream())
    PROFILE.append((label, flop, e0, e1))


def _conv_label(tag, transposed, KH, stride, Cin, Cout, H, W, N, n=1):
    return (f"{tag} {'scatter' if transposed else 'gather'} {KH}x{KH}s{stride} {Cin}->{Cout} @{H}x{W} n{N}"
            + (f" x{n}" if n > 1 else ""))


def new(shape_or_like, device=None):
    """fresh contiguous f32 device tensor (torch caching allocator = device memory plumbing)"""
    if isinstance(shape_or_like, torch.Tensor):
        return torch.empty(shape_or_like.shape, dtype=torch.float32, device=shape_or_like.device)
    return torch.empty(shape_or_like, dtype=torch.float32, device=device)


def zeros(shape, device):
    return torch.zeros(shape, dtype=torch.float32, device=device)


class VT:
    """virtual tensor: value = act(t)"""
    __slots__ = ("t", "act")

    def __init__(self, t: torch.Tensor, act: int = ACT_NONE):
        self.t, self.act = t, act


def _key(t: torch.Tensor):
    return (t.data_ptr(), tuple(t.shape), tuple(t.stride()))


# Materialised activations.  Pre-activations stay the stored tensors (the backward needs them for gelu'), but a
# producer called with act_out=True also stores gelu(y) from its epilogue; consumers of VT(y, ACT_GELU) -- the next
# convolution's operand, its weight gradient's operand, a residual add -- then read that tensor with NO activation:
# GELU is evaluated once per element instead of once per consumer (x co-blocks), and activation-free operands are
# eligible for the LDS-DMA staging of the weight-gradient loaders.  ICM_MATERIALIZE=0 keeps everything virtual.
_MATERIALIZE = _os.environ.get("ICM_MATERIALIZE", "1") != "0"
# launches below this many output pixels are host- / launch-bound (the 8x8 block maps of stf6: 1 024 pixels at B=16):
# the extra tensor per layer costs more on the host than the saved GELU evaluations are worth on the device
_MAT_MIN_PIXELS = int(_os.environ.get("ICM_MAT_MIN_PIXELS", "2048"))


def _operand(tape, xv: "VT"):
    """(tensor, activation) a kernel should read for the virtual tensor xv"""
    if xv.act == ACT_GELU and _MATERIALIZE:
        m = tape.mat.get(_key(xv.t))
        if m is not None:
            return m, ACT_NONE
    return xv.t, xv.act


class Tape:
    def __init__(self, need_grad: bool = True, packed_cache: Optional[dict] = None):
        """packed_cache: a dict kept by the caller across forward calls with CONSTANT weights (inference): the packed
        MFMA-order weight copies are then produced once instead of once per call.  The cache key holds the tensor's
        autograd version and the engine's weight generation (bumped by Trainer.step), so entries packed before an
        optimiser step are never served afterwards."""
        self.need_grad = need_grad
        self.bw: List[Callable[[], None]] = []
        self.grads: Dict[tuple, torch.Tensor] = {}
        self.ready = set()
        self.stopped = set()
        self._packed: Dict[tuple, torch.Tensor] = packed_cache if packed_cache is not None else {}
        self._ws: Optional[torch.Tensor] = None
        self._red: Optional[torch.Tensor] = None
        self.wjobs: list = []
        self.pack_log = None      # when a list: records (w, args) of every cache miss (the trainer's packing plan)
        self.pack_seq = None      # recorded miss sequence of an identical earlier step (windowed batch packing)
        self._seq_pos = None
        self.pack_window = 24
        self.side = None          # optional torch.cuda.Stream for deferred wgrads (overlaps the serial dgrad chain)
        self._side_ws: Optional[torch.Tensor] = None
        self._flushed: list = []  # keeps side-stream operands alive until the streams are joined
        self.progress_every = 0
        self.min_jobs = 8
        self._pending_res: Dict[tuple, tuple] = {}   # key(t) -> (t, dy, gelu): identity-path gradient not yet added
        self.mat: Dict[tuple, torch.Tensor] = {}   # key(pre-activation t) -> gelu(t) stored by t's producer (act_out)
        self.hold_wgrads = False  # True: queue weight gradients without periodic flushes (the slice-chain section
        #                           batches its 150 small problems by geometry at the section end)
        self.st = L.stream()

    # ---- gradient bookkeeping
    def stop(self, t):
        self.stopped.add(_key(t))

    def wants(self, t) -> bool:
        return self.need_grad and _key(t) not in self.stopped

    def bind_grad(self, t, g, initialized: bool):
        k = _key(t)
        self.grads[k] = g
        if initialized:
            self.ready.add(k)
        else:
            self.ready.discard(k)

    # ---- identity-path gradients of residual adds (ResidualUnit: out = conv(...) + x): instead of a separate
    # elementwise pass  d(x) += dy * act'(x)  the term rides on the epilogue of the dgrad that writes d(x) next
    # (the unit's first convolution consumes the same x with the same virtual activation)
    def defer_res_grad(self, t, dy, gelu: bool):
        if not self.wants(t):
            return
        k = _key(t)
        if k in self._pending_res:
            self._flush_res(k)
        self._pending_res[k] = (t, dy, gelu)

    def take_res_grad(self, t, gelu: bool):
        """the pending identity-path gradient of t if its activation matches the caller's dgrad epilogue, else None
        (a mismatching one is applied with a separate pass)"""
        k = _key(t)
        p = self._pending_res.get(k)
        if p is None:
            return None
        if p[2] != gelu:
            self._flush_res(k)
            return None
        del self._pending_res[k]
        return p[1]

    def _flush_res(self, k):
        t, dy, gelu = self._pending_res.pop(k)
        accumulate(self, t, dy, t if gelu else None)

    def grad_for_write(self, t) -> Tuple[torch.Tensor, int]:
        k = _key(t)
        if k in self._pending_res:
            self._flush_res(k)
        g = self.grads.get(k)
        if g is None:
            g = torch.empty(t.shape, dtype=torch.float32, device=t.device)
            self.grads[k] = g
        accum = 1 if k in self.ready else 0
        self.ready.add(k)
        return g, accum

    def grad_of(self, t) -> Optional[torch.Tensor]:
        k = _key(t)
        if k in self._pending_res:
            self._flush_res(k)
        return self.grads.get(k) if k in self.ready else None

    def backward(self):
        n = 0
        for f in reversed(self.bw):
            f()
            n += 1
            if (self.progress_every > 0 and n % self.progress_every == 0 and len(self.wjobs) >= self.min_jobs
                    and not self.hold_wgrads):
                flush_wgrads(self)
        self.bw = []
        for k in list(self._pending_res):
            self._flush_res(k)
        flush_wgrads(self)
        self.join_side()

    def join_side(self):
        """main stream waits for everything issued on the side stream"""
        if self.side is not None and self._flushed:
            torch.cuda.current_stream().wait_stream(self.side)
            self._flushed = []

    # ---- helpers
    def red_ws(self, nfloats: int, device) -> torch.Tensor:
        """scratch for the fixed-order (atomic-free) reductions of the main stream (bias / LayerNorm / table grads)"""
        if self._red is None or self._red.numel() < nfloats or self._red.device != device:
            self._red = torch.empty(max(nfloats, 1 << 16), dtype=torch.float32, device=device)
        return self._red

    def workspace(self, nfloats: int, device) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nfloats:
            self._ws = torch.empty(max(nfloats, 1 << 22), dtype=torch.float32, device=device)
        return self._ws

    def pack(self, w, M, K, KH, KW, src_out_major, transposed, stride, pad, nonneg=0, bound=0.0, ped=0.0, wino=0,
             log: bool = True):
        """MFMA-fragment-order copy of a weight for this step (one packing job; see pack_entry).
        wino: 1 / 2 = Winograd-domain weights (forward / input-gradient orientation) of a 3x3 stride-1 kernel.
        log=False: w is a per-step temporary (conv2d_thin_out): it is packed on demand only, never ahead of time from the
        sequence a previous step recorded."""
        args = (M, K, KH, KW, src_out_major, transposed, stride, pad, nonneg, float(bound), float(ped), 0, 0, 0, 0, wino)
        floats = L.lib().icm_packed_weight_floats(M, K, 4, 4) if wino else L.lib().icm_packed_weight_floats(M, K, KH, KW)
        return self.pack_entry(((w.data_ptr(),), (args,)), floats, [(w, args, 0)], log=log)

    def pack_cat(self, members, M, K, KH, KW, src_out_major, transposed, stride, pad, axis: str, wino=0):
        """ONE packed buffer holding sub-matrices of several weights laid end to end along a GEMM axis, so that a single
        launch contracts what the reference runs as separate first layers of the slice chains (cnn.py:89-127).
        members: [(w, col_off)]: member m contributes columns [col_off, col_off + count) of the INNER matrix index of
        its canonical weight (input channels of a Conv2d weight), count = K for the forward orientation
        (src_out_major=1), M for the input-gradient orientation (src_out_major=0).
        axis "M": the members' M rows are concatenated (each M rows, a multiple of 32): forward, outputs side by side.
        axis "K": the members' K channels are concatenated (each K, a multiple of 8): one contraction over all of them.
        M, K are PER MEMBER."""
        n = len(members)
        lib = L.lib()
        per = lib.icm_packed_weight_floats(M, K, 4, 4) if wino else lib.icm_packed_weight_floats(M, K, KH, KW)
        if axis == "M":
            if M % 32:
                raise ValueError("pack_cat: M-concatenation needs members of a multiple of 32 rows")
            ncot = M // 32
        elif K % 8:
            raise ValueError("pack_cat: K-concatenation needs members of a multiple of 8 channels")
        jobs, ids, argl = [], [], []
        for m, (w, col) in enumerate(members):
            ld = w.shape[1]   # canonical Conv2d weight [Cout][Cin_total][KH][KW]: the inner index is the input channel
            if axis == "M":
                a = (M, K, KH, KW, src_out_major, transposed, stride, pad, 0, 0.0, 0.0, ld, col, n * ncot, m * ncot, wino)
                off = 0
            else:
                a = (M, K, KH, KW, src_out_major, transposed, stride, pad, 0, 0.0, 0.0, ld, col, 0, 0, wino)
                off = m * per
            jobs.append((w, a, off))
            ids.append(w.data_ptr())
            argl.append(a)
        return self.pack_entry((tuple(ids), tuple(argl), axis), n * per, jobs)

    def pack_entry(self, ident, floats, jobs, log: bool = True):
        """ident: hashable, stable across steps (weight addresses + job arguments); jobs: [(w, args16, float offset)].
        With ``pack_seq`` (the miss sequence recorded on an earlier, identical step) a miss packs a short WINDOW of
        upcoming entries in one launch: the ~830 tiny per-layer pack launches of a training step become ~40, while every
        packed weight is still produced just before its consumer (packing everything up front pushes it out of the
        Infinity Cache: DESIGN.md 5)."""
        vers = tuple(w._version for w, _, _ in jobs)
        k = (ident, vers, _weight_gen[0])
        wp = self._packed.get(k)
        if wp is not None:
            return wp
        if self.pack_log is not None and log:
            self.pack_log.append((ident, floats, jobs))
        todo = [(k, floats, jobs)]
        seq = self.pack_seq if log else None
        if seq is not None:
            i = self._seq_pos.get(ident) if self._seq_pos is not None else None
            if i is not None:
                todo = []
                for (id2, fl2, jobs2) in seq[i:i + self.pack_window]:
                    k2 = (id2, tuple(w._version for w, _, _ in jobs2), _weight_gen[0])
                    if k2 not in self._packed:
                        todo.append((k2, fl2, jobs2))
        dev = jobs[0][0].device
        flat = []
        for k2, fl2, jobs2 in todo:
            buf = torch.empty(fl2, dtype=torch.float32, device=dev)
            self._packed[k2] = buf
            for w2, a2, off in jobs2:
                flat.append((w2, a2, ptr(buf) + 4 * off))
        arr = (L.PackJob * len(flat))()
        for j, (w2, a2, wpp) in zip(arr, flat):
            j.w, j.wp, j.Cout, j.Cin, j.KH, j.KW = ptr(w2), wpp, a2[0], a2[1], a2[2], a2[3]
            (j.src_out_major, j.transposed, j.stride, j.pad, j.nonneg, j.bound, j.pedestal, j.src_ld, j.src_off,
             j.dst_ncot, j.dst_cot_off, j.wino) = a2[4:16]
        e0 = _prof_begin()
        check(L.lib().icm_pack_weights_batch(arr, len(flat), self.st), "pack_weights_batch")
        _prof_end(e0, "pack weights (window)" if len(todo) > 1 else "pack weights (single)", 0.0)
        return self._packed[k]

    def use_pack_sequence(self, seq, window: int = 24):
        """seq: list of (ident, floats, jobs) in miss order (a ``pack_log``)"""
        self.pack_seq, self.pack_window = seq, window
        self._seq_pos = {}
        for i, (ident, _, _) in enumerate(seq):
            self._seq_pos.setdefault(ident, i)


# ------------------------------------------------------------------------------------------------ raw launches

def _wino_pretransform(tape, arr, xs):
    """fill arr[i].xv: B^T d B of every distinct input tensor of a Winograd launch, computed by ONE transform launch"""
    lib = L.lib()
    nfl = lib.icm_wino_transform_floats(C.byref(arr[0]))
    if nfl <= 0:
        raise ValueError("icm winograd transform: unsupported geometry")
    bufs, firsts = {}, []
    for i, x in enumerate(xs):
        k = _key(x)
        if k not in bufs:
            bufs[k] = torch.empty(nfl, dtype=torch.float32, device=x.device)
            firsts.append(i)
        arr[i].xv = ptr(bufs[k])
    for j0 in range(0, len(firsts), MAX_GROUP):
        part = firsts[j0:j0 + MAX_GROUP]
        tarr = (L.ConvArgs * len(part))(*[arr[i] for i in part])
        check(lib.icm_wino_transform(tarr, len(part), tape.st), "wino_transform")
    return bufs

def conv_launch(tape, x, wp, bias, y, *, Cin, Cout, KH, KW, stride, pad, transposed, OH, OW, pro_act=ACT_NONE,
                epi=EPI_NONE, res=None, aux=None, aux2=None, y2=None, accum=0, ps=0, tag="fwd", seg=None, algo=0):
    """seg = (run length, gap): blocked input-channel map (icm_conv_args.x_seg_len / x_seg_gap)"""
    a = L.ConvArgs()
    N, _, H, W = x.shape
    e0 = _prof_begin()
    a.x, a.x_bs, a.N, a.Cin, a.H, a.W = ptr(x), bs(x), N, Cin, H, W
    a.wp, a.bias = ptr(wp), ptr(bias)
    a.y, a.y_bs, a.Cout, a.OH, a.OW = ptr(y), bs(y), Cout, OH, OW
    a.KH, a.KW, a.stride, a.pad = KH, KW, stride, pad
    a.transposed, a.pro_act, a.epi = int(transposed), pro_act, epi
    a.res, a.res_bs = ptr(res), bs(res)
    a.aux, a.aux_bs = ptr(aux), bs(aux)
    a.aux2, a.aux2_bs = ptr(aux2), bs(aux2)
    a.y2, a.y2_bs = ptr(y2), bs(y2)
    a.accum, a.pixel_shuffle = accum, ps
    if seg is not None:
        a.x_seg_len, a.x_seg_gap = seg
    a.algo = algo
    keep = None
    if algo and WINO_PRE:
        arr1 = (L.ConvArgs * 1)(a)
        keep = _wino_pretransform(tape, arr1, [x])
        a = arr1[0]
    check(L.lib().icm_conv_run(C.byref(a), tape.st), "conv_run")
    if e0 is not None:
        px = H * W if transposed else OH * OW
        _prof_end(e0, _conv_label(tag + ("/wino" if algo else ""), transposed, KH, stride, Cin, Cout, H, W, N),
                  2.0 * N * Cin * Cout * KH * KW * px)


def wgrad_launch(tape, gs, gb, dw, *, Ca, Cb, KH, KW, stride, pad, act_s=ACT_NONE, act_b=ACT_NONE, accum=0,
                 dbias=None, accum_bias=0):
    a = L.WgradArgs()
    N, _, OH, OW = gs.shape
    _, _, H, W = gb.shape
    a.gs, a.gs_bs, a.Ca, a.OH, a.OW, a.act_s = ptr(gs), bs(gs), Ca, OH, OW, act_s
    a.gb, a.gb_bs, a.Cb, a.H, a.W, a.act_b = ptr(gb), bs(gb), Cb, H, W, act_b
    a.N, a.KH, a.KW, a.stride, a.pad = N, KH, KW, stride, pad
    a.dw, a.accum = ptr(dw), accum
    a.dbias, a.accum_bias = ptr(dbias), accum_bias
    a.ws = 0
    n = L.lib().icm_wgrad_workspace_floats(C.byref(a))
    if n < 0:
        raise ValueError("icm wgrad: invalid geometry")
    ws = tape.workspace(n, gs.device)
    a.ws, a.ws_floats = ptr(ws), n
    e0 = _prof_begin()
    check(L.lib().icm_conv_wgrad(C.byref(a), tape.st), "conv_wgrad")
    _prof_end(e0, f"wgrad {KH}x{KH}s{stride} {Cb}->{Ca} @{H}x{W} n{N}", 2.0 * N * Ca * Cb * KH * KW * OH * OW)


def wgrad_defer(tape, gs, gb, dw, *, Ca, Cb, KH, KW, stride, pad, act_s=ACT_NONE, act_b=ACT_NONE, accum=0,
                dbias=None, accum_bias=0, dw_ld=0, algo=None):
    """Queue a weight-gradient problem.  Nothing but the optimiser consumes a weight gradient, so problems are
    collected while the tape unwinds and issued in batches of identical geometry (flush_wgrads).
    dw_ld > 0: dw points at a column block of a [Ca][dw_ld][KH][KW] tensor (icm_wgrad_args.dw_ld)."""
    N, _, OH, OW = gs.shape
    _, _, H, W = gb.shape
    if algo is None:   # Winograd form for 3x3 stride-1 problems (wgrad_wino.hip); the work of a batch is what counts, so the
        #                threshold is applied per problem with a typical batch factor folded into ICM_WINO_MIN_WORK_WG
        algo = 1 if (USE_WINO_WGRAD and act_s == ACT_NONE and act_b == ACT_NONE and   # (materialised operands only)
                     wino_ok(KH, KW, stride, pad, min(Ca, Cb), work=float(N) * OH * OW * Ca * Cb * _WINO_WG_BATCH,
                             min_c=_WINO_WG_MIN_C)
                     and N * OH * OW < _WINO_WG_MAX_PIXELS) else 0
    key = (Ca, Cb, KH, KW, stride, pad, act_s, act_b, N, OH, OW, H, W, bs(gs), bs(gb), dbias is not None, algo)
    tape.wjobs.append((key, gs, gb, dw, accum, dbias, accum_bias, dw_ld))


def flush_wgrads(tape):
    """Issue the queued weight-gradient problems, batched by geometry.  With ``tape.side`` set they go to that
    stream (after an event on the main stream): nothing on the main stream consumes them before the optimiser,
    so they fill the CUs the latency-bound dgrad chain leaves idle."""
    if not tape.wjobs:
        return
    if _SKIP_WGRAD:   # measurement only (ICM_DEBUG_SKIP_WGRAD=1): time the main stream without weight gradients
        tape.wjobs = []
        return
    groups: Dict[tuple, list] = {}
    for job in tape.wjobs:
        groups.setdefault(job[0], []).append(job)
    jobs_all = tape.wjobs
    tape.wjobs = []
    lib = L.lib()
    st = tape.st
    side = tape.side
    if side is not None:
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        side.wait_event(ev)
        st = side.cuda_stream
        tape._flushed.append(jobs_all)
    for key, jobs in groups.items():
        Ca, Cb, KH, KW, stride, pad, act_s, act_b, N, OH, OW, H, W, gsb, gbb, _, algo = key
        for i0 in range(0, len(jobs), 32):
            chunk = jobs[i0:i0 + 32]
            arr = (L.WgradArgs * len(chunk))()
            for a, (_, gs, gb, dw, accum, dbias, accum_bias, dw_ld) in zip(arr, chunk):
                a.gs, a.gs_bs, a.Ca, a.OH, a.OW, a.act_s = ptr(gs), gsb, Ca, OH, OW, act_s
                a.gb, a.gb_bs, a.Cb, a.H, a.W, a.act_b = ptr(gb), gbb, Cb, H, W, act_b
                a.N, a.KH, a.KW, a.stride, a.pad = N, KH, KW, stride, pad
                a.dw, a.accum = ptr(dw), accum
                a.dbias, a.accum_bias = ptr(dbias), accum_bias
                a.dw_ld = dw_ld
                a.algo = algo
            n = lib.icm_wgrad_workspace_floats_grouped(C.byref(arr[0]), len(chunk))
            if n < 0:
                raise ValueError("icm wgrad: invalid geometry")
            n = (n + 63) // 64 * 64
            if side is not None:
                if tape._side_ws is None or tape._side_ws.numel() < n * len(chunk):
                    with torch.cuda.stream(side):
                        tape._side_ws = torch.empty(max(n * len(chunk), 1 << 24), dtype=torch.float32,
                                                    device=chunk[0][1].device)
                ws = tape._side_ws
            else:
                ws = tape.workspace(n * len(chunk), chunk[0][1].device)
            for j, a in enumerate(arr):
                a.ws, a.ws_floats = ptr(ws) + 4 * n * j, n
            e0 = _prof_begin(side)
            check(lib.icm_conv_wgrad_grouped(arr, len(chunk), st), "conv_wgrad_grouped")
            _prof_end(e0, f"wgrad{'/wino' if algo else ''} {KH}x{KH}s{stride} {Cb}->{Ca} @{H}x{W} n{N}" + (f" x{len(chunk)}" if len(chunk) > 1 else ""),
                      2.0 * len(chunk) * N * Ca * Cb * KH * KW * OH * OW, side)


def accumulate(tape, dst_t, src, mul_dgelu_of=None):
    """grad(dst_t) (+)= src * gelu'(mul_dgelu_of)"""
    if not tape.wants(dst_t):
        return
    g, acc = tape.grad_for_write(dst_t)
    if g.is_contiguous() and src.is_contiguous() and (mul_dgelu_of is None or mul_dgelu_of.is_contiguous()):
        check(L.lib().icm_add_grad(ptr(src), ptr(mul_dgelu_of), ptr(g), src.numel(), acc, tape.st), "add_grad")
    else:
        assert mul_dgelu_of is None
        N, Cc = src.shape[0], src.shape[1]
        HW = src.shape[2] * src.shape[3]
        check(L.lib().icm_copy_strided(ptr(src), bs(src), ptr(g), bs(g), N, Cc, HW, acc, tape.st), "copy_strided")


def channel_sum(tape, x, out, accum):
    """out[c] (+)= sum_{n,p} x[n,c,p] with fixed-order split partials (bias gradients, GDN d_beta)"""
    N, Cc = x.shape[0], x.shape[1]
    HW = x.shape[2] * x.shape[3]
    ws = tape.red_ws(32 * Cc, x.device)
    check(L.lib().icm_channel_sum(ptr(x), bs(x), N, Cc, HW, ptr(out), accum, ptr(ws), ws.numel(), tape.st), "channel_sum")


def copy_into(tape, src, dst, accum=0):
    N, Cc = src.shape[0], src.shape[1]
    HW = src.shape[2] * src.shape[3]
    check(L.lib().icm_copy_strided(ptr(src), bs(src), ptr(dst), bs(dst), N, Cc, HW, accum, tape.st), "copy_strided")


# ------------------------------------------------------------------------------------------------ conv family
def conv2d(tape: Tape, xv: VT, w, b, *, stride=1, pad=0, transposed=False, output_padding=0, res: Optional[VT] = None,
           out=None, pixel_shuffle=0, lrp_aux=None, w_as: Optional[Tuple[int, int]] = None, temp_weight: bool = False,
           act_out: bool = False) -> torch.Tensor:
    """nn.Conv2d / nn.ConvTranspose2d / nn.Linear(on NCHW) forward with fused neighbours.  Returns the
    pre-activation output tensor (or y_hat for the LRP epilogue).  w_as=(d0, d1): use the (contiguous) weight as a
    [d0, d1, 1, 1] matrix (thin-channel layers run as 1x1 GEMMs over (channel, tap) pairs); gradients still land in
    the weight's own buffer.  act_out: the consumers of the result apply GELU -- store gelu(y) next to y (tape.mat).
    temp_weight: w is a per-step temporary derived from a parameter (conv2d_thin_out): its packed copies are never
    produced ahead of time from a recorded sequence and its gradient is computed at once on the main stream (the caller
    carries it back into the parameter's layout right behind this layer's backward)."""
    x, act = xv.t, xv.act
    xf, actf = _operand(tape, xv)       # what the forward kernel and the weight gradient read
    N, Cin, H, W = x.shape
    w4 = w if w.dim() == 4 else w.view(w.shape[0], w.shape[1], 1, 1)
    if w_as is not None:
        w4 = w.view(w_as[0], w_as[1], 1, 1)
    wino = 0
    if not transposed:
        Cout, ci, KH, KW = w4.shape
        OH, OW = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
        wino = 1 if wino_ok(KH, KW, stride, pad, Cin, pixel_shuffle, work=float(N) * OH * OW * Cin * Cout) and w_as is None else 0
        wp = tape.pack(w4, Cout, Cin, KH, KW, 1, 0, stride, pad, wino=wino, log=not temp_weight)
    else:
        ci, Cout, KH, KW = w4.shape
        OH = (H - 1) * stride - 2 * pad + KH + output_padding
        OW = (W - 1) * stride - 2 * pad + KW + output_padding
        wp = tape.pack(w4, Cout, Cin, KH, KW, 0, 1, stride, pad, log=not temp_weight)
    if ci != Cin:
        raise ValueError(f"conv2d: weight expects {ci} input channels, got {Cin}")
    if pixel_shuffle == 2:
        oshape = (N, Cout // 4, OH * 2, OW * 2)
    else:
        oshape = (N, Cout, OH, OW)
    y = out if out is not None else torch.empty(oshape, dtype=torch.float32, device=x.device)
    assert tuple(y.shape) == oshape, (tuple(y.shape), oshape)
    epi, resv, aux, y2 = EPI_NONE, None, None, None
    if res is not None:
        assert res.act in (ACT_NONE, ACT_GELU)
        resv, ract = _operand(tape, res)
        epi = EPI_RES_GELU if ract == ACT_GELU else EPI_RES
    if lrp_aux is not None:
        assert res is None
        epi, aux = EPI_LRP, lrp_aux
        y2 = torch.empty(oshape, dtype=torch.float32, device=x.device)
    elif act_out and _MATERIALIZE and N * OH * OW >= _MAT_MIN_PIXELS:
        if EVAL_INPLACE_ACT and not tape.need_grad and out is None:
            y2 = y     # inference: y holds gelu(pre-activation); VT(y, GELU) consumers read it as is (tape.mat)
        else:
            y2 = torch.empty(oshape, dtype=torch.float32, device=x.device)
        tape.mat[_key(y)] = y2
    conv_launch(tape, xf, wp, b, y, Cin=Cin, Cout=Cout, KH=KH, KW=KW, stride=stride, pad=pad, transposed=transposed,
                OH=OH, OW=OW, pro_act=actf, epi=epi, res=resv, aux=aux, y2=y2, ps=pixel_shuffle, algo=wino)
    if not tape.need_grad:
        return y

    def bwd():
        dy = tape.grad_of(y)
        if dy is None:
            return
        if lrp_aux is not None:
            accumulate(tape, lrp_aux, dy)
            dpre = torch.empty(oshape, dtype=torch.float32, device=x.device)
            check(L.lib().icm_lrp_bwd(ptr(dy), bs(dy), ptr(y2), bs(y2), ptr(dpre), bs(dpre), N, Cout, OH * OW, tape.st),
                  "lrp_bwd")
            dy = dpre
        if pixel_shuffle == 2:
            du = torch.empty((N, Cout, OH, OW), dtype=torch.float32, device=x.device)
            assert dy.is_contiguous()
            check(L.lib().icm_pixel_unshuffle2(ptr(dy), ptr(du), N, Cout // 4, OH, OW, tape.st), "pixel_unshuffle2")
            dy = du
        if res is not None:
            tape.defer_res_grad(res.t, dy, res.act == ACT_GELU)
        want_b = b is not None and tape.wants(b)
        fuse_b = want_b and not transposed and tape.wants(w)   # bias grad rides on the wgrad loaders
        if want_b and not fuse_b:
            gb_, acc = tape.grad_for_write(b)
            channel_sum(tape, dy, gb_, acc)
        if tape.wants(w):
            gw, acc = tape.grad_for_write(w)
            wg = wgrad_launch if temp_weight else wgrad_defer
            if not transposed:
                gb_, accb = tape.grad_for_write(b) if fuse_b else (None, 0)
                wg(tape, dy, xf, gw, Ca=Cout, Cb=Cin, KH=KH, KW=KW, stride=stride, pad=pad, act_b=actf,
                   accum=acc, dbias=gb_, accum_bias=accb)
            else:
                wg(tape, xf, dy, gw, Ca=Cin, Cb=Cout, KH=KH, KW=KW, stride=stride, pad=pad, act_s=actf,
                   accum=acc)
        if tape.wants(x):
            rg = tape.take_res_grad(x, act == ACT_GELU) if act in (ACT_GELU, ACT_NONE) else None
            gx, acc = tape.grad_for_write(x)
            if act == ACT_GELU:
                epi_b, aux_b = (EPI_RES_MUL_DGELU if rg is not None else EPI_MUL_DGELU), x
            elif act == ACT_NONE:
                epi_b, aux_b = (EPI_RES if rg is not None else EPI_NONE), None
            else:
                raise NotImplementedError("dgrad through this virtual activation")
            if not transposed:   # conv dgrad = scatter with W ([K=Cout][M=Cin])
                wb = 2 if (wino_ok(KH, KW, stride, pad, Cout, pixel_shuffle, work=float(N) * H * W * Cin * Cout)
                           and w_as is None) else 0
                wpb = tape.pack(w4, Cin, Cout, KH, KW, 0, 1, stride, pad, wino=wb, log=not temp_weight)
                conv_launch(tape, dy, wpb, None, gx, Cin=Cout, Cout=Cin, KH=KH, KW=KW, stride=stride, pad=pad,
                            transposed=1, OH=H, OW=W, epi=epi_b, aux=aux_b, res=rg, accum=acc, tag="dgrad",
                            algo=1 if wb else 0)
            else:                # convT dgrad = gather with Wt ([M=Cin][K=Cout])
                wpb = tape.pack(w4, Cin, Cout, KH, KW, 1, 0, stride, pad, log=not temp_weight)
                conv_launch(tape, dy, wpb, None, gx, Cin=Cout, Cout=Cin, KH=KH, KW=KW, stride=stride, pad=pad,
                            transposed=0, OH=H, OW=W, epi=epi_b, aux=aux_b, res=rg, accum=acc, tag="dgrad")

    tape.bw.append(bwd)
    return y


def conv2d_thin_in(tape: Tape, x, w, b, *, stride, pad) -> torch.Tensor:
    """Conv2d with very few input channels (g_a.0: 3 -> 192, 5x5 s2; cnn.py:32) as im2col + 1x1 GEMM over the
    Cin*K*K (channel, tap) pairs: the implicit-GEMM kernel would pad 3 channels to an 8-channel K-chunk per tap and
    its wgrad a 3-wide N tile to 32."""
    N, Cin, H, W = x.shape
    Cout, ci, K, K2 = w.shape
    if ci != Cin or K != K2:
        raise ValueError("conv2d_thin_in: weight / input mismatch")
    OH, OW = (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1
    xc = x if x.is_contiguous() else x.contiguous()
    cols = torch.empty((N, Cin * K * K, OH, OW), dtype=torch.float32, device=x.device)
    check(L.lib().icm_im2col(ptr(xc), ptr(cols), N, Cin, H, W, K, stride, pad, tape.st), "im2col")
    if tape.need_grad:
        if tape.wants(x):
            def bwd():
                g = tape.grad_of(cols)
                if g is None:
                    return
                dx, acc = tape.grad_for_write(x)
                assert dx.is_contiguous() and g.is_contiguous()
                check(L.lib().icm_col2im(ptr(g), 0, ptr(dx), N, Cin, H, W, K, stride, pad, acc, tape.st), "col2im")
            tape.bw.append(bwd)
        else:
            tape.stop(cols)
    return conv2d(tape, VT(cols), w, b, w_as=(Cout, Cin * K * K))


def convT2d_thin_out(tape: Tape, xv: VT, w, b, *, stride, pad, output_padding, temp_weight: bool = False) -> torch.Tensor:
    """ConvTranspose2d with very few output channels (g_s.8: 192 -> 3, 5x5 s2; cnn.py:51) as a 1x1 GEMM producing the
    Cout*K*K per-tap contributions of every input pixel, then one col2im pass (+ bias)."""
    x = xv.t
    N, Cin, H, W = x.shape
    ci, Cout, K, K2 = w.shape
    if ci != Cin or K != K2:
        raise ValueError("convT2d_thin_out: weight / input mismatch")
    OH = (H - 1) * stride - 2 * pad + K + output_padding
    OW = (W - 1) * stride - 2 * pad + K + output_padding
    tmp = conv2d(tape, xv, w, None, transposed=True, stride=1, pad=0, w_as=(Cin, Cout * K * K), temp_weight=temp_weight)
    out = torch.empty((N, Cout, OH, OW), dtype=torch.float32, device=x.device)
    check(L.lib().icm_col2im(ptr(tmp), ptr(b), ptr(out), N, Cout, OH, OW, K, stride, pad, 0, tape.st), "col2im")
    if tape.need_grad:
        def bwd():
            dy = tape.grad_of(out)
            if dy is None:
                return
            dy = dy if dy.is_contiguous() else dy.contiguous()
            if b is not None and tape.wants(b):
                gb_, acc = tape.grad_for_write(b)
                channel_sum(tape, dy, gb_, acc)
            dt, acc = tape.grad_for_write(tmp)
            assert acc == 0
            check(L.lib().icm_im2col(ptr(dy), ptr(dt), N, Cout, OH, OW, K, stride, pad, tape.st), "im2col")
        tape.bw.append(bwd)
    return out


def conv2d_thin_out(tape: Tape, xv: VT, w, b, *, pad) -> torch.Tensor:
    """Stride-1 Conv2d with very few output channels (stf.py:401-404, end_conv[2]: 48 -> 3, 3x3 on the full-resolution
    map).  The implicit-GEMM kernels pad the 3 output channels to a 32-row tile per tap (forward 8 TF, weight gradient
    3 TF: 0.86 ms for 5 GFLOP); as the ConvTranspose2d of the same map -- W'[ci][co][ky][kx] = W[co][ci][K-1-ky][K-1-kx],
    pad' = K-1-pad -- it takes the thin-output path: one dense 1x1 GEMM with Cout*K*K = 27 rows + col2im.  W' is a
    per-step temporary (icm_permute_flip); its gradient is computed at once and carried back into W's layout by the same
    permutation, so the parameter's gradient is complete when this layer's backward returns (bucketed all-reduce)."""
    Cout, Cin, K, K2 = w.shape
    if K != K2 or not (0 <= pad <= K - 1):
        raise ValueError("conv2d_thin_out: square kernel and 0 <= pad <= K-1 expected")
    # eval with a packed cache: the permuted copy lives as long as the packed weights.  With gradients every call gets
    # its own copy (its own gradient buffer: a weight shared by two layers accumulates once per layer).
    ck = ("thin_wt", w.data_ptr(), w._version, _weight_gen[0])
    wt = None if tape.need_grad else tape._packed.get(ck)
    if wt is None:
        wt = torch.empty((Cin, Cout, K, K), dtype=torch.float32, device=w.device)
        wc = w if w.is_contiguous() else w.contiguous()
        check(L.lib().icm_permute_flip(ptr(wc), ptr(wt), Cout, Cin, K * K, 0, tape.st), "permute_flip")
        if not tape.need_grad:
            tape._packed[ck] = wt
    if tape.need_grad:
        if tape.wants(w):
            def bwd_w():                # registered first = runs after the backward of the layer below
                g = tape.grad_of(wt)
                if g is None:
                    return
                gw, acc = tape.grad_for_write(w)
                assert gw.is_contiguous() and g.is_contiguous()
                check(L.lib().icm_permute_flip(ptr(g), ptr(gw), Cin, Cout, K * K, acc, tape.st), "permute_flip")
            tape.bw.append(bwd_w)
        else:
            tape.stop(wt)
    return convT2d_thin_out(tape, xv, wt, b, stride=1, pad=K - 1 - pad, output_padding=0, temp_weight=True)


def conv_launch_grouped(tape, xs, wps, biases, ys, *, Cin, Cout, KH, KW, stride, pad, transposed, OH, OW,
                        pro_act=ACT_NONE, epi=EPI_NONE, auxs=None, y2s=None, ress=None, accum=0, ps=0, tag="fwd", algo=0):
    n = len(xs)
    arr = (L.ConvArgs * n)()
    e0 = _prof_begin()
    for i, a in enumerate(arr):
        x, y = xs[i], ys[i]
        N, _, H, W = x.shape
        a.x, a.x_bs, a.N, a.Cin, a.H, a.W = ptr(x), bs(x), N, Cin, H, W
        a.wp, a.bias = ptr(wps[i]), ptr(biases[i]) if biases is not None else 0
        a.y, a.y_bs, a.Cout, a.OH, a.OW = ptr(y), bs(y), Cout, OH, OW
        a.KH, a.KW, a.stride, a.pad = KH, KW, stride, pad
        a.transposed, a.pro_act, a.epi = int(transposed), pro_act, epi
        aux = auxs[i] if auxs is not None else None
        a.aux, a.aux_bs = ptr(aux), bs(aux)
        y2 = y2s[i] if y2s is not None else None
        a.y2, a.y2_bs = ptr(y2), bs(y2)
        res = ress[i] if ress is not None else None
        a.res, a.res_bs = ptr(res), bs(res)
        a.accum, a.pixel_shuffle = accum, ps
        a.algo = algo
        if i and (bs(x) != bs(xs[0]) or bs(y) != bs(ys[0]) or bs(aux) != bs(auxs[0] if auxs else None)
                  or bs(y2) != bs(y2s[0] if y2s else None) or bs(res) != bs(ress[0] if ress else None)):
            raise ValueError("grouped conv: members must share strides")
    keep = _wino_pretransform(tape, arr, xs) if (algo and WINO_PRE) else None
    check(L.lib().icm_conv_run_grouped(arr, n, tape.st), "conv_run_grouped")
    if e0 is not None:
        N, _, H, W = xs[0].shape
        px = H * W if transposed else OH * OW
        _prof_end(e0, _conv_label(tag + ("/wino" if algo else ""), transposed, KH, stride, Cin, Cout, H, W, N, n),
                  2.0 * n * N * Cin * Cout * KH * KW * px)




def conv2d_group(tape: Tape, xvs, ws, bs_, *, pad=1, outs=None, lrp_auxs=None, ress=None, pixel_shuffle=0,
                 act_out: bool = False):
    """The same stride-1 convolution shape applied to several independent (input, weight) pairs in ONE launch:
    cc_mean_transforms[i] || cc_scale_transforms[i] (cnn.py:164-168), and -- because the support of slice i is
    y_hat_slices[:max_support] (cnn.py:161), i.e. the FIRST five slices -- all chains of the slices >= max_support at
    once.  Members may share an input tensor (their input gradients are then summed).  outs / lrp_auxs: write into
    the given tensors with the LRP tail (cnn.py:175-178) fused.  ress: per-member residual VT (identity or virtual
    GELU) added in the epilogue (ResidualUnit tails of the two gate branches, layers.py:66-71).  Returns the outputs."""
    n = len(xvs)
    if n > MAX_GROUP:
        raise ValueError("conv2d_group: too many members")
    x0, act = xvs[0].t, xvs[0].act
    N, Cin, H, W = x0.shape
    Cout, ci, KH, KW = ws[0].shape
    assert all(v.act == act and v.t.shape == x0.shape for v in xvs) and all(w.shape == ws[0].shape for w in ws)
    if ci != Cin:
        raise ValueError("conv2d_group: channel mismatch")
    OH, OW = H + 2 * pad - KH + 1, W + 2 * pad - KW + 1
    wino = 1 if wino_ok(KH, KW, 1, pad, Cin, pixel_shuffle, work=float(n) * N * OH * OW * Cin * Cout) else 0
    wps = [tape.pack(w, Cout, Cin, KH, KW, 1, 0, 1, pad, wino=wino) for w in ws]
    oshape = (N, Cout // 4, OH * 2, OW * 2) if pixel_shuffle == 2 else (N, Cout, OH, OW)
    ys = list(outs) if outs is not None else [new(oshape, x0.device) for _ in range(n)]
    assert all(tuple(y.shape) == oshape for y in ys)
    lrp = lrp_auxs is not None
    assert not (pixel_shuffle and (lrp or ress is not None))
    y2s = [new((N, Cout, OH, OW), x0.device) for _ in range(n)] if lrp else None
    epi = EPI_LRP if lrp else EPI_NONE
    # materialised operands: used only when EVERY member has one (a launch has one activation flag)
    ops = [_operand(tape, v) for v in xvs]
    if all(a == ACT_NONE for _, a in ops):
        xfs, actf = [t for t, _ in ops], ACT_NONE
    else:
        xfs, actf = [v.t for v in xvs], act
    resf = None
    if ress is not None:
        assert not lrp and all(r.act == ress[0].act for r in ress) and ress[0].act in (ACT_NONE, ACT_GELU)
        rops = [_operand(tape, r) for r in ress]
        if all(a == ACT_NONE for _, a in rops):
            resf, epi = [t for t, _ in rops], EPI_RES
        else:
            resf, epi = [r.t for r in ress], (EPI_RES_GELU if ress[0].act == ACT_GELU else EPI_RES)
    if act_out and _MATERIALIZE and not lrp and N * OH * OW >= _MAT_MIN_PIXELS:
        if EVAL_INPLACE_ACT and not tape.need_grad and outs is None:
            y2s = list(ys)   # inference: only gelu(y) is stored (see conv2d)
        else:
            y2s = [new(oshape, x0.device) for _ in range(n)]
        for y, y2 in zip(ys, y2s):
            tape.mat[_key(y)] = y2
    conv_launch_grouped(tape, xfs, wps, bs_, ys, Cin=Cin, Cout=Cout, KH=KH, KW=KW, stride=1, pad=pad,
                        transposed=0, OH=OH, OW=OW, pro_act=actf, epi=epi,
                        auxs=list(lrp_auxs) if lrp else None, y2s=y2s,
                        ress=resf, ps=pixel_shuffle, algo=wino)
    if not tape.need_grad:
        return ys

    def bwd():
        dys = [tape.grad_of(y) for y in ys]
        if any(d is None for d in dys):
            raise RuntimeError("conv2d_group: every member needs a gradient")
        if pixel_shuffle == 2:   # gradient of the fused PixelShuffle store
            un = []
            for dy in dys:
                du = torch.empty((N, Cout, OH, OW), dtype=torch.float32, device=x0.device)
                dyc = dy if dy.is_contiguous() else dy.contiguous()
                check(L.lib().icm_pixel_unshuffle2(ptr(dyc), ptr(du), N, Cout // 4, OH, OW, tape.st), "pixel_unshuffle2")
                un.append(du)
            dys = un
        if lrp:
            pre = []
            for dy, aux, y2 in zip(dys, lrp_auxs, y2s):
                accumulate(tape, aux, dy)
                dpre = torch.empty((N, Cout, OH, OW), dtype=torch.float32, device=x0.device)
                check(L.lib().icm_lrp_bwd(ptr(dy), bs(dy), ptr(y2), bs(y2), ptr(dpre), bs(dpre), N, Cout, OH * OW, tape.st),
                      "lrp_bwd")
                pre.append(dpre)
            dys = pre
        if ress is not None:
            for r, dy in zip(ress, dys):
                tape.defer_res_grad(r.t, dy, r.act == ACT_GELU)
        for xf_, w, b, dy in zip(xfs, ws, bs_, dys):
            gw, acc = tape.grad_for_write(w)
            gb_, accb = tape.grad_for_write(b)
            wgrad_defer(tape, dy, xf_, gw, Ca=Cout, Cb=Cin, KH=KH, KW=KW, stride=1, pad=pad, act_b=actf, accum=acc,
                        dbias=gb_, accum_bias=accb)
        # input gradients: one grouped dgrad.  Members that share an input tensor (the fixed support of the late
        # slices) write private buffers which are then summed into the shared gradient; distinct inputs are
        # written (or accumulated) in place.
        keys = [_key(v.t) for v in xvs]
        shared = len(set(keys)) < n
        rgs = None
        if not shared and act in (ACT_GELU, ACT_NONE) and all(k in tape._pending_res and
                                                              tape._pending_res[k][2] == (act == ACT_GELU) for k in keys):
            rgs = [tape.take_res_grad(v.t, act == ACT_GELU) for v in xvs]   # identity-path terms ride on this dgrad
        if not shared:
            gxs, accs = [], []
            for v in xvs:
                gx, ax = tape.grad_for_write(v.t)
                gxs.append(gx)
                accs.append(ax)
            if len(set(accs)) > 1:   # one launch, one accumulate flag: zero the buffers that nobody has written yet
                for gx, ax in zip(gxs, accs):
                    if not ax:
                        gx.zero_()
                accs = [1] * n
            acc0 = accs[0]
        else:
            gxs = [torch.empty((N, Cin, H, W), dtype=torch.float32, device=x0.device) for _ in range(n)]
            acc0 = 0
        wb = 2 if wino_ok(KH, KW, 1, pad, Cout, pixel_shuffle, work=float(n) * N * H * W * Cin * Cout) else 0
        wpb = [tape.pack(w, Cin, Cout, KH, KW, 0, 1, 1, pad, wino=wb) for w in ws]
        if act == ACT_GELU:
            epi_b = EPI_RES_MUL_DGELU if rgs is not None else EPI_MUL_DGELU
        else:
            epi_b = EPI_RES if rgs is not None else EPI_NONE
        conv_launch_grouped(tape, dys, wpb, None, gxs, Cin=Cout, Cout=Cin, KH=KH, KW=KW, stride=1, pad=pad,
                            transposed=1, OH=H, OW=W, epi=epi_b,
                            auxs=[v.t for v in xvs] if act == ACT_GELU else None, ress=rgs, accum=acc0, tag="dgrad",
                            algo=1 if wb else 0)
        if shared:
            for v, g in zip(xvs, gxs):
                accumulate(tape, v.t, g)

    tape.bw.append(bwd)
    return ys


def gdn(tape: Tape, x, beta, gamma, inverse: bool, beta_min: float = 1e-6) -> torch.Tensor:
    """GDN / IGDN (layers/gdn.py:62-75): one 1x1 implicit GEMM with x^2 prologue and rsqrt/sqrt epilogue."""
    N, Cc, H, W = x.shape
    if gamma.shape != (Cc, Cc) or beta.shape != (Cc,):
        raise ValueError("GDN: channel mismatch")
    bound_b = (beta_min + PEDESTAL) ** 0.5
    bound_g = PEDESTAL ** 0.5
    beta_eff = torch.empty_like(beta)
    check(L.lib().icm_nonneg_fwd(ptr(beta), ptr(beta_eff), Cc, bound_b, PEDESTAL, tape.st), "nonneg_fwd")
    wp = tape.pack(gamma, Cc, Cc, 1, 1, 1, 0, 1, 0, nonneg=1, bound=bound_g, ped=PEDESTAL)
    y = new(x)
    nrm = new(x) if tape.need_grad else None
    conv_launch(tape, x, wp, beta_eff, y, Cin=Cc, Cout=Cc, KH=1, KW=1, stride=1, pad=0, transposed=0, OH=H, OW=W,
                pro_act=ACT_SQUARE, epi=EPI_IGDN if inverse else EPI_GDN, aux=x, y2=nrm, tag="fwd(gdn)")
    if not tape.need_grad:
        return y

    def bwd():
        g = tape.grad_of(y)
        if g is None:
            return
        g = g if g.is_contiguous() else g.contiguous()
        dn = new(x)
        t1 = new(x)
        xc = x if x.is_contiguous() else x.contiguous()
        check(L.lib().icm_gdn_bwd_pre(ptr(g), ptr(xc), ptr(nrm), ptr(dn), ptr(t1), x.numel(), int(inverse), tape.st),
              "gdn_bwd_pre")
        if tape.wants(beta):
            de = torch.empty_like(beta)
            channel_sum(tape, dn, de, 0)
            gb_, acc = tape.grad_for_write(beta)
            check(L.lib().icm_nonneg_bwd(ptr(beta), ptr(de), ptr(gb_), Cc, bound_b, acc, tape.st), "nonneg_bwd")
        if tape.wants(gamma):
            dg = torch.empty_like(gamma)
            wgrad_launch(tape, dn, x, dg, Ca=Cc, Cb=Cc, KH=1, KW=1, stride=1, pad=0, act_b=ACT_SQUARE)
            gg, acc = tape.grad_for_write(gamma)
            check(L.lib().icm_nonneg_bwd(ptr(gamma), ptr(dg), ptr(gg), Cc * Cc, bound_g, acc, tape.st), "nonneg_bwd")
        if tape.wants(x):
            gx, acc = tape.grad_for_write(x)
            wpt = tape.pack(gamma, Cc, Cc, 1, 1, 0, 0, 1, 0, nonneg=1, bound=bound_g, ped=PEDESTAL)
            conv_launch(tape, dn, wpt, None, gx, Cin=Cc, Cout=Cc, KH=1, KW=1, stride=1, pad=0, transposed=0, OH=H,
                        OW=W, epi=EPI_AXPY2, aux=x, aux2=t1, accum=acc, tag="dgrad(gdn)")

    tape.bw.append(bwd)
    return y


# ------------------------------------------------------------------------------------------------ attention gate
def residual_unit(tape, xv: VT, P, p, last: bool = False) -> VT:
    """layers/layers.py:52-72 with virtual GELUs: returns VT(pre, GELU).  last: the unit in front of the gate, whose
    output the gate kernel reads as a pre-activation -- in inference it is not stored activated."""
    u1 = conv2d(tape, xv, P[p + ".conv.0.weight"], P[p + ".conv.0.bias"], act_out=True)
    u2 = conv2d(tape, VT(u1, ACT_GELU), P[p + ".conv.2.weight"], P[p + ".conv.2.bias"], pad=1, act_out=True)
    u3 = conv2d(tape, VT(u2, ACT_GELU), P[p + ".conv.4.weight"], P[p + ".conv.4.bias"], res=xv,
                act_out=tape.need_grad or not last)
    return VT(u3, ACT_GELU)


def residual_unit_pair(tape, xa: VT, xb: VT, P, pa, pb, last: bool = False):
    """the same ResidualUnit step of the two independent gate branches (conv_a[j], conv_b[j+1]; layers.py:75-81) as
    grouped launches: the 4 096-pixel gates (dim 320) fill only half the chip one branch at a time"""
    u1 = conv2d_group(tape, [xa, xb], [P[pa + ".conv.0.weight"], P[pb + ".conv.0.weight"]],
                      [P[pa + ".conv.0.bias"], P[pb + ".conv.0.bias"]], pad=0, act_out=True)
    u2 = conv2d_group(tape, [VT(u1[0], ACT_GELU), VT(u1[1], ACT_GELU)],
                      [P[pa + ".conv.2.weight"], P[pb + ".conv.2.weight"]],
                      [P[pa + ".conv.2.bias"], P[pb + ".conv.2.bias"]], pad=1, act_out=True)
    u3 = conv2d_group(tape, [VT(u2[0], ACT_GELU), VT(u2[1], ACT_GELU)],
                      [P[pa + ".conv.4.weight"], P[pb + ".conv.4.weight"]],
                      [P[pa + ".conv.4.bias"], P[pb + ".conv.4.bias"]], pad=0, ress=[xa, xb],
                      act_out=tape.need_grad or not last)
    return VT(u3[0], ACT_GELU), VT(u3[1], ACT_GELU)


def window_msa_core(tape, qkv, table, C_, heads, ws, shift) -> torch.Tensor:
    """softmax(q k^T / sqrt(hd) + bias (+ shift mask)) v per window and head, straight on the NCHW qkv tensor
    (win_attention.py:84-115 / stf.py:95-121; roll + window_partition + mask are address arithmetic)."""
    N, _, H, W = qkv.shape
    o = torch.empty((N, C_, H, W), dtype=torch.float32, device=qkv.device)
    e0 = _prof_begin()
    check(L.lib().icm_winattn_fwd(ptr(qkv), ptr(table), ptr(o), N, C_, H, W, heads, ws, shift, tape.st), "winattn_fwd")
    _prof_end(e0, f"winattn fwd dim{C_} ws{ws} @{H}x{W} n{N}", 4.0 * N * H * W * ws * ws * C_)
    if tape.need_grad:
        def bwd():
            do = tape.grad_of(o)
            if do is None:
                return
            dqkv, acc = tape.grad_for_write(qkv)
            assert acc == 0
            gt, acct = tape.grad_for_write(table)
            nws = L.lib().icm_winattn_bwd_workspace_floats(N, C_, H, W, heads, ws)
            wsp = tape.red_ws(nws, qkv.device)
            e0 = _prof_begin()
            check(L.lib().icm_winattn_bwd(ptr(qkv), ptr(table), ptr(do), ptr(dqkv), ptr(gt), acct, ptr(wsp), wsp.numel(),
                                          N, C_, H, W, heads, ws, shift, tape.st), "winattn_bwd")
            _prof_end(e0, f"winattn bwd dim{C_} ws{ws} @{H}x{W} n{N}", 8.0 * N * H * W * ws * ws * C_)
        tape.bw.append(bwd)
    return o


def window_attention(tape, x, P, p, heads, ws, shift) -> torch.Tensor:
    """WinBasedAttention.forward (layers/win_attention.py:153-207): x + proj(attn(qkv(x)))."""
    N, Cc, H, W = x.shape
    if not (0 <= shift < ws):
        raise AssertionError("shift_size must in 0-window_size")
    table = P[p + ".attn.relative_position_bias_table"]
    qkv = conv2d(tape, VT(x), P[p + ".attn.qkv.weight"], P[p + ".attn.qkv.bias"])
    o = window_msa_core(tape, qkv, table, Cc, heads, ws, shift)
    return conv2d(tape, VT(o), P[p + ".attn.proj.weight"], P[p + ".attn.proj.bias"], res=VT(x))


# ------------------------------------------------------------------------------------------------ stf (Swin) pieces
LN_EPS = 1e-5


def layernorm(tape, x, gamma, beta) -> torch.Tensor:
    """nn.LayerNorm(C) over the channel axis of an NCHW tensor (= per token; stf.py:136,142,200,246,350)."""
    N, Cc, H, W = x.shape
    if gamma.shape != (Cc,) or beta.shape != (Cc,):
        raise ValueError("LayerNorm: channel mismatch")
    HW = H * W
    y = torch.empty((N, Cc, H, W), dtype=torch.float32, device=x.device)
    need = tape.need_grad
    mean = torch.empty(N * HW, dtype=torch.float32, device=x.device) if need else None
    rstd = torch.empty(N * HW, dtype=torch.float32, device=x.device) if need else None
    check(L.lib().icm_layernorm_fwd(ptr(x), bs(x), ptr(gamma), ptr(beta), ptr(y), bs(y), ptr(mean), ptr(rstd), N, Cc, HW,
                                    LN_EPS, tape.st), "layernorm_fwd")
    if need:
        def bwd():
            dy = tape.grad_of(y)
            if dy is None:
                return
            gg = gb_ = dx = None
            ap = ax = 0
            if tape.wants(gamma):
                gg, ap = tape.grad_for_write(gamma)
                gb_, ap2 = tape.grad_for_write(beta)
                assert ap == ap2
            rg = None
            if tape.wants(x):
                rg = tape.take_res_grad(x, False)   # x + f(LN(x)): the identity-path term rides on this pass
                dx, ax = tape.grad_for_write(x)
            wsp = tape.red_ws(2 * 1024 * Cc, x.device)   # one partial row of dgamma / dbeta per workgroup of the fused pass
            check(L.lib().icm_layernorm_bwd(ptr(x), bs(x), ptr(dy), bs(dy), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx),
                                            bs(dx), ptr(gg), ptr(gb_), N, Cc, HW, ax, ap, ptr(rg), bs(rg), ptr(wsp),
                                            wsp.numel(), tape.st), "layernorm_bwd")
        tape.bw.append(bwd)
    return y


def residual_scale(tape, shortcut, branch, scale) -> torch.Tensor:
    """shortcut + DropPath(branch) with per-sample scales (0 or 1/keep_prob; stf.py:190-191)."""
    N = branch.shape[0]
    per = branch[0].numel()
    assert shortcut.is_contiguous() and branch.is_contiguous()
    out = torch.empty_like(branch)
    check(L.lib().icm_residual_scale(ptr(shortcut), ptr(branch), ptr(scale), ptr(out), N, per, tape.st), "residual_scale")
    if tape.need_grad:
        def bwd():
            g = tape.grad_of(out)
            if g is None:
                return
            tape.defer_res_grad(shortcut, g, False)   # rides on the LayerNorm backward that writes d(shortcut) next
            db, acc = tape.grad_for_write(branch)
            assert acc == 0 and db.is_contiguous()
            check(L.lib().icm_residual_scale(0, ptr(g), ptr(scale), ptr(db), N, per, tape.st), "residual_scale_bwd")
        tape.bw.append(bwd)
    return out


def swin_block(tape, x, P, p, heads, ws, shift, dp=None) -> torch.Tensor:
    """SwinTransformerBlock.forward (stf.py:149-193) on an NCHW map: LN -> (S)W-MSA -> +shortcut -> LN -> MLP -> +.
    dp: None or a [2, N] device tensor of DropPath scales (attention branch, MLP branch)."""
    N, Cc, H, W = x.shape
    if H % ws or W % ws:
        raise ValueError("swin_block: feature map must be a multiple of the window size (inputs are multiples of 64)")
    n1 = layernorm(tape, x, P[p + ".norm1.weight"], P[p + ".norm1.bias"])
    qkv = conv2d(tape, VT(n1), P[p + ".attn.qkv.weight"], P[p + ".attn.qkv.bias"])
    o = window_msa_core(tape, qkv, P[p + ".attn.relative_position_bias_table"], Cc, heads, ws, shift)
    if dp is None:
        x1 = conv2d(tape, VT(o), P[p + ".attn.proj.weight"], P[p + ".attn.proj.bias"], res=VT(x))
    else:
        a = conv2d(tape, VT(o), P[p + ".attn.proj.weight"], P[p + ".attn.proj.bias"])
        x1 = residual_scale(tape, x, a, dp[0])
    n2 = layernorm(tape, x1, P[p + ".norm2.weight"], P[p + ".norm2.bias"])
    hdn = conv2d(tape, VT(n2), P[p + ".mlp.fc1.weight"], P[p + ".mlp.fc1.bias"], act_out=True)
    if dp is None:
        return conv2d(tape, VT(hdn, ACT_GELU), P[p + ".mlp.fc2.weight"], P[p + ".mlp.fc2.bias"], res=VT(x1))
    m = conv2d(tape, VT(hdn, ACT_GELU), P[p + ".mlp.fc2.weight"], P[p + ".mlp.fc2.bias"])
    return residual_scale(tape, x1, m, dp[1])


def patch_merging(tape, x, P, p) -> torch.Tensor:
    """PatchMerging.forward (stf.py:203-233): 2x2 gather -> LayerNorm(4C) -> Linear(4C, 2C, bias=False)."""
    N, Cc, H, W = x.shape
    if H % 2 or W % 2:
        raise ValueError("patch_merging: odd feature maps need padding (inputs are multiples of 64)")
    assert x.is_contiguous()
    g4 = torch.empty((N, 4 * Cc, H // 2, W // 2), dtype=torch.float32, device=x.device)
    check(L.lib().icm_space_to_depth2(ptr(x), ptr(g4), N, Cc, H, W, 0, 0, tape.st), "space_to_depth2")
    if tape.need_grad:
        def bwd():
            g = tape.grad_of(g4)
            if g is None or not tape.wants(x):
                return
            dx, acc = tape.grad_for_write(x)
            assert dx.is_contiguous() and g.is_contiguous()
            check(L.lib().icm_space_to_depth2(ptr(g), ptr(dx), N, Cc, H, W, 1, acc, tape.st), "depth_to_space2")
        tape.bw.append(bwd)
    n = layernorm(tape, g4, P[p + ".norm.weight"], P[p + ".norm.bias"])
    return conv2d(tape, VT(n), P[p + ".reduction.weight"], None)


def patch_split(tape, x, P, p) -> torch.Tensor:
    """PatchSplit.forward (stf.py:249-259): LayerNorm(C) -> Linear(C, 2C, bias=False) -> PixelShuffle(2) (fused store)."""
    n = layernorm(tape, x, P[p + ".norm.weight"], P[p + ".norm.bias"])
    return conv2d(tape, VT(n), P[p + ".reduction.weight"], None, pixel_shuffle=2)


def attention_gate(tape, x, P, p, heads, ws, shift) -> torch.Tensor:
    """Win_noShift_Attention.forward (layers/layers.py:83-89): a*sigmoid(b) + x."""
    a = VT(x)
    b = VT(window_attention(tape, x, P, p + ".conv_b.0", heads, ws, shift))
    if PAIR_GATE_BRANCHES:
        for i in range(3):   # conv_a[i] and conv_b[i + 1] are independent and shape-identical
            a, b = residual_unit_pair(tape, a, b, P, f"{p}.conv_a.{i}", f"{p}.conv_b.{i + 1}", last=i == 2)
    else:
        for i in range(3):
            a = residual_unit(tape, a, P, f"{p}.conv_a.{i}", last=i == 2)
        for i in (1, 2, 3):
            b = residual_unit(tape, b, P, f"{p}.conv_b.{i}", last=i == 3)
    b4 = conv2d(tape, b, P[p + ".conv_b.4.weight"], P[p + ".conv_b.4.bias"])
    out = new(x)
    xc = x if x.is_contiguous() else x.contiguous()
    check(L.lib().icm_gate_fwd(ptr(a.t), ptr(b4), ptr(xc), ptr(out), x.numel(), tape.st), "gate_fwd")
    if tape.need_grad:
        def bwd():
            g = tape.grad_of(out)
            if g is None:
                return
            g = g if g.is_contiguous() else g.contiguous()
            da, acca = tape.grad_for_write(a.t)
            db, accb = tape.grad_for_write(b4)
            assert accb == 0
            if tape.wants(x):
                dx, accx = tape.grad_for_write(x)
                assert dx.is_contiguous()
            else:
                dx, accx = torch.empty_like(xc), 0
            check(L.lib().icm_gate_bwd(ptr(g), ptr(a.t), ptr(b4), ptr(da), ptr(db), ptr(dx), x.numel(), acca, accx,
                                       tape.st), "gate_bwd")
        tape.bw.append(bwd)
    return out


# ------------------------------------------------------------------------------------------------ entropy models
EB_NAMES = [f"_matrix{i}" for i in range(5)] + [f"_bias{i}" for i in range(5)] + [f"_factor{i}" for i in range(4)]


def _eb_params(P, p):
    s = L.EbParams()
    for i in range(5):
        s.matrix[i] = ptr(P[f"{p}._matrix{i}"])
        s.bias[i] = ptr(P[f"{p}._bias{i}"])
    for i in range(4):
        s.factor[i] = ptr(P[f"{p}._factor{i}"])
    s.quantiles = ptr(P[f"{p}.quantiles"])
    return s


def eb_likelihood(tape, z, P, p="entropy_bottleneck", noise=None, lik_bound=1e-9, want_zt=False):
    """EntropyBottleneck.forward (entropy_models.py:446-489) -> (z_tilde or None, likelihood)."""
    N, Cc = z.shape[0], z.shape[1]
    HW = z[0, 0].numel()
    zc = z if z.is_contiguous() else z.contiguous()
    if P[f"{p}._matrix0"].shape[0] != Cc:
        raise ValueError("EntropyBottleneck: channel mismatch")
    lik = torch.empty_like(zc)
    zt = torch.empty_like(zc) if want_zt else None
    prm = _eb_params(P, p)
    check(L.lib().icm_eb_likelihood_fwd(ptr(zc), ptr(noise), C.byref(prm), ptr(lik), ptr(zt), N, Cc, HW, lik_bound,
                                        tape.st), "eb_fwd")
    if tape.need_grad:
        def bwd():
            dl = tape.grad_of(lik)
            if dl is None and (zt is None or tape.grad_of(zt) is None):
                return
            if dl is None:
                dl = torch.zeros_like(lik)
            dl = dl if dl.is_contiguous() else dl.contiguous()
            g = L.EbGrads()
            for i in range(5):
                gm, a1 = tape.grad_for_write(P[f"{p}._matrix{i}"])
                gb_, a2 = tape.grad_for_write(P[f"{p}._bias{i}"])
                assert a1 == 0 and a2 == 0
                g.matrix[i], g.bias[i] = ptr(gm), ptr(gb_)
            for i in range(4):
                gf, a3 = tape.grad_for_write(P[f"{p}._factor{i}"])
                assert a3 == 0
                g.factor[i] = ptr(gf)
            dmed = None
            if noise is None:
                dmed = torch.empty(Cc, dtype=torch.float32, device=z.device)
                g.dmedian = ptr(dmed)
            if tape.wants(z):
                dz, acc = tape.grad_for_write(z)
                assert dz.is_contiguous()
            else:
                dz, acc = torch.empty_like(zc), 0
            check(L.lib().icm_eb_likelihood_bwd(ptr(zc), ptr(noise), C.byref(prm), ptr(dl), ptr(dz), C.byref(g), N, Cc,
                                                HW, lik_bound, acc, tape.st), "eb_bwd")
            # the returned outputs z~ = quantize(z, "noise" | "dequantize", medians) carry a gradient too
            # (entropy_models.py:126-150,468-472): identity to z in noise mode; in dequantize mode zero to z (round) and
            # one to the medians (added back after the round)
            dzt = tape.grad_of(zt) if zt is not None else None
            if dzt is not None:
                if noise is not None:
                    accumulate(tape, z, dzt.contiguous())
                elif dmed is not None:
                    dm2 = torch.empty(Cc, dtype=torch.float32, device=z.device)
                    channel_sum(tape, dzt.contiguous().view(N, Cc, -1, 1), dm2, 0)
                    check(L.lib().icm_add_grad(ptr(dm2), 0, ptr(dmed), Cc, 1, tape.st), "add_grad")
            if dmed is not None and tape.wants(P[f"{p}.quantiles"]):
                gq, accq = tape.grad_for_write(P[f"{p}.quantiles"])
                if not accq:
                    gq.zero_()
                gq[:, 0, 1] += dmed
        tape.bw.append(bwd)
    return zt, lik


def ste_round_medians(tape, z, quantiles):
    """z_hat = ste_round(z - med) + med (cnn.py:150-152)."""
    N, Cc = z.shape[0], z.shape[1]
    zc = z if z.is_contiguous() else z.contiguous()
    zh = torch.empty_like(zc)
    check(L.lib().icm_ste_round_offset(ptr(zc), ptr(quantiles), ptr(zh), N, Cc, z[0, 0].numel(), tape.st), "ste_round")
    if tape.need_grad:
        def bwd():
            g = tape.grad_of(zh)
            if g is not None:
                accumulate(tape, z, g)
        tape.bw.append(bwd)
    return zh


def gc_likelihood_ste(tape, y, mu, scale, noise, lik_out, yh_out, yh2_out=None, scale_bound=0.11, lik_bound=1e-9):
    """GaussianConditional.forward fused with ste_round(y-mu)+mu (entropy_models.py:645-659, cnn.py:171-173)."""
    N, Cc, H, W = y.shape
    check(L.lib().icm_gc_likelihood_ste_fwd(ptr(y), bs(y), ptr(mu), bs(mu), ptr(scale), bs(scale), ptr(noise),
                                            bs(noise), ptr(lik_out), bs(lik_out), ptr(yh_out), bs(yh_out),
                                            ptr(yh2_out), bs(yh2_out), N, Cc, H * W, scale_bound, lik_bound, tape.st),
          "gc_fwd")
    if tape.need_grad:
        def bwd():
            dl = tape.grad_of(lik_out)
            dyh = tape.grad_of(yh_out) if yh_out is not None else None
            if dl is None and dyh is None:
                return
            if dl is None:
                dl = torch.zeros((N, Cc, H, W), dtype=torch.float32, device=y.device)
            dy, acc = tape.grad_for_write(y)
            dmu, a1 = tape.grad_for_write(mu)
            dsc, a2 = tape.grad_for_write(scale)
            assert a1 == 0 and a2 == 0
            check(L.lib().icm_gc_likelihood_ste_bwd(ptr(y), bs(y), ptr(mu), bs(mu), ptr(scale), bs(scale), ptr(noise),
                                                    bs(noise), ptr(dl), bs(dl), ptr(dyh), bs(dyh), ptr(dy), bs(dy),
                                                    ptr(dmu), bs(dmu), ptr(dsc), bs(dsc), N, Cc, H * W, scale_bound,
                                                    lik_bound, acc, tape.st), "gc_bwd")
        tape.bw.append(bwd)



# ------------------------------------------------------------------------------------------------ stf6 plumbing
def add(tape, a, b) -> torch.Tensor:
    """a + b of two contiguous tensors with both gradients (stf6.py:812: mu = mu + refinement)"""
    N = a.shape[0]
    per = a[0].numel()
    assert a.is_contiguous() and b.is_contiguous() and a.shape == b.shape
    out = torch.empty_like(a)
    ones = torch.ones(N, dtype=torch.float32, device=a.device)
    check(L.lib().icm_residual_scale(ptr(a), ptr(b), ptr(ones), ptr(out), N, per, tape.st), "add")
    if tape.need_grad:
        def bwd():
            g = tape.grad_of(out)
            if g is None:
                return
            accumulate(tape, a, g)
            accumulate(tape, b, g)
        tape.bw.append(bwd)
    return out


def zigzag_splits(tape, x, num_slices: int, nH: int = 2, nW: int = 2) -> torch.Tensor:
    """ZigzagSplits (stf6.py:654-714) as one permutation launch: [B,C,H,W] -> [B, ns*nH*nW, C/ns, H/nH, W/nW].
    The per-block views z[:, n] are what the slice loop consumes: their gradients alias one zeroed buffer that the
    backward permutes back in one launch."""
    B, Cc, H, W = x.shape
    if Cc % num_slices or H % nH or W % nW:
        raise ValueError("zigzag_splits: the latent does not split into exact blocks")
    nb = num_slices * nH * nW
    z = new((B, nb, Cc // num_slices, H // nH, W // nW), x.device)
    check(L.lib().icm_zigzag_splits(ptr(x), bs(x), ptr(z), B, Cc, H, W, num_slices, nH, nW, tape.st), "zigzag_splits")
    if tape.need_grad and tape.wants(x):
        dz = zeros(z.shape, x.device)
        tape.bind_grad(z, dz, True)
        for n in range(nb):
            tape.bind_grad(z[:, n], dz[:, n], True)

        def bwd():
            tmp = new((B, Cc, H, W), x.device)
            check(L.lib().icm_zigzag_reverse(ptr(dz), ptr(tmp), Cc * H * W, B, Cc, H, W, num_slices, nH, nW, tape.st),
                  "zigzag_splits_bwd")
            accumulate(tape, x, tmp)
        tape.bw.append(bwd)
    else:
        tape.stop(z)
    return z


def zigzag_reverse(tape, z, num_slices: int, nH: int = 2, nW: int = 2) -> torch.Tensor:
    """ZigzagReverse (stf6.py:716-762): [B, N, Cs, Hb, Wb] -> [B, Cs*ns, Hb*nH, Wb*nW]"""
    B, nb, Cs, Hb, Wb = z.shape
    Cc, H, W = Cs * num_slices, Hb * nH, Wb * nW
    assert z.is_contiguous() and nb == num_slices * nH * nW
    x = new((B, Cc, H, W), z.device)
    check(L.lib().icm_zigzag_reverse(ptr(z), ptr(x), Cc * H * W, B, Cc, H, W, num_slices, nH, nW, tape.st), "zigzag_reverse")
    if tape.need_grad:
        def bwd():
            g = tape.grad_of(x)
            if g is None:
                return
            tmp = new(z.shape, z.device)
            check(L.lib().icm_zigzag_splits(ptr(g), bs(g), ptr(tmp), B, Cc, H, W, num_slices, nH, nW, tape.st),
                  "zigzag_reverse_bwd")
            accumulate(tape, z, tmp)
        tape.bw.append(bwd)
    return x

# ------------------------------------------------------------------------------------------------ autograd bridge
class _TapeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, runner, nstop, *tensors):
        tape = Tape(need_grad=True, packed_cache=runner.packed_cache)
        outs = runner(tape, *tensors)
        ctx.tape, ctx.tensors, ctx.outs = tape, tensors, outs
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        tape = ctx.tape
        tape.st = L.stream()
        for o, g in zip(ctx.outs, gouts):
            if g is not None:
                tape.bind_grad(o, g.contiguous(), True)
        tape.backward()
        res = []
        for t, need in zip(ctx.tensors, ctx.needs_input_grad[2:]):
            res.append(tape.grad_of(t) if need else None)
        ctx.tape = None
        return (None, None, *res)


def tape_function(runner, tensors: Sequence[torch.Tensor], packed_cache: Optional[dict] = None):
    """Run ``runner(tape, *tensors) -> tuple(outputs)`` as ONE autograd node whose backward is the tape.
    packed_cache: see Tape (eval-mode calls of a module keep their MFMA-order weight copies between calls)."""
    if torch.is_grad_enabled() and any(t.requires_grad for t in tensors):
        return _TapeFn.apply(_Needs(runner, tensors, packed_cache), 0, *tensors)
    tape = Tape(need_grad=False, packed_cache=packed_cache)
    with torch.no_grad():
        return tuple(runner(tape, *[t.detach() for t in tensors]))


class _Needs:
    """callable wrapper that marks non-differentiable inputs as stopped before running"""

    def __init__(self, runner, tensors, packed_cache=None):
        self.runner = runner
        self.packed_cache = packed_cache
        self.flags = [bool(t.requires_grad) for t in tensors]

    def __call__(self, tape, *ts):
        for t, f in zip(ts, self.flags):
            if not f:
                tape.stop(t)
        return self.runner(tape, *ts)
