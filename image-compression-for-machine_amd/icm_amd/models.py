"""compressai/models mirror for the `cnn` model: CompressionModel (models/base.py) and WACNN (models/cnn.py).

``WACNN.forward(x)`` keeps the reference contract ``{"x_hat", "likelihoods": {"y", "z"}}`` and the 585-key
state-dict, but executes as ONE hand-scheduled tape of HIP kernels (``wacnn_forward``): torch.cat / chunk are
replaced by writes into persistent support buffers, GELU / PixelShuffle / residual adds / the LRP tail are
fused into the implicit-GEMM prologues and epilogues, and the backward pass is the tape's closures."""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _lib as L
from . import engine as E
from ._lib import ACT_GELU
from .engine import VT
from .entropy_models import EntropyBottleneck, GaussianConditional
from .layers import (GDN, Conv2d, Win_noShift_Attention, WindowAttention, conv, conv3x3, deconv, subpel_conv3x3,
                     _named)

SCALES_MIN, SCALES_MAX, SCALES_LEVELS = 0.11, 256, 64


def get_scale_table(min=SCALES_MIN, max=SCALES_MAX, levels=SCALES_LEVELS):
    return torch.exp(torch.linspace(math.log(min), math.log(max), levels))


class CompressionModel(nn.Module):
    """models/base.py:5-70."""

    def __init__(self, init_weights=True):
        super().__init__()
        # the reference calls _initialize_weights() here, before any sub-module exists: a no-op

    def _pack_cache(self) -> Optional[dict]:
        """eval mode: the MFMA-order weight copies are kept between calls (inference serving: packing 300 MB of weights
        per call costs more than a batch-1 forward).  The cache is dropped whenever a parameter changed -- in place
        through torch (version counters) or through the HIP optimiser (engine weight generation); train mode and
        graph capture (icm_amd/graphs.py: the packing launches belong in the graph) do not use it."""
        if self.training or getattr(self, "_eval_cache_off", False):
            return None
        sig = (E._weight_gen[0], sum(p._version for p in self.parameters()))
        if getattr(self, "_pack_sig", None) != sig:
            self._pack_eval, self._pack_sig = {}, sig
        return self._pack_eval

    def aux_loss(self):
        return sum(m.loss() for m in self.modules() if isinstance(m, EntropyBottleneck))

    def forward(self, *args):
        raise NotImplementedError()

    def update(self, force=False):
        """models/base.py:43-60: refresh the CDF tables of every EntropyBottleneck child"""
        updated = False
        for m in self.children():
            if not isinstance(m, EntropyBottleneck):
                continue
            updated |= m.update(force=force)
        return updated

    # ---- shared by the cnn and stf models: entropy coding around the slice loop (cnn.py:133-139,210-332)
    def _update_tables(self, scale_table=None, force=False):
        if scale_table is None:
            scale_table = get_scale_table()
        updated = self.gaussian_conditional.update_scale_table(scale_table, force=force)
        updated |= CompressionModel.update(self, force=force)
        return updated

    def _params(self):
        names, params = _named(self)
        return dict(zip(names, [p.detach() for p in params]))

    def _compress_latent(self, tape, P, y, debug=None):
        """cnn.py:214-266 from ``y`` on: {"strings": [[y_string], z_strings], "shape": z spatial size}"""
        from .ans import _encode
        codec = {"gc": self.gaussian_conditional, "symbols": {}, "indexes": {}}
        keep = {}
        hyper_slices(tape, P, y, None, None, self.num_slices, self.max_support_slices, keep=keep, codec=codec)
        z = keep["z"]
        z_strings = self.entropy_bottleneck.compress(z)
        sym = torch.cat([codec["symbols"][i].reshape(-1) for i in range(self.num_slices)]).cpu().numpy()
        idx = torch.cat([codec["indexes"][i].reshape(-1) for i in range(self.num_slices)]).cpu().numpy()
        self.gaussian_conditional._check_cdf_size()
        self.gaussian_conditional._check_cdf_length()
        self.gaussian_conditional._check_offsets_size()
        y_string = _encode(sym, idx, self.gaussian_conditional._tables())
        if debug is not None:      # tests: the coded symbols / CDF indexes (slice-major, cnn.py:254-255)
            debug.update(symbols=sym, indexes=idx)
        return {"strings": [[y_string], z_strings], "shape": z.size()[-2:]}

    def _decompress_latent(self, tape, P, strings, shape, M):
        """cnn.py:289-328: y_hat from the two streams"""
        if not isinstance(strings, (list, tuple)) or len(strings) != 2:
            raise ValueError("strings must be [y_strings, z_strings]")
        z_hat = self.entropy_bottleneck.decompress(strings[1], shape)
        return decode_slices(tape, P, z_hat, strings[0][0], self.gaussian_conditional, self.num_slices,
                             self.max_support_slices, M)

    def load_state_dict(self, state_dict, strict: bool = False):
        # models/base.py:62-70: resize the CDF buffers to whatever the checkpoint holds, then strict=False
        sd = dict(state_dict)
        for mod_name in ("entropy_bottleneck", "gaussian_conditional"):
            mod = getattr(self, mod_name, None)
            if mod is None:
                continue
            for b in ("_quantized_cdf", "_offset", "_cdf_length", "scale_table"):
                k = f"{mod_name}.{b}"
                if k in sd and hasattr(mod, b):
                    buf = getattr(mod, b)
                    if buf.numel() == 0 and sd[k].numel() != 0:
                        setattr(mod, b, torch.empty(sd[k].shape, dtype=buf.dtype, device=buf.device))
        return super().load_state_dict(sd, strict=False)


def _chain(tape, P, p, xv: VT, strides=(1, 1, 1, 1, 1), out=None, lrp_aux=None):
    """five conv3x3 with virtual GELUs between (cnn.py:54-64, 89-127)."""
    for j, i in enumerate((0, 2, 4, 6, 8)):
        last = j == 4
        t = E.conv2d(tape, xv, P[f"{p}.{i}.weight"], P[f"{p}.{i}.bias"], stride=strides[j], pad=1,
                     out=out if last else None, lrp_aux=lrp_aux if last else None, act_out=not last)
        xv = VT(t, ACT_GELU)
    return t


def _chain_pair(tape, P, p1, p2, xv1: VT, xv2: VT):
    """cc_mean_transforms[i] and cc_scale_transforms[i] are independent and shape-identical: run each of
    their five convolutions as one grouped launch (forward and dgrad)."""
    for i in (0, 2, 4, 6, 8):
        t1, t2 = E.conv2d_group(tape, [xv1, xv2], [P[f"{p1}.{i}.weight"], P[f"{p2}.{i}.weight"]],
                                [P[f"{p1}.{i}.bias"], P[f"{p2}.{i}.bias"]], pad=1, act_out=i != 8)
        xv1, xv2 = VT(t1, ACT_GELU), VT(t2, ACT_GELU)
    return t1, t2


def _h_s(tape, P, p, z_hat, out):
    """h_mean_s / h_scale_s (cnn.py:66-88): PixelShuffle fused into the subpel convs' stores."""
    u = E.conv2d(tape, VT(z_hat), P[p + ".0.weight"], P[p + ".0.bias"], pad=1, act_out=True)
    u = E.conv2d(tape, VT(u, ACT_GELU), P[p + ".2.0.weight"], P[p + ".2.0.bias"], pad=1, pixel_shuffle=2, act_out=True)
    u = E.conv2d(tape, VT(u, ACT_GELU), P[p + ".4.weight"], P[p + ".4.bias"], pad=1, act_out=True)
    u = E.conv2d(tape, VT(u, ACT_GELU), P[p + ".6.0.weight"], P[p + ".6.0.bias"], pad=1, pixel_shuffle=2, act_out=True)
    return E.conv2d(tape, VT(u, ACT_GELU), P[p + ".8.weight"], P[p + ".8.bias"], pad=1, out=out)


def _h_s_pair(tape, P, p1, p2, z_hat, out1, out2):
    """h_scale_s and h_mean_s (cnn.py:66-88) are independent and shape-identical: each of their five convolutions
    (two with the PixelShuffle store) runs as one grouped launch"""
    xv = [VT(z_hat), VT(z_hat)]
    for name, ps, last in ((".0", 0, False), (".2.0", 2, False), (".4", 0, False), (".6.0", 2, False), (".8", 0, True)):
        ts = E.conv2d_group(tape, xv, [P[p1 + name + ".weight"], P[p2 + name + ".weight"]],
                            [P[p1 + name + ".bias"], P[p2 + name + ".bias"]], pad=1, pixel_shuffle=ps,
                            outs=[out1, out2] if last else None, act_out=not last)
        xv = [VT(ts[0], ACT_GELU), VT(ts[1], ACT_GELU)]
    return ts


def _copy_op(tape, src, dst):
    """dst = src (cat/chunk plumbing) with gradient routed back"""
    E.copy_into(tape, src, dst)
    if tape.need_grad:
        def bwd():
            g = tape.grad_of(dst)
            if g is not None:
                E.accumulate(tape, src, g)
        tape.bw.append(bwd)


# ICM_SLICE_SPLIT=1 (default): the first layer of every slice chain is split by input-channel block (slices.py): the
# latent block of all 3 * num_slices chains runs as two wide convolutions outside the serial slice loop.  0 = the
# per-chain form below (support buffers, cat -> chain).  Both are kept and both are parity-tested at the bench geometry.
# Same-box A/B on MI355X (round 3, B=16): with the direct kernels only the split form LOSES (329.7 vs 335.7 img/s: its
# wide launches have few output tiles, and the support / own-slice blocks add ~30 small launches per direction); with
# the eight-wave Winograd kernel, whose wide co blocks are exactly what the 4 256-channel latent launch wants (460 us,
# 218 TF algorithmic), training is a tie (370.9 vs 371.8) and the eval forward gains 3 % (1 216 vs 1 180 img/s).
import os as _os
SLICE_SPLIT = _os.environ.get("ICM_SLICE_SPLIT", "1") != "0"


def hyper_slices(tape: E.Tape, P: Dict[str, torch.Tensor], y: torch.Tensor, noise_z, noise_y, num_slices: int,
                 max_support: int, keep: Optional[dict] = None, bucket_marks: Optional[dict] = None,
                 batch_tail: bool = True, codec: Optional[dict] = None, decode: Optional[dict] = None):
    """Hyperprior + channel-conditional slice loop shared by the cnn and stf models
    (cnn.py:144-183 == stf.py:596-637) -> (y_hat, y_likelihoods, z_likelihoods).  torch.cat / chunk become
    channel-slice views of persistent support buffers.
    codec (compress(), cnn.py:246-252): {"gc": GaussianConditional, "symbols": {}, "indexes": {}} -- the int32 symbols
    round(y - mu) and CDF indexes of every slice are recorded (device tensors, keyed by slice).
    decode (decompress(), cnn.py:289-326): {"z_hat", "M", "slice": callable(i, mu, sc, yh_pre)} -- y is None; the
    SAME launch sequence runs (identical kernels on identical inputs give the bit-identical mu / scale the encoder saw,
    which the CDF indexes depend on), but each slice's y_hat_pre comes from the entropy decoder instead of from y."""
    if SLICE_SPLIT and batch_tail:
        from .slices import hyper_slices_split
        return hyper_slices_split(tape, P, y, noise_z, noise_y, num_slices, max_support, keep, bucket_marks, codec,
                                  decode, lambda t, P_, y_: _chain(t, P_, "h_a", VT(y_), strides=(1, 1, 2, 1, 2)),
                                  _h_s_pair, _record_symbols)
    if decode is not None:
        z_hat = decode["z_hat"].contiguous()
        dev, N, M = z_hat.device, z_hat.shape[0], decode["M"]
        h, w = z_hat.shape[2] * 4, z_hat.shape[3] * 4
        if tape.need_grad:
            raise ValueError("decode mode is inference only")
    else:
        dev = y.device
        N = y.shape[0]
        M, h, w = y.shape[1], y.shape[2], y.shape[3]
    need = tape.need_grad
    if M % num_slices != 0:
        raise ValueError("latent channels must divide into num_slices")
    sc_ = M // num_slices
    if need:  # y is consumed whole (h_a) and by slices (GaussianConditional): one aliased gradient buffer
        dY = E.zeros(y.shape, dev)
        tape.bind_grad(y, dY, True)
        for i in range(num_slices):
            tape.bind_grad(y[:, i * sc_:(i + 1) * sc_], dY[:, i * sc_:(i + 1) * sc_], True)
    if bucket_marks is not None:
        bucket_marks[2] = len(tape.bw)   # backward reaching here => hyper-path gradients are complete
    # ---- h_a + entropy bottleneck (cnn.py:144-152)
    z = z_lik = None
    if decode is None:
        z = _chain(tape, P, "h_a", VT(y), strides=(1, 1, 2, 1, 2))
        _, z_lik = E.eb_likelihood(tape, z, P, "entropy_bottleneck", noise_z)
        z_hat = E.ste_round_medians(tape, z, P["entropy_bottleneck.quantiles"])
    # ---- hyper synthesis straight into the support buffers (cnn.py:154-155,163,167)
    nsup = sc_ * max_support
    MS = E.new((N, M + nsup, h, w), dev)
    SS = E.new((N, M + nsup, h, w), dev)
    if need:
        dMS, dSS = E.zeros(MS.shape, dev), E.zeros(SS.shape, dev)
        for k in range(max_support + 1):
            tape.bind_grad(MS[:, :M + sc_ * k], dMS[:, :M + sc_ * k], True)
            tape.bind_grad(SS[:, :M + sc_ * k], dSS[:, :M + sc_ * k], True)
        for j in range(max_support):
            sl = slice(M + sc_ * j, M + sc_ * (j + 1))
            tape.bind_grad(MS[:, sl], dMS[:, sl], True)
            tape.bind_grad(SS[:, sl], dSS[:, sl], True)
    if (h % 4) or (w % 4):
        raise ValueError("hyper-synthesis output does not match the latent size (input must be a multiple of 64)")
    _h_s_pair(tape, P, "h_scale_s", "h_mean_s", z_hat, SS[:, :M], MS[:, :M])
    Y_hat = E.new((N, M, h, w), dev)
    Y_lik = E.new((N, M, h, w), dev) if decode is None else None
    if need:
        dYh = E.zeros(Y_hat.shape, dev)
        tape.bind_grad(Y_hat, dYh, True)
        for i in range(num_slices):
            tape.bind_grad(Y_hat[:, i * sc_:(i + 1) * sc_], dYh[:, i * sc_:(i + 1) * sc_], True)
    mus, scs = [], []
    if bucket_marks is not None:
        bucket_marks[1] = len(tape.bw)   # => slice-chain gradients complete
    # ---- channel-conditional slice loop (cnn.py:161-180).  The support of slice i is y_hat_slices[:max_support]
    # (cnn.py:161): slices 0..max_support-1 form a serial chain, but every slice >= max_support sees the SAME fixed
    # support (latent + the first max_support slices) and is independent of its neighbours -> their chains run as
    # grouped launches (2 * n_tail mean/scale chains, then n_tail lrp chains).
    n_serial = min(num_slices, max_support) if batch_tail else num_slices
    for i in range(n_serial):
        k = min(i, max_support)
        ch = slice(i * sc_, (i + 1) * sc_)
        ms, ss = MS[:, :M + sc_ * k], SS[:, :M + sc_ * k]
        mu, sc = _chain_pair(tape, P, f"cc_mean_transforms.{i}", f"cc_scale_transforms.{i}", VT(ms), VT(ss))
        LS = E.new((N, M + sc_ * (k + 1), h, w), dev)
        yh_pre = LS[:, M + sc_ * k:]
        if need:
            dLS = E.zeros(LS.shape, dev)
            tape.bind_grad(LS, dLS, True)
            tape.bind_grad(LS[:, :M + sc_ * k], dLS[:, :M + sc_ * k], True)
            tape.bind_grad(yh_pre, dLS[:, M + sc_ * k:], True)
        _copy_op(tape, ms, LS[:, :M + sc_ * k])
        if decode is not None:
            decode["slice"](i, mu, sc, yh_pre)
        else:
            E.gc_likelihood_ste(tape, y[:, ch], mu, sc, None if noise_y is None else noise_y[:, ch], Y_lik[:, ch],
                                yh_pre)
        if codec is not None:
            _record_symbols(codec, i, y[:, ch], mu, sc)
        _chain(tape, P, f"lrp_transforms.{i}", VT(LS), out=Y_hat[:, ch], lrp_aux=yh_pre)
        if i < max_support:
            sl = slice(M + sc_ * i, M + sc_ * (i + 1))
            _copy_op(tape, Y_hat[:, ch], MS[:, sl])
            _copy_op(tape, Y_hat[:, ch], SS[:, sl])
        if keep is not None:
            mus.append(mu)
            scs.append(sc)
    tail = list(range(n_serial, num_slices))
    for t0 in range(0, len(tail), E.MAX_GROUP // 2):
        idx = tail[t0:t0 + E.MAX_GROUP // 2]
        nt = len(idx)
        k = max_support
        ms, ss = MS[:, :M + sc_ * k], SS[:, :M + sc_ * k]
        # mean / scale chains of all tail slices: one grouped launch per layer
        names = [f"cc_mean_transforms.{i}" for i in idx] + [f"cc_scale_transforms.{i}" for i in idx]
        xvs = [VT(ms)] * nt + [VT(ss)] * nt
        for li in (0, 2, 4, 6, 8):
            ts = E.conv2d_group(tape, xvs, [P[f"{p}.{li}.weight"] for p in names], [P[f"{p}.{li}.bias"] for p in names],
                                pad=1, act_out=li != 8)
            xvs = [VT(t, ACT_GELU) for t in ts]
        mu_t, sc_t = ts[:nt], ts[nt:]
        LSs, pres = [], []
        for j, i in enumerate(idx):
            ch = slice(i * sc_, (i + 1) * sc_)
            LS = E.new((N, M + sc_ * (k + 1), h, w), dev)
            yh_pre = LS[:, M + sc_ * k:]
            if need:
                dLS = E.zeros(LS.shape, dev)
                tape.bind_grad(LS, dLS, True)
                tape.bind_grad(LS[:, :M + sc_ * k], dLS[:, :M + sc_ * k], True)
                tape.bind_grad(yh_pre, dLS[:, M + sc_ * k:], True)
            _copy_op(tape, ms, LS[:, :M + sc_ * k])
            if decode is not None:
                decode["slice"](i, mu_t[j], sc_t[j], yh_pre)
            else:
                E.gc_likelihood_ste(tape, y[:, ch], mu_t[j], sc_t[j], None if noise_y is None else noise_y[:, ch],
                                    Y_lik[:, ch], yh_pre)
            if codec is not None:
                _record_symbols(codec, i, y[:, ch], mu_t[j], sc_t[j])
            LSs.append(LS)
            pres.append(yh_pre)
        # lrp chains of all tail slices, LRP tail fused into the last grouped launch
        lnames = [f"lrp_transforms.{i}" for i in idx]
        xvs = [VT(LS) for LS in LSs]
        for li in (0, 2, 4, 6):
            ts = E.conv2d_group(tape, xvs, [P[f"{p}.{li}.weight"] for p in lnames],
                                [P[f"{p}.{li}.bias"] for p in lnames], pad=1, act_out=True)
            xvs = [VT(t, ACT_GELU) for t in ts]
        E.conv2d_group(tape, xvs, [P[f"{p}.8.weight"] for p in lnames], [P[f"{p}.8.bias"] for p in lnames], pad=1,
                       outs=[Y_hat[:, i * sc_:(i + 1) * sc_] for i in idx], lrp_auxs=pres)
        if keep is not None:
            mus.extend(mu_t)
            scs.extend(sc_t)
    if bucket_marks is not None:
        bucket_marks[0] = len(tape.bw)   # => synthesis-transform gradients complete
    if keep is not None:
        keep.update(y=y, z=z, z_hat=z_hat, y_hat=Y_hat, mu=torch.cat(mus, 1), scale=torch.cat(scs, 1),
                    lat_means=MS[:, :M], lat_scales=SS[:, :M])
    return Y_hat, Y_lik, z_lik


def _record_symbols(codec, i, y_slice, mu, sc):
    """symbols = quantize(y_slice, "symbols", mu) and indexes = build_indexes(scale) of slice i (cnn.py:246-252)"""
    N, Cc, h, w = y_slice.shape
    sym = torch.empty((N, Cc, h, w), dtype=torch.int32, device=y_slice.device)
    L.check(L.lib().icm_quantize(L.ptr(y_slice), L.bs(y_slice), L.ptr(mu), L.bs(mu), h * w, 1, sym.data_ptr(), 0, N, Cc,
                                 h * w, L.stream()), "quantize")
    codec["symbols"][i] = sym
    codec["indexes"][i] = codec["gc"].build_indexes(sc)


def decode_slices(tape: E.Tape, P, z_hat, y_string: bytes, gc, num_slices: int, max_support: int, M: int):
    """Decoder side of the slice loop (cnn.py:296-326): hyper_slices in decode mode -- per slice the chains give
    mu / scale -> CDF indexes -> the rANS decoder yields the symbols -> y_hat_pre = symbols + mu; the LRP correction
    and the support bookkeeping are the forward's own.  Returns y_hat [N,M,h,w]."""
    from .ans import RansDecoder
    gc._check_cdf_size()
    gc._check_cdf_length()
    gc._check_offsets_size()
    tabs = gc._tables()
    dec = RansDecoder()
    dec.set_stream(y_string)
    state = {"next": 0}

    def one(i, mu, sc, yh_pre):
        if i != state["next"]:
            raise RuntimeError("slices must be decoded in stream order")
        state["next"] += 1
        N, Cc, h, w = mu.shape
        idx = gc.build_indexes(sc).cpu().numpy().reshape(-1)
        sym = torch.from_numpy(dec.decode_stream_np(idx, tabs).reshape(N, Cc, h, w)).to(mu.device)
        L.check(L.lib().icm_dequantize(sym.data_ptr(), L.ptr(mu), L.bs(mu), h * w, 1, L.ptr(yh_pre), L.bs(yh_pre), N, Cc,
                                       h * w, tape.st), "dequantize")

    Y_hat, _, _ = hyper_slices(tape, P, None, None, None, num_slices, max_support,
                               decode={"z_hat": z_hat, "M": M, "slice": one})
    return Y_hat


def _split_lik(tape, Y_lik, num_slices):
    """runs FIRST in backward (registered last): hand the seeded d(y_likelihoods) to the per-slice consumers"""
    sc_ = Y_lik.shape[1] // num_slices

    def split():
        g = tape.grad_of(Y_lik)
        if g is not None:
            for i in range(num_slices):
                tape.bind_grad(Y_lik[:, i * sc_:(i + 1) * sc_], g[:, i * sc_:(i + 1) * sc_], True)
    tape.bw.append(split)


def wacnn_forward(tape: E.Tape, P: Dict[str, torch.Tensor], x: torch.Tensor, noise_z=None, noise_y=None,
                  num_slices: int = 10, max_support: int = 5, keep: Optional[dict] = None,
                  bucket_marks: Optional[dict] = None):
    """WACNN.forward (models/cnn.py:141-189) on the HIP engine -> (x_hat, y_likelihoods, z_likelihoods)."""
    y = wacnn_g_a(tape, P, x)
    Y_hat, Y_lik, z_lik = hyper_slices(tape, P, y, noise_z, noise_y, num_slices, max_support, keep, bucket_marks)
    x_hat = wacnn_g_s(tape, P, Y_hat)
    if tape.need_grad:
        _split_lik(tape, Y_lik, num_slices)
    return x_hat, Y_lik, z_lik


def wacnn_g_a(tape: E.Tape, P, x):
    """analysis transform g_a (cnn.py:31-41)"""
    t = E.conv2d_thin_in(tape, x, P["g_a.0.weight"], P["g_a.0.bias"], stride=2, pad=2)
    t = E.gdn(tape, t, P["g_a.1.beta"], P["g_a.1.gamma"], False)
    t = E.conv2d(tape, VT(t), P["g_a.2.weight"], P["g_a.2.bias"], stride=2, pad=2)
    t = E.gdn(tape, t, P["g_a.3.beta"], P["g_a.3.gamma"], False)
    t = E.attention_gate(tape, t, P, "g_a.4", 8, 8, 4)
    t = E.conv2d(tape, VT(t), P["g_a.5.weight"], P["g_a.5.bias"], stride=2, pad=2)
    t = E.gdn(tape, t, P["g_a.6.beta"], P["g_a.6.gamma"], False)
    t = E.conv2d(tape, VT(t), P["g_a.7.weight"], P["g_a.7.bias"], stride=2, pad=2)
    return E.attention_gate(tape, t, P, "g_a.8", 8, 4, 2)


def wacnn_g_s(tape: E.Tape, P, Y_hat):
    """synthesis transform g_s (cnn.py:42-52)"""
    t = E.attention_gate(tape, Y_hat, P, "g_s.0", 8, 4, 2)
    t = E.conv2d(tape, VT(t), P["g_s.1.weight"], P["g_s.1.bias"], stride=2, pad=2, transposed=True, output_padding=1)
    t = E.gdn(tape, t, P["g_s.2.beta"], P["g_s.2.gamma"], True)
    t = E.conv2d(tape, VT(t), P["g_s.3.weight"], P["g_s.3.bias"], stride=2, pad=2, transposed=True, output_padding=1)
    t = E.gdn(tape, t, P["g_s.4.beta"], P["g_s.4.gamma"], True)
    t = E.attention_gate(tape, t, P, "g_s.5", 8, 8, 4)
    t = E.conv2d(tape, VT(t), P["g_s.6.weight"], P["g_s.6.bias"], stride=2, pad=2, transposed=True, output_padding=1)
    t = E.gdn(tape, t, P["g_s.7.beta"], P["g_s.7.gamma"], True)
    return E.convT2d_thin_out(tape, VT(t), P["g_s.8.weight"], P["g_s.8.bias"], stride=2, pad=2, output_padding=1)


# ------------------------------------------------------------------------------------------------ stf
STF_DEPTHS = (2, 2, 6, 2)
STF_HEADS = (3, 6, 12, 24)


def stf_drop_path_rates(drop_path_rate: float = 0.2, depths=STF_DEPTHS) -> Dict[str, float]:
    """stf.py:357 + 361-398: linspace(0, rate, sum(depths)); the synthesis layers index the same list with the
    reversed depths."""
    dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
    out, o = {}, 0
    for i, d in enumerate(depths):
        for j in range(d):
            out[f"layers.{i}.blocks.{j}"] = dpr[o + j]
        o += d
    o = 0
    for i, d in enumerate(depths[::-1]):
        for j in range(d):
            out[f"syn_layers.{i}.blocks.{j}"] = dpr[o + j]
        o += d
    return out


def _basic_layer(tape, P, p, t, depth, heads, ws, down, drops):
    """BasicLayer.forward (stf.py:271-313): the SW-MSA mask is address arithmetic inside the attention kernel."""
    for j in range(depth):
        key = f"{p}.blocks.{j}"
        t = E.swin_block(tape, t, P, key, heads, ws, 0 if j % 2 == 0 else ws // 2,
                         None if drops is None else drops.get(key))
    if down == "merge":
        return E.patch_merging(tape, t, P, p + ".downsample")
    if down == "split":
        return E.patch_split(tape, t, P, p + ".downsample")
    return t


def stf_forward(tape: E.Tape, P: Dict[str, torch.Tensor], x: torch.Tensor, noise_z=None, noise_y=None,
                drops: Optional[Dict[str, torch.Tensor]] = None, num_slices: int = 12, max_support: int = 6,
                window: int = 4, keep: Optional[dict] = None, bucket_marks: Optional[dict] = None):
    """SymmetricalTransFormer.forward (models/stf.py:582-645) on the HIP engine.  Tokens stay NCHW end to end:
    [B, L, C] <-> [B, C, H, W] transposes of the reference vanish, Linear layers are 1x1 implicit GEMMs,
    LayerNorm normalises the channel axis per pixel.  drops: {"<layer>.blocks.<j>": [2,B] DropPath scales}."""
    y = stf_analysis(tape, P, x, drops, window)
    Y_hat, Y_lik, z_lik = hyper_slices(tape, P, y, noise_z, noise_y, num_slices, max_support, keep, bucket_marks)
    x_hat = stf_synthesis(tape, P, Y_hat, drops, window)
    if tape.need_grad:
        _split_lik(tape, Y_lik, num_slices)
    return x_hat, Y_lik, z_lik


def stf_analysis(tape: E.Tape, P, x, drops=None, window: int = 4):
    """patch_embed + layers (stf.py:582-595)"""
    if x.shape[2] % 2 or x.shape[3] % 2:
        raise ValueError("stf: odd image sizes need PatchEmbed padding (pad the input to a multiple of 64)")
    # ---- patch_embed (stf.py:331-351): conv 2x2 s2 + LayerNorm
    t = E.conv2d_thin_in(tape, x, P["patch_embed.proj.weight"], P["patch_embed.proj.bias"], stride=2, pad=0)
    t = E.layernorm(tape, t, P["patch_embed.norm.weight"], P["patch_embed.norm.bias"])
    for i in range(4):
        t = _basic_layer(tape, P, f"layers.{i}", t, STF_DEPTHS[i], STF_HEADS[i], window, "merge" if i < 3 else None,
                         drops)
    return t


def stf_synthesis(tape: E.Tape, P, Y_hat, drops=None, window: int = 4):
    """syn_layers + end_conv (stf.py:638-645)"""
    t = Y_hat
    for i in range(4):
        t = _basic_layer(tape, P, f"syn_layers.{i}", t, STF_DEPTHS[3 - i], STF_HEADS[3 - i], window,
                         "split" if i < 3 else None, drops)
    # ---- end_conv (stf.py:401-404): conv5x5 -> PixelShuffle(2) (fused store) -> conv3x3
    t = E.conv2d(tape, VT(t), P["end_conv.0.weight"], P["end_conv.0.bias"], pad=2, pixel_shuffle=2)
    if E.THIN_OUT:   # 48 -> 3 output channels: dense 27-row GEMM + col2im instead of a 32-row tile per tap
        return E.conv2d_thin_out(tape, VT(t), P["end_conv.2.weight"], P["end_conv.2.bias"], pad=1)
    return E.conv2d(tape, VT(t), P["end_conv.2.weight"], P["end_conv.2.bias"], pad=1)



# ------------------------------------------------------------------------------------------------ stf6 (zigzag + Swin-refined means)
STF6_SLICES, STF6_NUMBER, STF6_SUPPORT, STF6_MU_HEADS = 6, 2, 16, 4


def stf6_drop_path_rates(drop_path_rate: float = 0.2, depths=STF_DEPTHS, nblocks: int = STF6_SLICES * 4) -> Dict[str, float]:
    """stf's rates plus the refinement stacks (stf6.py:469-484: built after ``depths = depths[::-1]``)"""
    out = stf_drop_path_rates(drop_path_rate, depths)
    dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
    for b in range(nblocks):
        o = 0
        for i, d in enumerate(depths[::-1]):
            for j in range(d):
                out[f"mu_Swin.{b}.{i}.blocks.{j}"] = dpr[o + j]
            o += d
    return out


def hyper_slices_zigzag(tape: E.Tape, P: Dict[str, torch.Tensor], y: torch.Tensor, noise_z, noise_y, drops,
                        num_slices: int = STF6_SLICES, max_support: int = STF6_SUPPORT, window: int = 4,
                        keep: Optional[dict] = None, bucket_marks: Optional[dict] = None):
    """Hyperprior + zigzag slice loop of SymmetricalTransFormer3.forward (stf6.py:778-858) ->
    (y_hat [B,M,h,w], y_likelihoods [B, nb*M/ns, h/2, w/2] in zigzag block order, z_likelihoods).
    The latent, its means and scales are cut into nb = ns*2*2 blocks by one permutation launch each; block i is coded
    from cat[means_i, y_hat of the previous min(i, max_support) blocks]: the y_hat blocks live in ONE buffer in coding
    order, so that support is a contiguous channel range copied behind the block's own means; the mean is refined by
    four Swin BasicLayers on the block map (mu_Swin[i]) before the likelihood; ZigzagReverse assembles y_hat.
    noise_y: None or [B, nb, M/ns, h/2, w/2] (zigzag order)."""
    dev, N, M, h, w = y.device, y.shape[0], y.shape[1], y.shape[2], y.shape[3]
    need = tape.need_grad
    nH = nW = STF6_NUMBER
    nb, cs, hb, wb = num_slices * nH * nW, M // num_slices, h // nH, w // nW
    if M % num_slices or h % nH or w % nW or hb % window or wb % window:
        raise ValueError("stf6: the latent must split into 2 x 2 blocks that are multiples of the window (input % 128 == 0)")
    if bucket_marks is not None:
        bucket_marks[2] = len(tape.bw)
    z = _chain(tape, P, "h_a", VT(y), strides=(1, 1, 2, 1, 2))
    _, z_lik = E.eb_likelihood(tape, z, P, "entropy_bottleneck", noise_z)
    z_hat = E.ste_round_medians(tape, z, P["entropy_bottleneck.quantiles"])
    lat_scales, lat_means = E.new((N, M, h, w), dev), E.new((N, M, h, w), dev)
    _h_s_pair(tape, P, "h_scale_s", "h_mean_s", z_hat, lat_scales, lat_means)
    y_zz = E.zigzag_splits(tape, y, num_slices, nH, nW)
    sc_zz = E.zigzag_splits(tape, lat_scales, num_slices, nH, nW)
    mu_zz = E.zigzag_splits(tape, lat_means, num_slices, nH, nW)
    YH = E.new((N, nb, cs, hb, wb), dev)          # y_hat blocks in coding order
    YHc = YH.view(N, nb * cs, hb, wb)
    Y_lik = E.new((N, nb, cs, hb, wb), dev)
    if need:
        dYH = E.zeros(YH.shape, dev)
        dYHc = dYH.view(N, nb * cs, hb, wb)
        tape.bind_grad(YH, dYH, True)
        for i in range(nb):
            tape.bind_grad(YH[:, i], dYH[:, i], True)
            k = min(i, max_support)
            if k > 1:
                tape.bind_grad(YHc[:, cs * (i - k):cs * i], dYHc[:, cs * (i - k):cs * i], True)
    if bucket_marks is not None:
        bucket_marks[1] = len(tape.bw)
    mus, scs = [], []
    rdepths = STF_DEPTHS[::-1]
    for i in range(nb):
        k = min(i, max_support)
        MSi, SSi = E.new((N, cs * (1 + k), hb, wb), dev), E.new((N, cs * (1 + k), hb, wb), dev)
        LSi = E.new((N, cs * (2 + k), hb, wb), dev)
        yh_pre = LSi[:, cs * (1 + k):]
        if need:
            for buf in (MSi, SSi, LSi):
                d = E.zeros(buf.shape, dev)
                tape.bind_grad(buf, d, True)
                tape.bind_grad(buf[:, :cs], d[:, :cs], True)
                if k > 0:
                    tape.bind_grad(buf[:, cs:cs * (1 + k)], d[:, cs:cs * (1 + k)], True)
                if buf is LSi:
                    tape.bind_grad(buf[:, :cs * (1 + k)], d[:, :cs * (1 + k)], True)
                    tape.bind_grad(yh_pre, d[:, cs * (1 + k):], True)
        _copy_op(tape, mu_zz[:, i], MSi[:, :cs])
        _copy_op(tape, sc_zz[:, i], SSi[:, :cs])
        if k > 0:
            sup = YHc[:, cs * (i - k):cs * i]
            _copy_op(tape, sup, MSi[:, cs:])
            _copy_op(tape, sup, SSi[:, cs:])
        mu, sc = _chain_pair(tape, P, f"cc_mean_transforms2.{i}", f"cc_scale_transforms2.{i}", VT(MSi), VT(SSi))
        t = mu
        for l in range(4):
            t = _basic_layer(tape, P, f"mu_Swin.{i}.{l}", t, rdepths[l], STF6_MU_HEADS, window, None, drops)
        mu = E.add(tape, mu, t)
        E.gc_likelihood_ste(tape, y_zz[:, i], mu, sc, None if noise_y is None else noise_y[:, i], Y_lik[:, i], yh_pre)
        _copy_op(tape, MSi, LSi[:, :cs * (1 + k)])
        _chain(tape, P, f"lrp_transforms2.{i}", VT(LSi), out=YH[:, i], lrp_aux=yh_pre)
        if keep is not None:
            mus.append(mu)
            scs.append(sc)
    if bucket_marks is not None:
        bucket_marks[0] = len(tape.bw)
    Y_hat = E.zigzag_reverse(tape, YH, num_slices, nH, nW)
    Y_lik4 = Y_lik.view(N, nb * cs, hb, wb)
    if keep is not None:
        keep.update(y=y, z=z, z_hat=z_hat, y_hat=Y_hat, y_zz=y_zz, mu=torch.stack(mus, 1), scale=torch.stack(scs, 1))
    return Y_hat, Y_lik4, z_lik


def stf6_forward(tape: E.Tape, P: Dict[str, torch.Tensor], x: torch.Tensor, noise_z=None, noise_y=None,
                 drops: Optional[Dict[str, torch.Tensor]] = None, window: int = 4, keep: Optional[dict] = None,
                 bucket_marks: Optional[dict] = None):
    """SymmetricalTransFormer3.forward (models/stf6.py:764-872) on the HIP engine: stf's analysis / synthesis stacks
    around the zigzag slice loop."""
    y = stf_analysis(tape, P, x, drops, window)
    Y_hat, Y_lik, z_lik = hyper_slices_zigzag(tape, P, y, noise_z, noise_y, drops, window=window, keep=keep,
                                              bucket_marks=bucket_marks)
    x_hat = stf_synthesis(tape, P, Y_hat, drops, window)
    if tape.need_grad:
        _split_lik(tape, Y_lik, STF6_SLICES * 4)
    return x_hat, Y_lik, z_lik


def _pad_eval_input(x, training: bool):
    """Inputs whose sides are not multiples of 64 (six stride-2 stages).  The reference pads feature maps inside its
    Swin blocks (stf.py:158-163) and crops the slice-chain outputs (cnn.py:165,169), but its own evaluation entry
    point zero-pads the IMAGE to a multiple of 64 first (utils/eval_model/__main__.py:162-175), after which none of
    those branches is taken; this mirror does exactly that outer padding in eval mode and crops x_hat back
    (likelihoods cover the padded image, as they do in the reference's eval loop).  Training crops are 256x256."""
    if x.shape[2] % 64 == 0 and x.shape[3] % 64 == 0:
        return x, None
    if training or torch.is_grad_enabled() and x.requires_grad:
        raise ValueError("training inputs must be multiples of 64 (the reference trains on 256x256 crops)")
    from .utils import pad_to_multiple
    return pad_to_multiple(x, 64)


def _crop_eval_output(x_hat, pads):
    if pads is None:
        return x_hat
    from .utils import crop
    return crop(x_hat, pads)


def _check_codec_input(x):
    if x.dim() != 4 or x.shape[1] != 3:
        raise ValueError("expected [B,3,H,W]")
    if x.shape[2] % 64 or x.shape[3] % 64:
        raise ValueError("compress(): pad the image to a multiple of 64 first (icm_amd.utils.pad_to_multiple; the "
                         "reference's eval loop does the same, utils/eval_model/__main__.py:102-117)")


def _train_noise(injected, x, cz, cy):
    """U(-1/2,1/2) samples the entropy models add in train mode (entropy_models.py:131-135)"""
    B, _, H, W = x.shape
    dev = x.device
    if injected is not None:
        return (injected["z"].to(dev, torch.float32).contiguous(), injected["y"].to(dev, torch.float32).contiguous())
    return (torch.rand((B, cz, H // 64, W // 64), dtype=torch.float32, device=dev) - 0.5,
            torch.rand((B, cy, H // 16, W // 16), dtype=torch.float32, device=dev) - 0.5)


class WACNN(CompressionModel):
    """CNN based model (models/cnn.py:23-189): same modules, names and defaults as the reference."""

    def __init__(self, N=192, M=320, **kwargs):
        super().__init__(**kwargs)
        self.num_slices = 10
        self.max_support_slices = 5
        self.g_a = nn.Sequential(
            conv(3, N, kernel_size=5, stride=2), GDN(N),
            conv(N, N, kernel_size=5, stride=2), GDN(N),
            Win_noShift_Attention(dim=N, num_heads=8, window_size=8, shift_size=4),
            conv(N, N, kernel_size=5, stride=2), GDN(N),
            conv(N, M, kernel_size=5, stride=2),
            Win_noShift_Attention(dim=M, num_heads=8, window_size=4, shift_size=2))
        self.g_s = nn.Sequential(
            Win_noShift_Attention(dim=M, num_heads=8, window_size=4, shift_size=2),
            deconv(M, N, kernel_size=5, stride=2), GDN(N, inverse=True),
            deconv(N, N, kernel_size=5, stride=2), GDN(N, inverse=True),
            Win_noShift_Attention(dim=N, num_heads=8, window_size=8, shift_size=4),
            deconv(N, N, kernel_size=5, stride=2), GDN(N, inverse=True),
            deconv(N, 3, kernel_size=5, stride=2))
        self.h_a = nn.Sequential(conv3x3(320, 320), nn.GELU(), conv3x3(320, 288), nn.GELU(),
                                 conv3x3(288, 256, stride=2), nn.GELU(), conv3x3(256, 224), nn.GELU(),
                                 conv3x3(224, 192, stride=2))

        def hs():
            return nn.Sequential(conv3x3(192, 192), nn.GELU(), subpel_conv3x3(192, 224, 2), nn.GELU(),
                                 conv3x3(224, 256), nn.GELU(), subpel_conv3x3(256, 288, 2), nn.GELU(),
                                 conv3x3(288, 320))
        self.h_mean_s = hs()
        self.h_scale_s = hs()

        def cc(extra):
            return nn.ModuleList(nn.Sequential(
                conv(320 + 32 * min(i + extra, 5 + extra), 224, stride=1, kernel_size=3), nn.GELU(),
                conv(224, 176, stride=1, kernel_size=3), nn.GELU(),
                conv(176, 128, stride=1, kernel_size=3), nn.GELU(),
                conv(128, 64, stride=1, kernel_size=3), nn.GELU(),
                conv(64, 32, stride=1, kernel_size=3)) for i in range(10))
        self.cc_mean_transforms = cc(0)
        self.cc_scale_transforms = cc(0)
        self.lrp_transforms = cc(1)
        self.entropy_bottleneck = EntropyBottleneck(N)
        self.gaussian_conditional = GaussianConditional(None)
        self._noise = None

    def inject_noise(self, noise: Optional[dict]):
        """testing hook: {"z": [B,192,h/4,w/4], "y": [B,320,h,w]} U(-1/2,1/2) samples used in train mode"""
        self._noise = noise

    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("WACNN.forward expects [B,3,H,W]")
        names, params = _named(self)
        training = self.training
        dev = x.device
        x, pads = _pad_eval_input(x, training)
        nz = ny = None
        if training:
            nz, ny = _train_noise(self._noise, x, 192, 320)
        ns, ms = self.num_slices, self.max_support_slices

        def runner(tape, xin, *ps):
            return wacnn_forward(tape, dict(zip(names, ps)), xin, nz, ny, ns, ms)

        x_hat, y_lik, z_lik = E.tape_function(runner, [x.contiguous(), *params], self._pack_cache())
        return {"x_hat": _crop_eval_output(x_hat, pads), "likelihoods": {"y": y_lik, "z": z_lik}}

    @classmethod
    def from_state_dict(cls, state_dict):
        net = cls(192, 320)
        net.load_state_dict(state_dict)
        return net

    def update(self, scale_table=None, force=False):
        """cnn.py:133-138"""
        return self._update_tables(scale_table, force)

    @torch.no_grad()
    def compress(self, x, _debug=None):
        """cnn.py:210-266 -> {"strings": [[y_string], z_strings], "shape": z.shape[-2:]}"""
        _check_codec_input(x)
        P = self._params()
        tape = E.Tape(need_grad=False, packed_cache=self._pack_cache())
        return self._compress_latent(tape, P, wacnn_g_a(tape, P, x.contiguous()), _debug)

    @torch.no_grad()
    def decompress(self, strings, shape):
        """cnn.py:289-332 -> {"x_hat"} clamped to [0, 1]"""
        P = self._params()
        tape = E.Tape(need_grad=False, packed_cache=self._pack_cache())
        x_hat = wacnn_g_s(tape, P, self._decompress_latent(tape, P, strings, shape, 320))
        L.check(L.lib().icm_clamp(L.ptr(x_hat), x_hat.numel(), 0.0, 1.0, tape.st), "clamp")
        return {"x_hat": x_hat}


# ------------------------------------------------------------------------------------------------ stf modules
class Mlp(nn.Module):
    """stf.py:24-40 (parameter holder; runs inside the model tape)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        if drop != 0.0:
            raise NotImplementedError("icm Mlp: dropout is unused by the reference")
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)


class SwinTransformerBlock(nn.Module):
    """stf.py:124-193.  forward(x, mask_matrix=None) takes an NCHW map [B, C, H, W] (the reference's token layout
    [B, H*W, C] is its transpose; the mask is derived inside the kernel, the argument is ignored)."""

    def __init__(self, dim, num_heads, window_size=7, shift_size=0, mlp_ratio=4., qkv_bias=True, qk_scale=None,
                 drop=0., attn_drop=0., drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm, inverse=False):
        super().__init__()
        self.dim, self.num_heads, self.window_size, self.shift_size = dim, num_heads, window_size, shift_size
        self.mlp_ratio = mlp_ratio
        assert 0 <= self.shift_size < self.window_size, "shift_size must in 0-window_size"
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(dim, window_size=(window_size, window_size), num_heads=num_heads,
                                    qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path_rate = float(drop_path)
        self.drop_path = nn.Identity()   # stochastic depth is drawn by the model forward (per-sample scales)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.H = self.W = None

    def forward(self, x, mask_matrix=None):
        from .layers import run_module
        h, ws, sh = self.num_heads, self.window_size, self.shift_size
        dp = _draw_drop(self.drop_path_rate, x.shape[0], x.device) if self.training else None
        return run_module(self, lambda tape, P, t: (E.swin_block(tape, t, {"b." + k: v for k, v in P.items()}, "b", h,
                                                                 ws, sh, dp),), x.contiguous())[0]


class PatchMerging(nn.Module):
    """stf.py:196-233 on NCHW maps: [B, C, H, W] -> [B, 2C, H/2, W/2]."""

    def __init__(self, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim = dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = norm_layer(4 * dim)

    def forward(self, x, H=None, W=None):
        from .layers import run_module
        return run_module(self, lambda tape, P, t: (E.patch_merging(tape, t, {"d." + k: v for k, v in P.items()}, "d"),),
                          x.contiguous())[0]


class PatchSplit(nn.Module):
    """stf.py:236-259 on NCHW maps: [B, C, H, W] -> [B, C/2, 2H, 2W]."""

    def __init__(self, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim = dim
        self.reduction = nn.Linear(dim, dim * 2, bias=False)
        self.norm = norm_layer(dim)
        self.shuffle = nn.PixelShuffle(2)

    def forward(self, x, H=None, W=None):
        from .layers import run_module
        return run_module(self, lambda tape, P, t: (E.patch_split(tape, t, {"d." + k: v for k, v in P.items()}, "d"),),
                          x.contiguous())[0]


class BasicLayer(nn.Module):
    """stf.py:261-313 (parameter holder; stf_forward walks the blocks)."""

    def __init__(self, dim, depth, num_heads, window_size=7, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop=0.,
                 attn_drop=0., drop_path=0., norm_layer=nn.LayerNorm, downsample=None, use_checkpoint=False,
                 inverse=False):
        super().__init__()
        self.window_size, self.shift_size, self.depth = window_size, window_size // 2, depth
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim=dim, num_heads=num_heads, window_size=window_size,
                                 shift_size=0 if (i % 2 == 0) else window_size // 2, mlp_ratio=mlp_ratio,
                                 qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop, attn_drop=attn_drop,
                                 drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path,
                                 norm_layer=norm_layer, inverse=inverse) for i in range(depth)])
        self.downsample = downsample(dim=dim, norm_layer=norm_layer) if downsample is not None else None


class PatchEmbed(nn.Module):
    """stf.py:316-351."""

    def __init__(self, patch_size=4, in_chans=3, embed_dim=96, norm_layer=None):
        super().__init__()
        self.patch_size = (patch_size, patch_size)
        self.in_chans, self.embed_dim = in_chans, embed_dim
        self.proj = Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None


def _draw_drop(rate: float, B: int, device, generator=None):
    """timm DropPath in train mode: per-sample Bernoulli(keep)/keep, one draw per residual branch -> [2, B]"""
    if rate <= 0.0:
        return None
    keep = 1.0 - rate
    return (torch.rand((2, B), device=device, generator=generator) < keep).to(torch.float32) / keep


class SymmetricalTransFormer(CompressionModel):
    """Swin-transformer codec (models/stf.py:318-645): same modules, names, defaults and state-dict as the
    reference; ``forward`` runs as one tape of HIP kernels (``stf_forward``)."""

    def __init__(self, pretrain_img_size=256, patch_size=2, in_chans=3, embed_dim=48, depths=[2, 2, 6, 2],
                 num_heads=[3, 6, 12, 24], window_size=4, num_slices=12, mlp_ratio=4., qkv_bias=True, qk_scale=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0.2, norm_layer=nn.LayerNorm, patch_norm=True,
                 frozen_stages=-1, use_checkpoint=False):
        super().__init__()
        if (patch_size, in_chans, embed_dim, list(depths), list(num_heads), mlp_ratio, patch_norm) != \
                (2, 3, 48, [2, 2, 6, 2], [3, 6, 12, 24], 4., True) or drop_rate or attn_drop_rate or frozen_stages >= 0:
            raise NotImplementedError("icm SymmetricalTransFormer: only the reference's default architecture")
        self.pretrain_img_size = pretrain_img_size
        self.num_layers = len(depths)
        self.embed_dim = embed_dim
        self.patch_norm = patch_norm
        self.frozen_stages = frozen_stages
        self.num_slices = num_slices
        self.max_support_slices = num_slices // 2
        self.window_size = window_size
        self.drop_path_rate = drop_path_rate
        self.patch_embed = PatchEmbed(patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim,
                                      norm_layer=norm_layer if patch_norm else None)
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(BasicLayer(
                dim=int(embed_dim * 2 ** i), depth=depths[i], num_heads=num_heads[i], window_size=window_size,
                mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate, attn_drop=attn_drop_rate,
                drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])], norm_layer=norm_layer,
                downsample=PatchMerging if (i < self.num_layers - 1) else None, inverse=False))
        depths, num_heads = depths[::-1], num_heads[::-1]
        self.syn_layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.syn_layers.append(BasicLayer(
                dim=int(embed_dim * 2 ** (3 - i)), depth=depths[i], num_heads=num_heads[i], window_size=window_size,
                mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate, attn_drop=attn_drop_rate,
                drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])], norm_layer=norm_layer,
                downsample=PatchSplit if (i < self.num_layers - 1) else None, inverse=True))
        self.end_conv = nn.Sequential(Conv2d(embed_dim, embed_dim * patch_size ** 2, kernel_size=5, stride=1, padding=2),
                                      nn.PixelShuffle(patch_size),
                                      Conv2d(embed_dim, 3, kernel_size=3, stride=1, padding=1))
        self.num_features = [int(embed_dim * 2 ** i) for i in range(self.num_layers)]
        self.g_a = None
        self.g_s = None
        self.h_a = nn.Sequential(conv3x3(384, 384), nn.GELU(), conv3x3(384, 336), nn.GELU(),
                                 conv3x3(336, 288, stride=2), nn.GELU(), conv3x3(288, 240), nn.GELU(),
                                 conv3x3(240, 192, stride=2))

        def hs():
            return nn.Sequential(conv3x3(192, 240), nn.GELU(), subpel_conv3x3(240, 288, 2), nn.GELU(),
                                 conv3x3(288, 336), nn.GELU(), subpel_conv3x3(336, 384, 2), nn.GELU(),
                                 conv3x3(384, 384))
        self.h_mean_s = hs()
        self.h_scale_s = hs()

        def cc(extra):
            return nn.ModuleList(nn.Sequential(
                conv(384 + 32 * min(i + extra, 6 + extra), 224, stride=1, kernel_size=3), nn.GELU(),
                conv(224, 176, stride=1, kernel_size=3), nn.GELU(),
                conv(176, 128, stride=1, kernel_size=3), nn.GELU(),
                conv(128, 64, stride=1, kernel_size=3), nn.GELU(),
                conv(64, 32, stride=1, kernel_size=3)) for i in range(num_slices))
        self.cc_mean_transforms = cc(0)
        self.cc_scale_transforms = cc(0)
        self.lrp_transforms = cc(1)
        self.entropy_bottleneck = EntropyBottleneck(embed_dim * 4)
        self.gaussian_conditional = GaussianConditional(None)
        self._noise = None
        self._drops = None

    def inject_noise(self, noise: Optional[dict], drops: Optional[dict] = None):
        """testing hook: {"z","y"} U(-1/2,1/2) samples and {"<layer>.blocks.<j>": [2,B]} DropPath scales (train mode)"""
        self._noise, self._drops = noise, drops

    def init_weights(self):
        """stf.py:567-579"""
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=.02, a=-2.0, b=2.0)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)

    def draw_drops(self, B: int, device, generator=None) -> Dict[str, torch.Tensor]:
        out = {}
        for k, r in stf_drop_path_rates(self.drop_path_rate).items():
            d = _draw_drop(r, B, device, generator)
            if d is not None:
                out[k] = d
        return out

    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("SymmetricalTransFormer.forward expects [B,3,H,W]")
        names, params = _named(self)
        dev = x.device
        x, pads = _pad_eval_input(x, self.training)
        nz = ny = drops = None
        if self.training:
            nz, ny = _train_noise(self._noise, x, 192, 384)
            drops = self._drops if self._drops is not None else self.draw_drops(x.shape[0], dev)
            drops = {k: v.to(dev, torch.float32).contiguous() for k, v in drops.items()}
        ns, ms, ws = self.num_slices, self.max_support_slices, self.window_size

        def runner(tape, xin, *ps):
            return stf_forward(tape, dict(zip(names, ps)), xin, nz, ny, drops, ns, ms, ws)

        x_hat, y_lik, z_lik = E.tape_function(runner, [x.contiguous(), *params], self._pack_cache())
        return {"x_hat": _crop_eval_output(x_hat, pads), "likelihoods": {"y": y_lik, "z": z_lik}}

    @classmethod
    def from_state_dict(cls, state_dict):
        net = cls()
        net.load_state_dict(state_dict)
        return net

    def update(self, scale_table=None, force=False):
        """stf.py: same table refresh as the cnn model (cnn.py:133-138)"""
        return self._update_tables(scale_table, force)

    @torch.no_grad()
    def compress(self, x, _debug=None):
        """stf.py compress(): analysis transform, then the shared latent coder"""
        _check_codec_input(x)
        P = self._params()
        tape = E.Tape(need_grad=False, packed_cache=self._pack_cache())
        return self._compress_latent(tape, P, stf_analysis(tape, P, x.contiguous(), None, self.window_size), _debug)

    @torch.no_grad()
    def decompress(self, strings, shape):
        P = self._params()
        tape = E.Tape(need_grad=False, packed_cache=self._pack_cache())
        y_hat = self._decompress_latent(tape, P, strings, shape, 384)
        x_hat = stf_synthesis(tape, P, y_hat, None, self.window_size)
        L.check(L.lib().icm_clamp(L.ptr(x_hat), x_hat.numel(), 0.0, 1.0, tape.st), "clamp")
        return {"x_hat": x_hat}


class SymmetricalTransFormer3(SymmetricalTransFormer):
    """``stf6`` of the reference's zoo (models/stf6.py:384-872): the stf analysis / synthesis transforms around a
    zigzag-ordered entropy model -- the latent is coded as 24 blocks (6 channel groups x 2 x 2 spatial halves,
    ``ZigzagSplits``), every block's mean is refined by a four-layer Swin stack (``mu_Swin``), up to 16 previous
    blocks condition the next.  Same module names and state-dict as the reference (``sigma_Swin`` / ``LRP_Swin`` are
    registered but unused by ``forward``, as there); inputs must be multiples of 128 (block maps are multiples of the
    4 x 4 window).  ``compress`` / ``decompress`` are not mirrored for this variant."""

    def __init__(self, pretrain_img_size=256, patch_size=2, in_chans=3, embed_dim=48, depths=[2, 2, 6, 2],
                 num_heads=[3, 6, 12, 24], window_size=4, num_slices=6, Mask_win_size=8, num_sliding=4, mlp_ratio=4.,
                 qkv_bias=True, qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.2,
                 norm_layer=nn.LayerNorm, patch_norm=True, frozen_stages=-1, use_checkpoint=False):
        if num_slices != 6:
            raise NotImplementedError("icm SymmetricalTransFormer3: only the reference's default architecture")
        super().__init__(pretrain_img_size, patch_size, in_chans, embed_dim, depths, num_heads, window_size, 12,
                         mlp_ratio, qkv_bias, qk_scale, drop_rate, attn_drop_rate, drop_path_rate, norm_layer, patch_norm,
                         frozen_stages, use_checkpoint)
        # re-register everything after syn_layers in the reference's order (stf6.py:465-621)
        tail = {}
        for name in ("end_conv", "h_a", "h_mean_s", "h_scale_s", "entropy_bottleneck", "gaussian_conditional"):
            tail[name] = getattr(self, name)
            delattr(self, name)
        for name in ("cc_mean_transforms", "cc_scale_transforms", "lrp_transforms"):
            delattr(self, name)
        self.num_slices = num_slices
        self.max_support_slices = STF6_SUPPORT
        self.Mask_win_size = Mask_win_size
        cs = 384 // num_slices
        rdepths = list(depths)[::-1]
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]

        def refine_stacks(n):
            return nn.ModuleList(nn.ModuleList(BasicLayer(
                dim=cs, depth=rdepths[i], num_heads=STF6_MU_HEADS, window_size=window_size, mlp_ratio=mlp_ratio,
                qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate, attn_drop=attn_drop_rate,
                drop_path=dpr[sum(rdepths[:i]):sum(rdepths[:i + 1])], norm_layer=norm_layer, downsample=None,
                inverse=True) for i in range(self.num_layers)) for _ in range(n))
        self.mu_Swin = refine_stacks(num_slices * 4)
        self.sigma_Swin = refine_stacks(num_slices)
        self.LRP_Swin = refine_stacks(num_slices)
        for name in ("end_conv", "h_a", "h_mean_s", "h_scale_s"):
            setattr(self, name, tail[name])

        def cc(extra):
            return nn.ModuleList(nn.Sequential(
                conv(cs + cs * min(i + extra, STF6_SUPPORT + extra), 224, stride=1, kernel_size=3), nn.GELU(),
                conv(224, 176, stride=1, kernel_size=3), nn.GELU(),
                conv(176, 128, stride=1, kernel_size=3), nn.GELU(),
                conv(128, 64, stride=1, kernel_size=3), nn.GELU(),
                conv(64, cs, stride=1, kernel_size=3)) for i in range(num_slices * 4))
        self.cc_mean_transforms2 = cc(0)
        self.cc_scale_transforms2 = cc(0)
        self.lrp_transforms2 = cc(1)
        self.entropy_bottleneck = tail["entropy_bottleneck"]
        self.gaussian_conditional = tail["gaussian_conditional"]

    def draw_drops(self, B: int, device, generator=None) -> Dict[str, torch.Tensor]:
        out = {}
        for k, r in stf6_drop_path_rates(self.drop_path_rate).items():
            d = _draw_drop(r, B, device, generator)
            if d is not None:
                out[k] = d
        return out

    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("SymmetricalTransFormer3.forward expects [B,3,H,W]")
        if x.shape[2] % 128 or x.shape[3] % 128:
            raise ValueError("SymmetricalTransFormer3: image sides must be multiples of 128 (2 x 2 zigzag blocks of "
                             "4 x 4-window Swin maps at 1/16 resolution)")
        names, params = _named(self)
        dev = x.device
        nz = ny = drops = None
        if self.training:
            nz, ny = _train_noise(self._noise, x, 192, 384)
            B, hb, wb = x.shape[0], x.shape[2] // 32, x.shape[3] // 32
            ny = ny.reshape(B, 24, 64, hb, wb) if ny.dim() == 4 else ny.contiguous()   # iid noise: any layout is the same draw
            drops = self._drops if self._drops is not None else self.draw_drops(B, dev)
            drops = {k: v.to(dev, torch.float32).contiguous() for k, v in drops.items()}
        ws = self.window_size

        def runner(tape, xin, *ps):
            return stf6_forward(tape, dict(zip(names, ps)), xin, nz, ny, drops, ws)

        x_hat, y_lik, z_lik = E.tape_function(runner, [x.contiguous(), *params], self._pack_cache())
        return {"x_hat": x_hat, "likelihoods": {"y": y_lik, "z": z_lik}}

    def update(self, scale_table=None, force=False):
        return self._update_tables(scale_table, force)

    def compress(self, x, _debug=None):
        raise NotImplementedError("stf6: the zigzag entropy coder loop (stf6.py:898-1057) is not mirrored")

    def decompress(self, strings, shape):
        raise NotImplementedError("stf6: the zigzag entropy coder loop (stf6.py:898-1057) is not mirrored")
