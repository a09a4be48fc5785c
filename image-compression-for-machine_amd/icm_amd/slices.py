"""Channel-conditional slice loop of the cnn / stf models (compressai/models/cnn.py:154-183 == stf.py:607-637) with the
FIRST layer of every chain split by input-channel block.

The reference builds, per slice i, ``mean_support = cat([latent_means] + y_hat_slices[:max_support])`` and runs
cc_mean / cc_scale on it (and lrp on ``cat([mean_support, y_hat_slice])``).  A convolution is linear in its input
channels, so the first layer of each of the 3 * num_slices chains is

    W[:, :M] * latent  +  W[:, M:M+c*k] * y_hat[:c*k]  (+ W[:, M+c*k:] * y_hat_pre_i for lrp)  + bias,  k = min(i, max_support)

  A  the latent block (M of M + c*k channels: 72 % of the first-layer FLOP at M=320) depends on NO slice: it is computed
     for all chains up front as two wide convolutions (latent_means -> all cc_mean and lrp chains, latent_scales -> all
     cc_scale chains: weights concatenated along GEMM-M, 4 480 / 2 240 output channels instead of 30 launches of 224);
  B  the support block reads y_hat[:, :c*k], which IS the support (no torch.cat copies): one grouped launch per serial
     slice (three chains) and one per batch of tail slices, accumulated onto A in place;
  C  the lrp chains' own-slice block (c = 32 channels) after the Gaussian conditional.

Only B (K = 9*c*k) and C (K = 9*c) remain inside the serial dependency chain slice 0 -> 1 -> ... -> max_support-1.
Backward mirrors it: the input gradients of block A are two contractions over K = 20*224 / 10*224 channels of the
pre-activation gradient buffer, those of block B one contraction per slice over its three chains (blocked channel map),
and the weight gradients land in column blocks of the canonical [Cout][Cin_total][3][3] tensors (``dw_ld``).

Buffers (NCHW f32): PRE0 / G0 / DPRE0 [N, 3*S*D0, h, w] = first-layer pre-activations, their materialised GELU and
their gradients, laid out family-major: channel ((f*S + i)*D0 + d) for family f in (cc_mean, lrp, cc_scale), slice i.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import _lib as L
from . import engine as E
from ._lib import ACT_GELU, EPI_NONE, check, ptr
from .engine import VT, _key

import os as _os
# K-split factors of the wide input-gradient launches (0 / 1 = one launch): same-box A/B switches
KSPLIT_A = int(_os.environ.get("ICM_KSPLIT_A", "4"))
KSPLIT_B = _os.environ.get("ICM_KSPLIT_B", "1") != "0"

FAM = ("cc_mean_transforms", "lrp_transforms", "cc_scale_transforms")   # PRE0 family order: the two latent_means consumers adjacent


def _tail_layers(tape, P, names, xvs, outs=None, lrp_auxs=None):
    """layers 2, 4, 6, 8 of the given chains (first layers done), one grouped launch per layer (cnn.py:89-127)"""
    ts = None
    for li in (2, 4, 6, 8):
        last = li == 8
        ts = E.conv2d_group(tape, xvs, [P[f"{p}.{li}.weight"] for p in names], [P[f"{p}.{li}.bias"] for p in names],
                            pad=1, outs=outs if last else None, lrp_auxs=lrp_auxs if last else None, act_out=not last)
        xvs = [VT(t, ACT_GELU) for t in ts]
    return ts


def _ksplit_dgrad(tape, x, wp, K, Cout, taps, gx, acc, ks, conv_b, tag):
    """Input gradient of a wide first-layer block, x [N, K, h, w] (contiguous channel range) -> gx [N, Cout, h, w], with
    the contraction split into ``ks`` equal channel ranges that run as the members of ONE grouped launch and write
    partial results, which are then added in member order (deterministic).  Such a launch has few output tiles (Cout =
    160 / 320 channels x 4 096 pixels) and a very deep K: on its own it occupies 160 of the 256 CUs; split four ways
    it fills whole rounds of the chip.  wp: the K-concatenated packed weights (chunk-major, so a channel range is a
    pointer offset: taps * cdiv(Cout, 32) * 256 floats per 8 channels)."""
    if ks <= 1 or K % (8 * ks):
        E.conv_launch(tape, x, wp, None, gx, Cin=K, Cout=Cout, transposed=1, accum=acc, tag=tag, **conv_b)
        return
    kp = K // ks
    per8 = taps * ((Cout + 31) // 32) * 256
    N, _, h, w = x.shape
    parts = [E.new((N, Cout, h, w), x.device) for _ in range(ks)]
    E.conv_launch_grouped(tape, [x[:, i * kp:(i + 1) * kp] for i in range(ks)], [wp[(i * kp // 8) * per8:] for i in range(ks)],
                          None, parts, Cin=kp, Cout=Cout, transposed=1, tag=tag, **conv_b)
    for i, part in enumerate(parts):
        if gx.is_contiguous():
            check(L.lib().icm_add_grad(ptr(part), 0, ptr(gx), part.numel(), acc if i == 0 else 1, tape.st), "add_grad")
        else:
            E.copy_into(tape, part, gx, acc if i == 0 else 1)


def _block_grad(tape, w):
    """gradient tensor of a weight that is written in disjoint column blocks: always accumulate into a defined buffer"""
    g, acc = tape.grad_for_write(w)
    if not acc:
        g.zero_()
    return g


def hyper_slices_split(tape: E.Tape, P: Dict[str, torch.Tensor], y, noise_z, noise_y, num_slices: int, max_support: int,
                       keep: Optional[dict], bucket_marks: Optional[dict], codec: Optional[dict], decode: Optional[dict],
                       h_a, h_s_pair, record_symbols):
    """see module docstring; h_a / h_s_pair / record_symbols are models.py's helpers (shared with the unsplit path)"""
    if decode is not None:
        z_hat = decode["z_hat"].contiguous()
        dev, N, M = z_hat.device, z_hat.shape[0], decode["M"]
        h, w = z_hat.shape[2] * 4, z_hat.shape[3] * 4
        if tape.need_grad:
            raise ValueError("decode mode is inference only")
    else:
        dev, N = y.device, y.shape[0]
        M, h, w = y.shape[1], y.shape[2], y.shape[3]
    need = tape.need_grad
    S, ms = num_slices, max_support
    if M % S != 0:
        raise ValueError("latent channels must divide into num_slices")
    c = M // S
    if need:  # y is consumed whole (h_a) and by slices (GaussianConditional): one aliased gradient buffer
        dY = E.zeros(y.shape, dev)
        tape.bind_grad(y, dY, True)
        for i in range(S):
            tape.bind_grad(y[:, i * c:(i + 1) * c], dY[:, i * c:(i + 1) * c], True)
    if bucket_marks is not None:
        bucket_marks[2] = len(tape.bw)   # backward reaching here => hyper-path gradients are complete
    z = z_lik = None
    if decode is None:
        z = h_a(tape, P, y)
        _, z_lik = E.eb_likelihood(tape, z, P, "entropy_bottleneck", noise_z)
        z_hat = E.ste_round_medians(tape, z, P["entropy_bottleneck.quantiles"])
    if (h % 4) or (w % 4):
        raise ValueError("hyper-synthesis output does not match the latent size (input must be a multiple of 64)")
    LM, LSC = E.new((N, M, h, w), dev), E.new((N, M, h, w), dev)       # latent_means, latent_scales (cnn.py:154-155)
    h_s_pair(tape, P, "h_scale_s", "h_mean_s", z_hat, LSC, LM)
    if bucket_marks is not None:
        bucket_marks[1] = len(tape.bw)   # => slice-chain gradients complete

    W0 = {(i, f): P[f"{FAM[f]}.{i}.0.weight"] for i in range(S) for f in range(3)}
    B0 = {(i, f): P[f"{FAM[f]}.{i}.0.bias"] for i in range(S) for f in range(3)}
    D0 = W0[(0, 0)].shape[0]
    KH = W0[(0, 0)].shape[2]
    pad = KH // 2
    for (i, f), wt in W0.items():
        k = min(i, ms)
        if tuple(wt.shape) != (D0, M + c * k + (c if f == 1 else 0), KH, KH):
            raise ValueError(f"{FAM[f]}.{i}.0.weight: unexpected shape {tuple(wt.shape)}")
    if D0 % 32:
        raise ValueError("first-layer width must be a multiple of 32 (concatenated GEMM-M tiles)")
    CT = 3 * S * D0

    def off(i, f):
        return (f * S + i) * D0

    PRE0 = E.new((N, CT, h, w), dev)
    mat = E._MATERIALIZE and N * h * w >= E._MAT_MIN_PIXELS
    G0 = E.new((N, CT, h, w), dev) if mat else None
    pre = {k_: PRE0[:, off(*k_):off(*k_) + D0] for k_ in W0}
    g0 = {k_: G0[:, off(*k_):off(*k_) + D0] for k_ in W0} if mat else None
    DPRE0 = None
    if need:
        DPRE0 = E.new((N, CT, h, w), dev)
        for k_ in W0:
            tape.bind_grad(pre[k_], DPRE0[:, off(*k_):off(*k_) + D0], False)   # first writer: the second layer's dgrad
    Y_hat = E.new((N, M, h, w), dev)
    YP = E.new((N, M, h, w), dev)                   # y_hat before the LRP correction, per slice (cnn.py:171-173)
    MU, SC = E.new((N, M, h, w), dev), E.new((N, M, h, w), dev)
    Y_lik = E.new((N, M, h, w), dev) if decode is None else None
    if need:
        dYh = E.zeros(Y_hat.shape, dev)
        tape.bind_grad(Y_hat, dYh, True)
        for i in range(S):
            tape.bind_grad(Y_hat[:, i * c:(i + 1) * c], dYh[:, i * c:(i + 1) * c], True)
    lib = L.lib()
    conv = dict(KH=KH, KW=KH, stride=1, pad=pad, OH=h, OW=w)
    px = float(N) * h * w

    def algo(cin, cout, members=1):
        """(forward pack flag, input-gradient pack flag, launch argument dicts) of a block launch: Winograd when the
        launch is large enough (engine.wino_ok), else the direct implicit GEMM"""
        on = E.wino_ok(KH, KH, 1, pad, min(cin, cout), work=px * cin * cout * members)
        return (1 if on else 0), (2 if on else 0), dict(conv, algo=1 if on else 0)

    def done(i, f):
        """pre(i, f) is complete: its consumers read the materialised GELU"""
        if mat:
            tape.mat[_key(pre[(i, f)])] = g0[(i, f)]

    # ---------------------------------------------------------------------------------------------- A: latent blocks
    order = [(i, f) for f in range(3) for i in range(S)]                # PRE0 channel order
    bias0 = E.new((CT,), dev)
    srcs = (C.c_void_p * len(order))(*[ptr(B0[k_]) for k_ in order])
    check(lib.icm_gather_vectors(srcs, len(order), D0, ptr(bias0), tape.st), "gather_vectors")
    # slice 0's mean / scale chains have no other block: their first layer is complete (and materialised) here
    wf, _, conv_f = algo(M, D0, 2)
    wp0 = [tape.pack_cat([(W0[(0, f)], 0)], D0, M, KH, KH, 1, 0, 1, pad, "M", wino=wf) for f in (0, 2)]
    E.conv_launch_grouped(tape, [LM, LSC], wp0, [B0[(0, 0)], B0[(0, 2)]], [pre[(0, 0)], pre[(0, 2)]], Cin=M, Cout=D0,
                          transposed=0, y2s=[g0[(0, 0)], g0[(0, 2)]] if mat else None, tag="fwd(A0)", **conv_f)
    done(0, 0)
    done(0, 2)
    # everything else: latent_means -> channels [D0, 2*S*D0), latent_scales -> [2*S*D0 + D0, 3*S*D0)
    mem_m = [k_ for k_ in order if k_[1] in (0, 1)][1:]
    mem_s = [k_ for k_ in order if k_[1] == 2][1:]
    for xin, mem in ((LM, mem_m), (LSC, mem_s)):
        c0 = off(*mem[0])
        wf, _, conv_f = algo(M, len(mem) * D0)
        wpA = tape.pack_cat([(W0[k_], 0) for k_ in mem], D0, M, KH, KH, 1, 0, 1, pad, "M", wino=wf)
        E.conv_launch(tape, xin, wpA, bias0[c0:c0 + len(mem) * D0], PRE0[:, c0:c0 + len(mem) * D0], Cin=M,
                      Cout=len(mem) * D0, transposed=0, tag="fwd(A)", **conv_f)
    if need:
        def bwd_A():
            # weight (+ bias) gradients of the latent blocks: 3*S problems of one geometry, columns [0, M) of each weight
            for k_ in order:
                gw = _block_grad(tape, W0[k_])
                gb_, accb = tape.grad_for_write(B0[k_])
                E.wgrad_defer(tape, DPRE0[:, off(*k_):off(*k_) + D0], LM if k_[1] != 2 else LSC, gw[:, :M], Ca=D0, Cb=M,
                              KH=KH, KW=KH, stride=1, pad=pad, accum=1, dbias=gb_, accum_bias=accb, dw_ld=W0[k_].shape[1])
            # input gradients: ONE contraction per latent tensor over all of its consumers' channels
            for xin, mem in ((LM, [k_ for k_ in order if k_[1] in (0, 1)]), (LSC, [k_ for k_ in order if k_[1] == 2])):
                if not tape.wants(xin):
                    continue
                c0 = off(*mem[0])
                _, wbk, conv_b = algo(len(mem) * D0, M)
                wpb = tape.pack_cat([(W0[k_], 0) for k_ in mem], M, D0, KH, KH, 0, 1, 1, pad, "K", wino=wbk)
                gx, acc = tape.grad_for_write(xin)
                _ksplit_dgrad(tape, DPRE0[:, c0:c0 + len(mem) * D0], wpb, len(mem) * D0, M, 16 if wbk else KH * KH, gx, acc,
                              KSPLIT_A, conv_b, "dgrad(A)")
        tape.bw.append(bwd_A)

    # ---------------------------------------------------------------------------------------------- B / C helpers
    def support_block(idx, k):
        """B: y_hat[:, :c*k] (the support of slices idx, contiguous, all with the same k > 0) into the first layers of
        their three chains, accumulated in place; one member per family (weights of the slices concatenated along M)"""
        n = len(idx)
        sup = Y_hat[:, :c * k]
        r0 = [off(idx[0], f) for f in range(3)]
        wf, _, conv_f = algo(c * k, n * D0, 3)
        wps = [tape.pack_cat([(W0[(i, f)], M) for i in idx], D0, c * k, KH, KH, 1, 0, 1, pad, "M", wino=wf) for f in range(3)]
        ys = [PRE0[:, r:r + n * D0] for r in r0]
        y2s = [G0[:, r:r + n * D0] for r in r0] if mat else None
        E.conv_launch_grouped(tape, [sup] * 3, wps, None, ys, Cin=c * k, Cout=n * D0, transposed=0, y2s=y2s, accum=1,
                              tag="fwd(B)", **conv_f)
        for i in idx:
            done(i, 0)
            done(i, 2)
        if need:
            def bwd_B():
                for i in idx:
                    for f in range(3):
                        gw = _block_grad(tape, W0[(i, f)])
                        E.wgrad_defer(tape, DPRE0[:, off(i, f):off(i, f) + D0], sup, gw[:, M:M + c * k], Ca=D0, Cb=c * k,
                                      KH=KH, KW=KH, stride=1, pad=pad, accum=1, dw_ld=W0[(i, f)].shape[1])
                # d y_hat[:, :c*k] += sum over the 3*n chains: one contraction, channels (f, i) picked by the blocked map
                _, wbk, conv_b = algo(3 * n * D0, c * k)
                wpb = tape.pack_cat([(W0[(i, f)], M) for f in range(3) for i in idx], c * k, D0, KH, KH, 0, 1, 1, pad, "K",
                                    wino=wbk)
                gx, acc = tape.grad_for_write(sup)
                if n > 1 and KSPLIT_B:
                    # batch of tail slices: one member per family (each a contiguous channel run: no blocked map needed),
                    # partial sums added in family order
                    per8 = (16 if wbk else KH * KH) * ((c * k + 31) // 32) * 256
                    parts = [E.new((N, c * k, h, w), dev) for _ in range(3)]
                    E.conv_launch_grouped(tape, [DPRE0[:, r:r + n * D0] for r in r0],
                                          [wpb[(f * n * D0 // 8) * per8:] for f in range(3)], None, parts, Cin=n * D0,
                                          Cout=c * k, transposed=1, tag="dgrad(B)", **conv_b)
                    for f, part in enumerate(parts):
                        E.copy_into(tape, part, gx, acc if f == 0 else 1)
                else:
                    E.conv_launch(tape, DPRE0[:, r0[0]:], wpb, None, gx, Cin=3 * n * D0, Cout=c * k, transposed=1,
                                  accum=acc, seg=(n * D0, (S - n) * D0), tag="dgrad(B)", **conv_b)
            tape.bind_grad(sup, dYh[:, :c * k], True)
            tape.bw.append(bwd_B)

    def own_block(idx, k):
        """C: y_hat_pre of slice i (c channels) into the first layer of its lrp chain (cnn.py:174), completing it"""
        xs = [YP[:, i * c:(i + 1) * c] for i in idx]
        wf, _, conv_f = algo(c, D0, len(idx))
        wps = [tape.pack_cat([(W0[(i, 1)], M + c * k)], D0, c, KH, KH, 1, 0, 1, pad, "M", wino=wf) for i in idx]
        E.conv_launch_grouped(tape, xs, wps, None, [pre[(i, 1)] for i in idx], Cin=c, Cout=D0, transposed=0,
                              y2s=[g0[(i, 1)] for i in idx] if mat else None, accum=1, tag="fwd(C)", **conv_f)
        for i in idx:
            done(i, 1)
        if need:
            def bwd_C():
                gxs = []
                for i, x in zip(idx, xs):
                    gw = _block_grad(tape, W0[(i, 1)])
                    E.wgrad_defer(tape, DPRE0[:, off(i, 1):off(i, 1) + D0], x, gw[:, M + c * k:M + c * k + c], Ca=D0, Cb=c,
                                  KH=KH, KW=KH, stride=1, pad=pad, accum=1, dw_ld=W0[(i, 1)].shape[1])
                    gx, acc = tape.grad_for_write(x)
                    if not acc:          # (the LRP tail's identity path wrote it first; a launch has one accumulate flag)
                        gx.zero_()
                    gxs.append(gx)
                _, wbk, conv_b = algo(D0, c, len(idx))
                wpb = [tape.pack_cat([(W0[(i, 1)], M + c * k)], c, D0, KH, KH, 0, 1, 1, pad, "K", wino=wbk) for i in idx]
                E.conv_launch_grouped(tape, [DPRE0[:, off(i, 1):off(i, 1) + D0] for i in idx], wpb, None, gxs, Cin=D0,
                                      Cout=c, transposed=1, accum=1, tag="dgrad(C)", **conv_b)
            tape.bw.append(bwd_C)

    def gauss(idx):
        """GaussianConditional + ste_round per slice (cnn.py:170-173): y_hat_pre -> YP, likelihoods -> Y_lik"""
        for i in idx:
            s1 = slice(i * c, (i + 1) * c)
            if decode is not None:
                decode["slice"](i, MU[:, s1], SC[:, s1], YP[:, s1])
            else:
                E.gc_likelihood_ste(tape, y[:, s1], MU[:, s1], SC[:, s1], None if noise_y is None else noise_y[:, s1],
                                    Y_lik[:, s1], YP[:, s1])
            if codec is not None:
                record_symbols(codec, i, y[:, s1], MU[:, s1], SC[:, s1])

    def chains(idx, k):
        if k > 0:
            support_block(idx, k)
        # layers 2..8 of the mean / scale chains of these slices, straight into the mu / scale buffers
        names = [f"cc_mean_transforms.{i}" for i in idx] + [f"cc_scale_transforms.{i}" for i in idx]
        xvs = [VT(pre[(i, 0)], ACT_GELU) for i in idx] + [VT(pre[(i, 2)], ACT_GELU) for i in idx]
        outs = [MU[:, i * c:(i + 1) * c] for i in idx] + [SC[:, i * c:(i + 1) * c] for i in idx]
        _tail_layers(tape, P, names, xvs, outs=outs)
        gauss(idx)
        own_block(idx, k)
        lnames = [f"lrp_transforms.{i}" for i in idx]
        _tail_layers(tape, P, lnames, [VT(pre[(i, 1)], ACT_GELU) for i in idx],
                     outs=[Y_hat[:, i * c:(i + 1) * c] for i in idx], lrp_auxs=[YP[:, i * c:(i + 1) * c] for i in idx])

    # ---- serial slices 0 .. max_support-1, then the tail slices (all with the same, complete support) in batches
    n_serial = min(S, ms)
    for i in range(n_serial):
        chains([i], i)
    tail = list(range(n_serial, S))
    step = max(1, E.MAX_GROUP // 2)
    for t0 in range(0, len(tail), step):
        chains(tail[t0:t0 + step], ms)
    if bucket_marks is not None:
        bucket_marks[0] = len(tape.bw)   # => synthesis-transform gradients complete
    if keep is not None:
        keep.update(y=y, z=z, z_hat=z_hat, y_hat=Y_hat, mu=MU, scale=SC, lat_means=LM, lat_scales=LSC)
    return Y_hat, Y_lik, z_lik
