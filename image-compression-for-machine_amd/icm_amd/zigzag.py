"""Zigzag block ordering of the stf6 / oj_ICM variants: ``ZigzagSplits`` / ``ZigzagReverse``
(compressai/models/stf6.py:654-714, 716-762; fasterRCNN_ICM.py:103-293) as one HIP index-permutation launch each
(``icm_zigzag_splits`` / ``icm_zigzag_reverse``) instead of 24 Python-level slice + ``contiguous`` + ``cat`` copies.

Same call signatures and results as the reference methods (``zigzag, number, number = ZigzagSplits(inputs, num_slices)``;
``ZigzagReverse(inputs, num_slices, num_H, num_W)``), differentiable (the backward of one is the other applied to the
gradient).  The reference's ``view`` (:664-666) only works when C divides by num_slices and H, W are even; the same
inputs are required here (``ValueError`` otherwise)."""
from __future__ import annotations

import ctypes as C
from typing import List, Tuple

import torch

from . import _lib as L
from ._lib import check, ptr


def zigzag_order(num_slices: int, num_H: int = 2, num_W: int = 2) -> List[Tuple[int, int, int]]:
    """(channel-group, row-half, column-half) index of every output block, in output order"""
    n = L.lib().icm_zigzag_order(num_slices, num_H, num_W, None, 0)
    if n <= 0:
        raise ValueError("zigzag_order: bad block counts")
    buf = (C.c_int32 * n)()
    if L.lib().icm_zigzag_order(num_slices, num_H, num_W, buf, n) != n:
        raise ValueError("zigzag_order failed")
    return [(v // (num_H * num_W), (v // num_W) % num_H, v % num_W) for v in buf]


def _check(Cc, H, W, num_slices, nH, nW):
    if num_slices <= 0 or Cc % num_slices or H % nH or W % nW:
        raise ValueError(f"zigzag: [C={Cc}, H={H}, W={W}] does not split into {num_slices} x {nH} x {nW} blocks")
    if num_slices * nH * nW > 64:
        raise ValueError("zigzag: more than 64 blocks")


def _splits(x: torch.Tensor, ns: int, nH: int, nW: int) -> torch.Tensor:
    B, Cc, H, W = x.shape
    z = torch.empty((B, ns * nH * nW, Cc // ns, H // nH, W // nW), dtype=torch.float32, device=x.device)
    check(L.lib().icm_zigzag_splits(ptr(x), Cc * H * W, ptr(z), B, Cc, H, W, ns, nH, nW, L.stream()), "zigzag_splits")
    return z


def _reverse(z: torch.Tensor, ns: int, nH: int, nW: int) -> torch.Tensor:
    B, N, Cs, Hb, Wb = z.shape
    x = torch.empty((B, Cs * ns, Hb * nH, Wb * nW), dtype=torch.float32, device=z.device)
    check(L.lib().icm_zigzag_reverse(ptr(z), ptr(x), Cs * ns * Hb * nH * Wb * nW, B, Cs * ns, Hb * nH, Wb * nW, ns, nH, nW,
                                     L.stream()), "zigzag_reverse")
    return x


class _SplitFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ns, nH, nW):
        ctx.cfg = (ns, nH, nW)
        return _splits(x.to(torch.float32).contiguous(), ns, nH, nW)

    @staticmethod
    def backward(ctx, g):
        return _reverse(g.contiguous(), *ctx.cfg), None, None, None


class _ReverseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, ns, nH, nW):
        ctx.cfg = (ns, nH, nW)
        return _reverse(z.to(torch.float32).contiguous(), ns, nH, nW)

    @staticmethod
    def backward(ctx, g):
        return _splits(g.contiguous(), *ctx.cfg), None, None, None


def ZigzagSplits(inputs: torch.Tensor, num_slices: int):
    """stf6.py:654-714: [B,C,H,W] -> ([B, num_slices*4, C/num_slices, H/2, W/2], 2, 2)"""
    if inputs.dim() != 4:
        raise ValueError("ZigzagSplits expects [B,C,H,W]")
    number = 2
    _check(inputs.shape[1], inputs.shape[2], inputs.shape[3], num_slices, number, number)
    return _SplitFn.apply(inputs, num_slices, number, number), number, number


def ZigzagReverse(inputs: torch.Tensor, num_slices: int, num_H: int, num_W: int) -> torch.Tensor:
    """stf6.py:716-762: [B, N, Cs, Hb, Wb] -> [B, Cs*num_slices, Hb*num_H, Wb*num_W]"""
    if inputs.dim() != 5 or inputs.shape[1] != num_slices * num_H * num_W:
        raise ValueError("ZigzagReverse expects [B, num_slices*num_H*num_W, C, H, W]")
    _check(inputs.shape[2] * num_slices, inputs.shape[3] * num_H, inputs.shape[4] * num_W, num_slices, num_H, num_W)
    return _ReverseFn.apply(inputs, num_slices, num_H, num_W)
