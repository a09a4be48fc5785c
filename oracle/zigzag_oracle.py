"""CPU ORACLE for the zigzag block ordering of the stf6 / oj_ICM variants -- TEST INFRASTRUCTURE ONLY.

numpy restatement of ``ZigzagSplits`` / ``ZigzagReverse`` (compressai/models/stf6.py:654-714, 716-762; the same two
methods appear in fasterRCNN_ICM.py:103-293).  The latent [B, C, H, W] is cut into ``num_slices`` channel groups x
``nH`` row halves x ``nW`` column halves (contiguous blocks: the reference ``view``s the tensor as
[B, ns, C/ns, nH, H/nH, nW, W/nW], :664-666); the blocks are emitted shell by shell -- shell i holds the blocks whose
largest index is i -- and inside a shell with the channel-group index running fastest, then the row-half index, then the
column-half index (:671-696).  Pinned by tests/golden/zigzag.npz (emitted from the real methods by
tests/golden/make_golden_zigzag.py).  Only tests/ import this module."""
from __future__ import annotations

from typing import List, Tuple

import numpy as np


def zigzag_order(num_slices: int, num_h: int = 2, num_w: int = 2) -> List[Tuple[int, int, int]]:
    """(channel-group, row-half, column-half) of every output block, in output order (stf6.py:671-696)"""
    order: List[Tuple[int, int, int]] = []
    for i in range(max(num_slices, num_h, num_w)):
        c = h = w = 0
        for _ in range(min(i + 1, num_slices) * min(i + 1, num_h) * min(i + 1, num_w)):
            if not (max(c, h, w) < i and i > 0):
                order.append((c, h, w))
            # odometer step: channel group fastest, then row half, then column half, each bounded by the shell index
            if c + 2 > num_slices or c + 1 > i:
                c = 0
                if h + 2 > num_h or h + 1 > i:
                    w += 1
                    h = 0
                else:
                    h += 1
            else:
                c += 1
    return order


def zigzag_splits(x: np.ndarray, num_slices: int, number: int = 2) -> np.ndarray:
    """[B, C, H, W] -> [B, num_slices * number^2, C / num_slices, H / number, W / number] (stf6.py:654-714)"""
    B, C, H, W = x.shape
    if C % num_slices or H % number or W % number:
        raise ValueError("zigzag_splits: C must divide by num_slices and H, W by the block count")
    v = x.reshape(B, num_slices, C // num_slices, number, H // number, number, W // number)
    return np.stack([v[:, c, :, h, :, w, :] for (c, h, w) in zigzag_order(num_slices, number, number)], axis=1)


def zigzag_reverse(z: np.ndarray, num_slices: int, num_h: int = 2, num_w: int = 2) -> np.ndarray:
    """[B, N, Cs, Hb, Wb] -> [B, Cs * num_slices, Hb * num_h, Wb * num_w] (stf6.py:716-762)"""
    B, N, Cs, Hb, Wb = z.shape
    order = zigzag_order(num_slices, num_h, num_w)
    if N != len(order):
        raise ValueError("zigzag_reverse: block count mismatch")
    out = np.zeros((B, num_slices, Cs, num_h, Hb, num_w, Wb), dtype=z.dtype)
    for n, (c, h, w) in enumerate(order):
        out[:, c, :, h, :, w, :] = z[:, n]
    return out.reshape(B, Cs * num_slices, Hb * num_h, Wb * num_w)
