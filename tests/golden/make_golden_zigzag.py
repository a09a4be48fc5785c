"""Emit tests/golden/zigzag.npz from the REAL reference methods (this container only; needs /root/reference).

``SymmetricalTransFormer3.ZigzagSplits`` / ``ZigzagReverse`` (compressai/models/stf6.py:654-762) use ``self`` for
nothing, so they are called unbound on formula-generated tensors; the fixture stores inputs, outputs and -- through a
tensor whose values encode their own block id -- the block order for several slice counts.
Usage: python tests/golden/make_golden_zigzag.py"""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
from _ref_loader import load_reference  # noqa: E402
from oracle import weights as W  # noqa: E402


def main():
    load_reference()
    stf6 = importlib.import_module("compressai.models.stf6")
    M = stf6.SymmetricalTransFormer3
    out = {}
    cases = [("a", 2, 12, 4, 6, 6), ("b", 1, 8, 6, 2, 2), ("c", 3, 5, 2, 4, 1), ("d", 1, 24, 8, 8, 12), ("e", 2, 6, 2, 2, 3)]
    for tag, B, C, H, Wd, ns in cases:
        x = W._u("zigzag." + tag, (B, C, H, Wd), -4.0, 4.0).float()
        z, nh, nw = M.ZigzagSplits(None, x, ns)
        back = M.ZigzagReverse(None, z, ns, nh, nw)
        assert (nh, nw) == (2, 2) and torch.equal(back, x)
        # block ids: value = (c_blk * 2 + h_blk) * 2 + w_blk, constant inside a block
        ids = torch.zeros(1, ns, 1, 2, 1, 2, 1)
        for c in range(ns):
            for h in range(2):
                for w in range(2):
                    ids[0, c, 0, h, 0, w, 0] = (c * 2 + h) * 2 + w
        ids = ids.expand(1, ns, C // ns, 2, H // 2, 2, Wd // 2).reshape(1, C, H, Wd).contiguous()
        zi, _, _ = M.ZigzagSplits(None, ids, ns)
        order = zi[0, :, 0, 0, 0].to(torch.int32)
        out[tag + "_x"] = x.numpy()
        out[tag + "_z"] = z.numpy()
        out[tag + "_order"] = order.numpy()
        out[tag + "_ns"] = np.int32(ns)
    np.savez_compressed(os.path.join(HERE, "zigzag.npz"), **out)
    print("wrote zigzag.npz:", {k: v.shape for k, v in out.items() if k.endswith("_z")})


if __name__ == "__main__":
    main()
