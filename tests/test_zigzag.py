"""Zigzag block ordering (compressai/models/stf6.py:654-762).  CPU: the oracle and the C-ABI order function
against the fixture emitted from the real reference methods (tests/golden/make_golden_zigzag.py).  GPU: the HIP
permutation against fixture and oracle, round trips at full latent size, gradients, error behaviour."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "image-compression-for-machine_amd"))
from oracle import zigzag_oracle as ZO  # noqa: E402

CASES = ["a", "b", "c", "d", "e"]


def _fx(golden_dir):
    return np.load(os.path.join(golden_dir, "zigzag.npz"), allow_pickle=False)


@pytest.mark.parametrize("tag", CASES)
def test_oracle_matches_reference_fixture(golden_dir, tag):
    f = _fx(golden_dir)
    x, z, ns = f[tag + "_x"], f[tag + "_z"], int(f[tag + "_ns"])
    assert np.array_equal(ZO.zigzag_splits(x, ns), z)
    assert np.array_equal(ZO.zigzag_reverse(z, ns), x)
    order = [(c * 2 + h) * 2 + w for (c, h, w) in ZO.zigzag_order(ns, 2, 2)]
    assert order == f[tag + "_order"].tolist()
    assert sorted(order) == list(range(ns * 4))      # a permutation of all blocks


def test_c_abi_order_matches_oracle():
    from icm_amd import _lib
    lib = _lib.lib()
    for ns, nh, nw in [(6, 2, 2), (1, 2, 2), (2, 2, 2), (3, 2, 2), (12, 2, 2), (16, 2, 2), (4, 1, 3), (2, 3, 1), (5, 4, 2)]:
        n = lib.icm_zigzag_order(ns, nh, nw, None, 0)
        assert n == ns * nh * nw
        buf = (ctypes.c_int32 * n)()
        assert lib.icm_zigzag_order(ns, nh, nw, buf, n) == n
        assert list(buf) == [(c * nh + h) * nw + w for (c, h, w) in ZO.zigzag_order(ns, nh, nw)]
        assert lib.icm_zigzag_order(ns, nh, nw, buf, n - 1) == -1
    assert lib.icm_zigzag_order(0, 2, 2, None, 0) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_hip_zigzag_matches_fixture(golden_dir, tag):
    from icm_amd import zigzag as Z
    f = _fx(golden_dir)
    x, z, ns = torch.from_numpy(f[tag + "_x"]).cuda(), torch.from_numpy(f[tag + "_z"]), int(f[tag + "_ns"])
    got, nh, nw = Z.ZigzagSplits(x, ns)
    assert (nh, nw) == (2, 2) and torch.equal(got.cpu(), z)
    assert torch.equal(Z.ZigzagReverse(got, ns, nh, nw), x)
    assert [(c * 2 + h) * 2 + w for (c, h, w) in Z.zigzag_order(ns)] == f[tag + "_order"].tolist()


@pytest.mark.gpu
def test_hip_zigzag_full_size_roundtrip_and_gradients():
    from icm_amd import zigzag as Z
    torch.manual_seed(0)
    x = torch.randn(16, 384, 16, 16, device="cuda", requires_grad=True)        # the stf6 latent at B=16, 256x256
    z, nh, nw = Z.ZigzagSplits(x, 6)
    assert tuple(z.shape) == (16, 24, 64, 8, 8)
    ref = torch.from_numpy(ZO.zigzag_splits(x.detach().cpu().numpy(), 6))
    assert torch.equal(z.detach().cpu(), ref)
    back = Z.ZigzagReverse(z, 6, nh, nw)
    assert torch.equal(back.detach(), x.detach())                              # size-independent property: exact inverse
    # adjoint: <splits(x), g> == <x, reverse(g)> exactly (a permutation) -> the gradient is reverse(g)
    g = torch.randn_like(z)
    gx, = torch.autograd.grad(z, x, g)
    assert torch.equal(gx, Z.ZigzagReverse(g, 6, 2, 2).detach())
    z2 = z.detach().clone().requires_grad_(True)
    gz, = torch.autograd.grad(Z.ZigzagReverse(z2, 6, 2, 2), z2, x.detach())
    assert torch.equal(gz, z.detach())


@pytest.mark.gpu
def test_hip_zigzag_argument_errors():
    from icm_amd import zigzag as Z
    with pytest.raises(ValueError):
        Z.ZigzagSplits(torch.zeros(1, 10, 4, 4, device="cuda"), 6)             # C % num_slices
    with pytest.raises(ValueError):
        Z.ZigzagSplits(torch.zeros(1, 12, 5, 4, device="cuda"), 6)             # odd H: the reference's view() fails too
    with pytest.raises(ValueError):
        Z.ZigzagReverse(torch.zeros(1, 23, 2, 2, 2, device="cuda"), 6, 2, 2)   # wrong block count
    with pytest.raises(ValueError):
        Z.ZigzagSplits(torch.zeros(1, 34, 2, 2, device="cuda"), 17)            # > 64 blocks
