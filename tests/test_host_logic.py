"""CPU: host-side logic of the product path with kernel launches stubbed out.

The C-ABI library must load and export every symbol of include/icm_hip.h; its pure-host entry points
(packed-weight sizes, wgrad workspace planning, argument validation) are called for real.  The Python
engine (tape bookkeeping, gradient aliasing of the support buffers, state-dict layout) is then driven
end-to-end with launches replaced by no-ops: every parameter must receive a gradient buffer of its own
shape and the forward contract must hold.  No numerics are checked here (that is the -m gpu suite)."""
import ctypes
import os
import re

import pytest
import torch

from icm_amd import _lib


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "libicm_hip.so not built (run __graft_entry__.build())"
    L = ctypes.CDLL(_lib.LIB_PATH)
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "icm_hip.h")).read()
    declared = set(re.findall(r"\b(icm_[a-zA-Z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert set(_lib.SYMBOLS) <= declared | {"icm_conv_run_grouped"}
    L.icm_strerror.restype = ctypes.c_char_p
    assert L.icm_strerror(0) == b"ok" and L.icm_version() >= 1


def test_host_planning_entry_points():
    L = _lib.lib()
    assert L.icm_packed_weight_floats(192, 192, 5, 5) == 24 * 25 * 6 * 256
    assert L.icm_packed_weight_floats(3, 192, 5, 5) == 24 * 25 * 1 * 256
    a = _lib.WgradArgs()
    a.gs, a.gb = 1, 1   # never dereferenced by the planner
    a.Ca, a.OH, a.OW, a.Cb, a.H, a.W, a.N = 192, 64, 64, 192, 128, 128, 16
    a.KH = a.KW = 5
    a.stride, a.pad = 2, 2
    n = L.icm_wgrad_workspace_floats(ctypes.byref(a))
    assert n > 0 and n % 192 == 0
    # the pixel-split count (hence the slab size) depends on how many problems share a grouped launch
    b = _lib.WgradArgs()
    b.gs, b.gb = 1, 1
    b.Ca, b.OH, b.OW, b.Cb, b.H, b.W, b.N = 1536, 16, 16, 384, 16, 16, 16
    b.KH = b.KW = 1
    b.stride, b.pad = 1, 0
    sizes = {k: L.icm_wgrad_workspace_floats_grouped(ctypes.byref(b), k) for k in (1, 6)}
    assert sizes[1] == L.icm_wgrad_workspace_floats(ctypes.byref(b)) and sizes[6] != sizes[1]
    assert all(v > 0 and v % (1536 * 384 + 1536) == 0 for v in sizes.values())   # whole slabs (+ bias partials) per split
    # a workspace that is too small for the launch's split count is refused before anything is launched
    b.dw, b.ws, b.ws_floats = 1, 1, sizes[6] - 64
    arr = (_lib.WgradArgs * 6)(*[b] * 6)
    assert L.icm_conv_wgrad_grouped(arr, 6, None) == 1
    a.OH = 63  # inconsistent geometry -> rejected like a shape error
    assert L.icm_wgrad_workspace_floats(ctypes.byref(a)) == -1
    # argument validation happens before any launch: NULL pointers / bad stride are refused on CPU too
    c = _lib.ConvArgs()
    assert L.icm_conv_run(ctypes.byref(c), None) == 1
    with pytest.raises(ValueError):
        _lib.check(1, "x")
    with pytest.raises(_lib.IcmError):
        _lib.check(2, "x")


class _FakeLib:
    """real host-side planners, no-op launches"""

    def __init__(self, real):
        self._real = real
        self.calls = {}

    def __getattr__(self, name):
        if name in ("icm_packed_weight_floats", "icm_wgrad_workspace_floats", "icm_wgrad_workspace_floats_grouped",
                    "icm_strerror"):
            return getattr(self._real, name)

        def f(*a):
            self.calls[name] = self.calls.get(name, 0) + 1
            return 0
        return f


@pytest.fixture()
def dry(monkeypatch):
    real = _lib.lib()
    fake = _FakeLib(real)
    monkeypatch.setattr(_lib, "lib", lambda: fake)
    monkeypatch.setattr(_lib, "stream", lambda: 0)
    orig_bs = _lib.bs

    def bs(t):
        if t is None:
            return 0
        if t.dim() == 4:
            st = t.stride()
            N, C, H, W = t.shape
            assert (W == 1 or st[3] == 1) and (H == 1 or st[2] == W) and (C == 1 or st[1] == H * W)
            return st[0]
        return t[0].numel()
    import icm_amd.engine as E
    monkeypatch.setattr(_lib, "bs", bs)
    monkeypatch.setattr(E, "bs", bs)
    return fake


@pytest.mark.parametrize("slice_split", [False, True], ids=["default", "split_slices"])
def test_wacnn_tape_plumbing_dry_run(dry, slice_split, monkeypatch):
    from icm_amd.zoo import models
    from icm_amd.losses import RateDistortionLoss
    from icm_amd import models as M_
    monkeypatch.setattr(M_, "SLICE_SPLIT", slice_split)
    torch.manual_seed(0)
    net = models["cnn"]().train()
    x = torch.rand(2, 3, 64, 64)
    out = net(x)
    assert set(out) == {"x_hat", "likelihoods"} and set(out["likelihoods"]) == {"y", "z"}
    assert out["x_hat"].shape == (2, 3, 64, 64)
    assert out["likelihoods"]["y"].shape == (2, 320, 4, 4) and out["likelihoods"]["z"].shape == (2, 192, 1, 1)
    loss = out["x_hat"].sum() + out["likelihoods"]["y"].sum() + out["likelihoods"]["z"].sum()
    loss.backward()
    missing = [n for n, p in net.named_parameters() if p.grad is None and not n.endswith("quantiles")]
    assert not missing, missing[:10]
    for n, p in net.named_parameters():
        if p.grad is not None:
            assert p.grad.shape == p.shape, n
    if slice_split:
        # grouped launches of the slice section (icm_amd/slices.py), forward: slice 0's latent blocks (1), 4 support blocks
        # and 5 own-slice blocks of the serial slices, their second..fifth layers (5 x (4 + 4)), and for the batch of tail
        # slices 5..9 one support block, one own-slice block and 4 + 4 layers; backward: the same second..fifth layers, the
        # own-slice blocks and 3 K-split wide input gradients (the serial slices' support input gradients are single
        # launches).  Plus the 4 gates (3 ResidualUnit steps x 3 convs of the two branches paired) and the h_scale_s /
        # h_mean_s pair, forward and backward.
        assert dry.calls["icm_conv_run_grouped"] == (1 + 4 + 5 + 40 + 10) + (40 + 8 + 6) + 3 + 2 * (4 * 9 + 5)
        assert dry.calls["icm_conv_run"] >= 60 and dry.calls["icm_gather_vectors"] == 1
    else:
        # grouped launches (forward and dgrad): the mean/scale pair of the 5 serial slices (5 x 5 convs), then the 10
        # mean/scale chains and the 5 lrp chains of the independent tail slices 5..9 as one chain each (5 + 5 layers);
        # plus the 4 gates and the h_scale_s / h_mean_s pair
        assert dry.calls["icm_conv_run"] > 100 and dry.calls["icm_conv_run_grouped"] == 2 * (25 + 5 + 5 + 4 * 9 + 5)
    assert 20 <= dry.calls["icm_conv_wgrad_grouped"] <= 80 and dry.calls.get("icm_conv_wgrad", 0) == 6
    assert dry.calls["icm_gc_likelihood_ste_fwd"] == 10 and dry.calls["icm_gc_likelihood_ste_bwd"] == 10
    assert dry.calls["icm_winattn_fwd"] == 4 and dry.calls["icm_winattn_bwd"] == 4


def test_eval_mode_has_no_tape(dry):
    from icm_amd.zoo import models
    net = models["cnn"]().eval()
    with torch.no_grad():
        out = net(torch.rand(1, 3, 64, 64))
    assert not out["x_hat"].requires_grad
    assert "icm_conv_wgrad" not in dry.calls


def test_reference_error_behaviour(dry):
    from icm_amd import layers
    from icm_amd.entropy_models import EntropyBottleneck
    with pytest.raises(AssertionError):
        layers.WinBasedAttention(dim=64, num_heads=8, window_size=4, shift_size=4)
    with pytest.raises(ValueError):
        EntropyBottleneck(8).quantize(torch.zeros(1), "bogus")
    with pytest.raises(ValueError):
        layers.Conv2d(8, 8, 3, padding=1)(torch.zeros(1, 4, 8, 8))


def test_stf_tape_plumbing_dry_run(dry):
    from icm_amd.zoo import models
    torch.manual_seed(0)
    net = models["stf"]().train()
    x = torch.rand(2, 3, 64, 64)
    out = net(x)
    assert out["x_hat"].shape == (2, 3, 64, 64)
    assert out["likelihoods"]["y"].shape == (2, 384, 4, 4) and out["likelihoods"]["z"].shape == (2, 192, 1, 1)
    loss = out["x_hat"].sum() + out["likelihoods"]["y"].sum() + out["likelihoods"]["z"].sum()
    loss.backward()
    missing = [n for n, p in net.named_parameters() if p.grad is None and not n.endswith("quantiles")]
    assert not missing, missing[:10]
    for n, p in net.named_parameters():
        if p.grad is not None:
            assert p.grad.shape == p.shape, n
    # 24 Swin blocks: one attention core each, two LayerNorms each (+ patch_embed + 3 merges + 3 splits)
    assert dry.calls["icm_winattn_fwd"] == 24 and dry.calls["icm_winattn_bwd"] == 24
    assert dry.calls["icm_layernorm_fwd"] == 24 * 2 + 1 + 6 and dry.calls["icm_layernorm_bwd"] == 24 * 2 + 1 + 6
    assert dry.calls["icm_space_to_depth2"] == 6            # 3 PatchMerging gathers forward + 3 scatters backward
    assert dry.calls["icm_gc_likelihood_ste_fwd"] == 12
    # the first analysis and first synthesis block have drop-path rate 0 (fused residual epilogues); the other 22
    # draw per-sample scales: 2 branches each, forward + backward
    assert dry.calls["icm_residual_scale"] == 22 * 2 * 2


def test_stf_rejects_non_default_architecture():
    from icm_amd.models import SymmetricalTransFormer
    with pytest.raises(NotImplementedError):
        SymmetricalTransFormer(embed_dim=96)


@pytest.mark.parametrize("name", ["cnn", "stf"])
def test_trainer_step_plumbing_dry_run(dry, name):
    """two native training steps with launches stubbed: bucket markers, deferred weight gradients, the recorded
    weight-packing sequence (step 1) and its windowed replay (step 2), clip + 2 x Adam + aux step all get issued"""
    from icm_amd.zoo import models
    from icm_amd.trainer import Trainer, BUCKETS
    torch.manual_seed(0)
    tr = Trainer(models[name](), device="cpu")
    assert tr.side is None and len(tr.flat.bucket_ranges) == len(BUCKETS)
    x = torch.rand(2, 3, 64, 64)
    s1 = tr.step(x)
    assert s1.shape == (8,) and tr._pack_seq is not None and len(tr._pack_seq) > 100
    # step 1 packs entry by entry (one batch call per cache miss) and records the miss sequence ...
    first = dry.calls["icm_pack_weights_batch"]
    # (stf: + the two on-demand packs -- forward and input-gradient orientation -- of the thin-output temporary of
    # end_conv[2], engine.conv2d_thin_out: a per-step tensor is never packed ahead of time from a recorded sequence)
    temps = 2 if name == "stf" else 0
    assert first == len(tr._pack_seq) + temps and "icm_pack_weights" not in dry.calls
    tr.step(x)
    # ... step 2 replays it in windows of 24 entries
    assert dry.calls["icm_pack_weights_batch"] - first == -(-len(tr._pack_seq) // 24) + temps
    assert dry.calls["icm_adam_step"] == 4 and dry.calls["icm_grad_sqnorm"] == 2 and dry.calls["icm_eb_aux_loss"] == 2
    assert dry.calls["icm_rd_loss_fwd"] == 2 and dry.calls["icm_conv_wgrad_grouped"] > 40
    if name == "stf":   # thin-output end_conv[2]: W -> W' and dW' -> dW once per step, its gradient computed at once
        assert dry.calls.get("icm_permute_flip") == 4 and dry.calls.get("icm_conv_wgrad") == 2
