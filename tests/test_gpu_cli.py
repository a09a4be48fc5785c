"""MI355X: the callers either side of the hot path end to end -- ``python -m icm_amd.train`` semantics
(train.py:380-530) and ``python -m icm_amd.eval_model`` (eval_model/__main__.py:627-671) on small image folders."""
import json
import os
import sys

import numpy as np
import pytest
import torch
from PIL import Image

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "image-compression-for-machine_amd"))

pytestmark = pytest.mark.gpu


def _write(folder, sizes, seed=0):
    os.makedirs(folder, exist_ok=True)
    rng = np.random.default_rng(seed)
    for i, (h, w) in enumerate(sizes):
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([(yy * 3 + xx * 2 + 40 * c) % 256 for c in range(3)], -1)
        a = np.clip(base + rng.integers(-12, 13, size=(h, w, 3)), 0, 255).astype(np.uint8)
        Image.fromarray(a).save(os.path.join(folder, f"im{i:02d}.png"))


def test_eval_cli_matches_direct_calls(tmp_path, capsys):
    from icm_amd import eval_model as EM
    from icm_amd import utils as U
    from icm_amd.zoo import models
    folder = str(tmp_path / "val")
    _write(folder, [(96, 80), (64, 64), (70, 130)])
    torch.manual_seed(3)
    net = models["cnn"]()
    ck = str(tmp_path / "w.ckpt")
    torch.save({"epoch": 0, "state_dict": net.state_dict()}, ck)
    rec = str(tmp_path / "rec")

    reports = {}
    for mode in ("coder", "estimate"):
        argv = ["-d", folder, "-a", "cnn", "-p", ck, "-r", rec if mode == "coder" else ""]
        if mode == "estimate":
            argv.append("--entropy-estimation")
        assert EM.main(argv) == 0
        rep = json.loads(capsys.readouterr().out)
        assert rep["name"] == "cnn"
        assert rep["description"] == ("Inference (ans)" if mode == "coder" else "Inference (entropy estimation)")
        assert set(rep["results"]) == {"psnr", "bpp", "encoding_time", "decoding_time"}
        reports[mode] = {k: v[0] for k, v in rep["results"].items()}

    # the same numbers from direct calls on the same checkpoint (eval_model/__main__.py:472-487: mean over the folder)
    model = EM.load_checkpoint("cnn", ck).to("cuda")
    model.update(force=True)
    files = EM.collect_images(folder)
    direct = [U.inference(model, EM.read_image(f).to("cuda")) for f in files]
    est = [U.inference_entropy_estimation(model, EM.read_image(f).to("cuda")) for f in files]
    assert reports["coder"]["bpp"] == pytest.approx(np.mean([d["bpp"] for d in direct]), rel=1e-12)
    assert reports["coder"]["psnr"] == pytest.approx(np.mean([d["psnr"] for d in direct]), rel=1e-6)
    assert reports["estimate"]["bpp"] == pytest.approx(np.mean([d["bpp"] for d in est]), rel=1e-5)
    assert reports["estimate"]["psnr"] == pytest.approx(np.mean([d["psnr"] for d in est]), rel=1e-6)
    # reconstructions: one file per image, original (un-padded) size, equal to the decoder output quantised to 8 bits
    for f in files:
        r = np.asarray(Image.open(os.path.join(rec, os.path.basename(f))))
        x = EM.read_image(f)
        assert r.shape == (x.shape[1], x.shape[2], 3)
    xp, pads = U.pad_to_multiple(EM.read_image(files[0]).unsqueeze(0).to("cuda"), 64)
    enc = model.compress(xp)
    xh = U.crop(model.decompress(enc["strings"], enc["shape"])["x_hat"], pads).clamp(0, 1)[0]
    want = (xh.cpu() * 255.0).to(torch.uint8).numpy().transpose(1, 2, 0)
    got = np.asarray(Image.open(os.path.join(rec, os.path.basename(files[0]))))
    assert np.array_equal(got, want)


def test_train_cli_runs_saves_and_resumes(tmp_path, capsys):
    from icm_amd import eval_model as EM
    from icm_amd import train as T
    root = str(tmp_path / "data")
    _write(os.path.join(root, "train"), [(80, 72)] * 6, seed=1)
    _write(os.path.join(root, "test"), [(64, 64)] * 2, seed=2)
    save = str(tmp_path / "ck") + os.sep
    common = ["-d", root, "--batch-size", "2", "--test-batch-size", "2", "--patch-size", "64", "64", "-n", "0",
              "--seed", "7", "--save", "--save_path", save, "--test-every", "1", "-lr", "1e-4"]
    assert T.main(common + ["-e", "1"]) == 0
    out = capsys.readouterr().out
    assert "Train epoch 0: [0/6" in out and "Test epoch 0: Average losses:" in out and "Learning rate: 0.0001" in out
    ck_path = os.path.join(save, "0.ckpt")
    ck = torch.load(ck_path, map_location="cpu", weights_only=True)
    assert {"epoch", "state_dict", "loss", "optimizer", "aux_optimizer", "lr_scheduler"} <= set(ck)
    assert ck["epoch"] == 0 and ck["optimizer"]["step"] == 3 and len(ck["state_dict"]) == 585
    assert all(torch.isfinite(v).all() for v in ck["state_dict"].values() if v.dtype.is_floating_point)
    m0 = ck["optimizer"]["m"]["g_a.0.weight"]
    assert m0.abs().max() > 0          # Adam moments were saved per parameter name

    # resume: epoch counter, weights and optimizer state continue (train.py:453-485)
    assert T.main(common + ["-e", "2", "--checkpoint", ck_path]) == 0
    out = capsys.readouterr().out
    assert "Train epoch 1:" in out and "Train epoch 0:" not in out
    # the saved checkpoint evaluates through the CLI
    assert EM.main(["-d", os.path.join(root, "test"), "-p", ck_path, "--entropy-estimation"]) == 0
    rep = json.loads(capsys.readouterr().out)
    assert rep["results"]["bpp"][0] > 0 and np.isfinite(rep["results"]["psnr"][0])


def test_optimizer_state_roundtrip_continues_bit_exactly():
    """Trainer.optimizer_state / load_optimizer_state: a restored trainer takes the same next step as the original"""
    from icm_amd.trainer import Trainer
    from icm_amd.zoo import models
    torch.manual_seed(11)
    net = models["cnn"]()
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    x = torch.rand(2, 3, 64, 64, device="cuda")
    noise = [{"z": torch.rand(2, 192, 1, 1) - 0.5, "y": torch.rand(2, 320, 4, 4) - 0.5} for _ in range(3)]
    tr = Trainer(net, device="cuda:0")
    tr.step(x, noise=noise[0])
    tr.step(x, noise=noise[1])
    opt, aux = tr.optimizer_state()
    mid = {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()}
    tr.step(x, noise=noise[2])
    torch.cuda.synchronize()
    want = {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()}

    net2 = models["cnn"]()
    net2.load_state_dict(mid)
    tr2 = Trainer(net2, device="cuda:0")
    tr2.load_optimizer_state(opt, aux)
    tr2.step(x, noise=noise[2])
    torch.cuda.synchronize()
    got = tr2.model.state_dict()
    for k in ("g_a.0.weight", "g_s.8.bias", "entropy_bottleneck.quantiles", "lrp_transforms.9.8.weight", "h_a.0.weight"):
        assert torch.equal(got[k].cpu(), want[k]), k
        assert not torch.equal(want[k], sd0[k]), k
