"""CPU: the data pipeline and the CLIs either side of the hot path: datasets/utils.py:23-89,
train.py:290-385,393-425,438, eval_model/__main__.py:70-94,489-547.  torchvision is not installed here, so the crop
transforms are checked against their documented semantics written out in numpy ("parity unpinned" versus torchvision
itself; the arithmetic is integer offsets and a uint8 -> f32 / 255 conversion)."""
import json
import os
import random
import sys

import numpy as np
import pytest
import torch
from PIL import Image

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "image-compression-for-machine_amd"))
from icm_amd import datasets as D  # noqa: E402


def _img(h, w, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)


def _folder(tmp_path, split, sizes):
    d = tmp_path / split
    d.mkdir(parents=True)
    arrs = []
    for i, (h, w) in enumerate(sizes):
        a = _img(h, w, 100 + i)
        Image.fromarray(a).save(d / f"img{i:03d}.png")
        arrs.append(a)
    return arrs


def test_image_folder(tmp_path):
    arrs = _folder(tmp_path, "train", [(20, 30), (17, 9), (8, 8)])
    (tmp_path / "train" / "sub").mkdir()          # directories are not samples (utils.py:52)
    Image.fromarray(_img(5, 5, 1)[:, :, 0]).save(tmp_path / "train" / "grey.png")   # converted to RGB (utils.py:65)
    ds = D.ImageFolder(str(tmp_path), split="train")
    assert len(ds) == 4
    names = [p.name for p in ds.samples]
    assert names == sorted(names)
    im = ds[names.index("img001.png")]
    assert isinstance(im, Image.Image) and im.mode == "RGB" and im.size == (9, 17)
    assert ds[names.index("grey.png")].mode == "RGB"
    ds = D.ImageFolder(str(tmp_path), split="train", transform=D.ToTensor())
    t = ds[names.index("img000.png")]
    assert t.dtype == torch.float32 and tuple(t.shape) == (3, 20, 30)
    assert torch.equal(t, torch.from_numpy(arrs[0].transpose(2, 0, 1).astype(np.float32) / 255.0))
    with pytest.raises(RuntimeError):
        D.ImageFolder(str(tmp_path), split="nope")


def test_image_folder_vs_reference_fixture(golden_dir):
    """icm_amd.datasets.ImageFolder against what the reference's own class (datasets/utils.py:23-89) returned for the
    committed folder tests/golden/imagefolder (fixture written by tests/golden/make_golden_imagefolder.py from the real
    class): same sample set, bit-identical RGB arrays for RGB / grey / RGBA / palette / BMP files, the value handed to
    a transform, the error for a missing split.  (The reference lists samples in directory order; the mirror sorts.)"""
    z = np.load(os.path.join(golden_dir, "imagefolder.npz"), allow_pickle=False)
    root = os.path.join(golden_dir, "imagefolder")
    ds = D.ImageFolder(root, split="train")
    assert len(ds) == int(z["len"])
    assert [p.name for p in ds.samples] == [str(n) for n in z["names"]]
    got = []
    ds_t = D.ImageFolder(root, transform=lambda im: (got.append(im.mode), np.asarray(im))[1], split="train")
    for i, p_ in enumerate(ds.samples):
        im = ds[i]
        assert im.mode == "RGB"
        ref = z["img." + p_.name]
        assert np.array_equal(np.asarray(im), ref), p_.name
        assert np.array_equal(ds_t[i], ref)
    assert got == ["RGB"] * len(ds)
    with pytest.raises(RuntimeError) as ei:
        D.ImageFolder(root, split="missing")
    assert str(ei.value).replace(root, "<root>") == str(z["missing_split_raises"])


def test_center_crop_semantics():
    a = _img(21, 30, 7)
    im = Image.fromarray(a)
    out = np.asarray(D.CenterCrop((8, 11))(im))
    top, left = int(round((21 - 8) / 2.0)), int(round((30 - 11) / 2.0))
    assert np.array_equal(out, a[top:top + 8, left:left + 11])
    # smaller than the crop on one axis: symmetric zero padding first (extra pixel right / bottom), then the crop
    out = np.asarray(D.CenterCrop(24)(im))
    assert out.shape == (24, 24, 3)
    padded = np.zeros((24, 30, 3), np.uint8)
    padded[1:22] = a                      # (24 - 21) // 2 = 1 on top, 2 at the bottom
    assert np.array_equal(out, padded[:, 3:27])
    assert np.array_equal(np.asarray(D.CenterCrop((21, 30))(im)), a)


def test_random_crop_semantics():
    a = _img(16, 12, 9)
    im = Image.fromarray(a)
    random.seed(5)
    outs = [np.asarray(D.RandomCrop((8, 8))(im)) for _ in range(20)]
    offs = set()
    for o in outs:
        found = [(t, l) for t in range(9) for l in range(5) if np.array_equal(o, a[t:t + 8, l:l + 8])]
        assert len(found) == 1
        offs.add(found[0])
    assert len(offs) > 5                                   # positions vary
    random.seed(5)
    again = [np.asarray(D.RandomCrop((8, 8))(im)) for _ in range(20)]
    assert all(np.array_equal(x, y) for x, y in zip(outs, again))   # seeded by random.seed (train.py:388-390)
    with pytest.raises(ValueError):
        D.RandomCrop((32, 8))(im)
    o = np.asarray(D.RandomCrop((32, 8), pad_if_needed=True)(im))    # height padded by 16 on both sides -> 48 rows
    assert o.shape == (32, 8, 3)
    pipeline = D.Compose([D.RandomCrop(8), D.ToTensor()])
    t = pipeline(im)
    assert tuple(t.shape) == (3, 8, 8) and 0.0 <= float(t.min()) and float(t.max()) <= 1.0


def test_to_pil_roundtrip():
    a = _img(6, 5, 3)
    t = D.ToTensor()(Image.fromarray(a))
    assert np.array_equal(np.asarray(D.to_pil_image(t)), a)
    assert np.asarray(D.to_pil_image(torch.full((3, 2, 2), 0.999))).max() == 254    # truncation, like ToPILImage


def test_eval_cli_host_side(tmp_path, capsys):
    from icm_amd import eval_model as EM
    _folder(tmp_path, "imgs", [(8, 8), (9, 7)])
    (tmp_path / "imgs" / "notes.txt").write_text("x")
    (tmp_path / "imgs" / "UPPER.JPG").write_bytes((tmp_path / "imgs" / "img000.png").read_bytes())
    files = EM.collect_images(str(tmp_path / "imgs"))
    assert [os.path.basename(f) for f in files] == ["UPPER.JPG", "img000.png", "img001.png"]
    x = EM.read_image(files[1])
    assert tuple(x.shape) == (3, 8, 8) and x.dtype == torch.float32
    EM.reconstruct(x.unsqueeze(0) * 2.0 - 0.5, "r.png", str(tmp_path / "rec"))     # clamped to [0, 1] before saving
    r = np.asarray(Image.open(tmp_path / "rec" / "r.png"))
    assert r.shape == (8, 8, 3) and r.min() == 0 and r.max() == 255
    a = EM.setup_args().parse_args(["-d", "x", "-a", "stf", "--entropy-estimation", "-p", "c.ckpt"])
    assert (a.dataset, a.architecture, a.entropy_estimation, a.paths, a.entropy_coder) == ("x", "stf", True, "c.ckpt", "ans")
    empty = tmp_path / "empty"
    empty.mkdir()
    assert EM.main(["-d", str(empty)]) == 1
    assert "no images" in capsys.readouterr().err
    assert EM.main(["-d", str(tmp_path / "imgs"), "--half"]) == 2
    if not torch.cuda.is_available():
        assert EM.main(["-d", str(tmp_path / "imgs")]) == 3          # the product path has no CPU fallback


def test_train_cli_host_side():
    from icm_amd import train as T
    a = T.parse_args(["-d", "root", "--lambda", "0.013", "--patch-size", "128", "192", "-lr", "5e-5", "--save"])
    assert (a.model, a.lmbda, tuple(a.patch_size), a.learning_rate, a.save, a.clip_max_norm) == \
        ("cnn", 0.013, (128, 192), 5e-5, True, 1.0)
    m = T.AverageMeter()
    m.update(2.0)
    m.update(4.0, n=3)
    assert m.avg == pytest.approx(3.5) and m.count == 4

    # PlateauLR == torch's ReduceLROnPlateau("min", factor 0.6, patience 6) on the same metric sequence
    class FakeTrainer:
        lr = 1e-4
    ft = FakeTrainer()
    mine = T.PlateauLR(ft, factor=0.6, patience=6)
    prm = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([prm], lr=1e-4)
    ref = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, "min", factor=0.6, patience=6)
    rng = np.random.default_rng(0)
    seq = list(1.0 + 0.2 * rng.random(12)) + [0.5] + list(0.6 + 0.1 * rng.random(40)) + [0.49995, 0.4] + [0.45] * 20
    for v in seq:
        mine.step(float(v))
        ref.step(float(v))
        assert ft.lr == pytest.approx(opt.param_groups[0]["lr"], rel=1e-12)
    assert ft.lr < 1e-4 * 0.6 ** 3
    sd = mine.state_dict()
    other = T.PlateauLR(FakeTrainer())
    other.load_state_dict(json.loads(json.dumps(sd)))
    assert (other.best, other.num_bad) == (mine.best, mine.num_bad)
