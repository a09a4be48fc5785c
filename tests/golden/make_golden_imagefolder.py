"""Emit tests/golden/imagefolder/ + imagefolder.npz from the REAL reference dataset class (this container only; needs
/root/reference and PIL).

``compressai.datasets.utils.ImageFolder`` (datasets/utils.py:23-89) is imported as it lies (a synthetic parent package
keeps compressai/__init__.py from running, as in _ref_loader.py); it is pointed at a small folder of seeded images
in several PIL modes (RGB, L, RGBA, P, 16-bit I;16) written by this script, and what it returns -- the sample set, the
RGB arrays of ``__getitem__`` without a transform, the value a transform receives, the RuntimeError for a missing
split -- is stored as the fixture that tests/test_data_pipeline.py::test_image_folder_vs_reference_fixture checks
``icm_amd.datasets.ImageFolder`` against.  The committed image files ARE the inputs (data, not source).
Usage: python tests/golden/make_golden_imagefolder.py"""
import importlib
import os
import sys
import types

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
ROOT = os.path.join(HERE, "imagefolder")


def write_images():
    d = os.path.join(ROOT, "train")
    os.makedirs(d, exist_ok=True)
    os.makedirs(os.path.join(d, "nested_dir"), exist_ok=True)       # directories are not samples
    with open(os.path.join(d, "nested_dir", ".keep"), "w") as fh:
        fh.write("")
    rng = np.random.default_rng(20261004)
    Image.fromarray(rng.integers(0, 256, (12, 20, 3), dtype=np.uint8), "RGB").save(os.path.join(d, "b_rgb.png"))
    Image.fromarray(rng.integers(0, 256, (9, 7), dtype=np.uint8), "L").save(os.path.join(d, "a_grey.png"))
    Image.fromarray(rng.integers(0, 256, (6, 11, 4), dtype=np.uint8), "RGBA").save(os.path.join(d, "c_rgba.png"))
    pal = Image.fromarray(rng.integers(0, 16, (8, 8), dtype=np.uint8), "P")
    pal.putpalette([int(v) for v in rng.integers(0, 256, 48)])
    pal.save(os.path.join(d, "d_palette.png"))
    Image.fromarray(rng.integers(0, 256, (10, 10, 3), dtype=np.uint8), "RGB").save(os.path.join(d, "e_rgb.bmp"))


def main():
    write_images()
    sys.dont_write_bytecode = True
    pkg = types.ModuleType("compressai")
    pkg.__path__ = [REF + "/compressai"]
    sys.modules["compressai"] = pkg
    dpkg = types.ModuleType("compressai.datasets")
    dpkg.__path__ = [REF + "/compressai/datasets"]
    sys.modules["compressai.datasets"] = dpkg
    U = importlib.import_module("compressai.datasets.utils")
    ds = U.ImageFolder(ROOT, split="train")
    out = {"names": np.array(sorted(p.name for p in ds.samples)), "len": np.array(len(ds))}
    seen = []
    ds_t = U.ImageFolder(ROOT, transform=lambda im: (seen.append((im.mode, im.size)), np.asarray(im))[1], split="train")
    for i in range(len(ds)):
        name = ds.samples[i].name
        im = ds[i]
        assert im.mode == "RGB"
        out["img." + name] = np.asarray(im).copy()
        k = [p.name for p in ds_t.samples].index(name)
        assert np.array_equal(ds_t[k], out["img." + name])
    assert all(m == "RGB" for m, _ in seen)
    try:
        U.ImageFolder(ROOT, split="missing")
        out["missing_split_raises"] = np.array("")
    except RuntimeError as e:
        out["missing_split_raises"] = np.array(str(e).replace(ROOT, "<root>"))
    np.savez_compressed(os.path.join(HERE, "imagefolder.npz"), **out)
    print("wrote imagefolder.npz:", list(out["names"]), int(out["len"]), str(out["missing_split_raises"]))


if __name__ == "__main__":
    main()
