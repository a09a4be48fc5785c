"""compressai.datasets mirror (compressai/datasets/utils.py:23-89): the image-folder dataset the reference's training
script feeds the hot path with (train.py:404-425), plus the three torchvision transforms that script composes
(``CenterCrop`` / ``RandomCrop(pad_if_needed)`` / ``ToTensor``, train.py:393-402) -- torchvision is not a dependency
here, so they are restated on PIL + numpy with torchvision's semantics (centre offset rounding, zero padding, CHW f32
in [0, 1]).  Host-side data plumbing: no device work happens here."""
from __future__ import annotations

import random
from pathlib import Path
from typing import Callable, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset

IMG_EXTENSIONS = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".pgm", ".tif", ".tiff", ".webp")   # eval_model/__main__.py:56-67


class ImageFolder(Dataset):
    """``rootdir/<split>/*``: every regular file of the split directory is a sample (datasets/utils.py:46-55);
    ``__getitem__`` opens it as RGB and applies ``transform`` (:57-86)."""

    def __init__(self, root, transform: Optional[Callable] = None, split: str = "train"):
        splitdir = Path(root) / split
        if not splitdir.is_dir():
            raise RuntimeError(f'Invalid directory "{root}"')
        self.samples = sorted(f for f in splitdir.iterdir() if f.is_file())
        self.transform = transform

    def __getitem__(self, index):
        img = Image.open(self.samples[index]).convert("RGB")
        if self.transform:
            return self.transform(img)
        return img

    def __len__(self):
        return len(self.samples)


def _size2(size) -> Tuple[int, int]:
    if isinstance(size, (int, float)):
        return int(size), int(size)
    if len(size) == 1:
        return int(size[0]), int(size[0])
    return int(size[0]), int(size[1])


def _pad(img: Image.Image, left: int, top: int, right: int, bottom: int) -> Image.Image:
    if not (left or top or right or bottom):
        return img
    out = Image.new(img.mode, (img.width + left + right, img.height + top + bottom), 0)
    out.paste(img, (left, top))
    return out


class Compose:
    def __init__(self, transforms: Sequence[Callable]):
        self.transforms = list(transforms)

    def __call__(self, img):
        for t in self.transforms:
            img = t(img)
        return img


class ToTensor:
    """PIL RGB (or HxWxC uint8 array) -> f32 [C,H,W] in [0, 1]"""

    def __call__(self, img) -> torch.Tensor:
        a = np.asarray(img)
        if a.ndim == 2:
            a = a[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
        return t.to(torch.float32).div_(255.0) if t.dtype == torch.uint8 else t.to(torch.float32)


class CenterCrop:
    """torchvision semantics: images smaller than the crop are zero-padded symmetrically first (extra pixel on the
    right / bottom), the crop offset is ``round((size - crop) / 2)``"""

    def __init__(self, size):
        self.size = _size2(size)

    def __call__(self, img: Image.Image) -> Image.Image:
        ch, cw = self.size
        w, h = img.size
        if cw > w or ch > h:
            pl, pt = max((cw - w) // 2, 0), max((ch - h) // 2, 0)
            pr, pb = max((cw - w + 1) // 2, 0), max((ch - h + 1) // 2, 0)
            img = _pad(img, pl, pt, pr, pb)
            w, h = img.size
            if (cw, ch) == (w, h):
                return img
        top, left = int(round((h - ch) / 2.0)), int(round((w - cw) / 2.0))
        return img.crop((left, top, left + cw, top + ch))


class RandomCrop:
    """uniform crop position (Python ``random``: seeded by train.py:388-390's ``random.seed``); ``pad_if_needed`` zero-pads
    both sides of a too-small axis like torchvision"""

    def __init__(self, size, pad_if_needed: bool = False):
        self.size = _size2(size)
        self.pad_if_needed = pad_if_needed

    def __call__(self, img: Image.Image) -> Image.Image:
        ch, cw = self.size
        w, h = img.size
        if self.pad_if_needed and w < cw:
            img = _pad(img, cw - w, 0, cw - w, 0)
        if self.pad_if_needed and h < ch:
            img = _pad(img, 0, ch - h, 0, ch - h)
        w, h = img.size
        if w < cw or h < ch:
            raise ValueError(f"Required crop size {(ch, cw)} is larger than input image size {(h, w)}")
        top = random.randint(0, h - ch)
        left = random.randint(0, w - cw)
        return img.crop((left, top, left + cw, top + ch))


def to_pil_image(x: torch.Tensor) -> Image.Image:
    """f32 [3,H,W] in [0,1] -> PIL RGB (torchvision ToPILImage: x*255 truncated to uint8; eval_model/__main__.py:89-94)"""
    a = x.detach().to("cpu", torch.float32).mul(255.0).to(torch.uint8).numpy()
    return Image.fromarray(np.ascontiguousarray(a.transpose(1, 2, 0)), mode="RGB")
