"""``python -m icm_amd.eval_model`` -- the image-compression branch of the reference's evaluation CLI
(compressai/utils/eval_model/__main__.py:627-671: collect images, load a checkpoint, ``update(force=True)``, run every
image through ``inference`` (real rANS bit-streams) or ``inference_entropy_estimation`` (forward pass), average the
metrics and print the JSON report) on the HIP path.  Same flags (-d/-r/-a/-c/-p/--entropy-estimation/--half/-v), same
report layout; the task-model branches of the reference (detectron2 / segmentation, :553-625) are out of scope.

Differences kept deliberate: ``--entropy-estimation`` and ``--cuda`` are real booleans (the reference declares them
without ``type``/``action``, so any string is truthy); ``--half`` is refused (the HIP path is f32 end to end, like the
training path it mirrors); the reference's hard-coded default paths are replaced by required arguments."""
from __future__ import annotations

import argparse
import json
import os
import sys
from collections import defaultdict
from typing import Dict, List

import torch

from . import utils as U
from .datasets import IMG_EXTENSIONS, ToTensor, to_pil_image
from .zoo import models


def collect_images(rootpath: str) -> List[str]:
    """eval_model/__main__.py:70-75"""
    return sorted(os.path.join(rootpath, f) for f in os.listdir(rootpath)
                  if os.path.splitext(f)[-1].lower() in IMG_EXTENSIONS)


def read_image(filepath: str) -> torch.Tensor:
    """eval_model/__main__.py:83-86"""
    from PIL import Image
    if not os.path.isfile(filepath):
        raise FileNotFoundError(filepath)
    return ToTensor()(Image.open(filepath).convert("RGB"))


def reconstruct(reconstruction: torch.Tensor, filename: str, recon_path: str) -> None:
    """eval_model/__main__.py:89-94: clamp to [0,1] and save next to the metrics"""
    os.makedirs(recon_path, exist_ok=True)
    to_pil_image(reconstruction.squeeze(0).clamp(0, 1)).save(os.path.join(recon_path, filename))


def load_checkpoint(arch: str, checkpoint_path: str) -> torch.nn.Module:
    """eval_model/__main__.py:250-253: ``models[arch]()`` + the "state_dict" entry of a training checkpoint.  Only
    tensors are unpickled (``weights_only=True``); a bare state-dict file is accepted as well."""
    ck = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    sd = ck["state_dict"] if isinstance(ck, dict) and "state_dict" in ck else ck
    net = models[arch]()
    net.load_state_dict(sd)
    return net.eval()


@torch.no_grad()
def eval_model(model, filepaths: List[str], entropy_estimation: bool = False, recon_path: str = "") -> Dict[str, float]:
    """eval_model/__main__.py:472-487 (per-image metrics averaged over the folder)"""
    device = next(model.parameters()).device
    metrics: Dict[str, float] = defaultdict(float)
    for f in filepaths:
        x = read_image(f).to(device)
        fn = U.inference_entropy_estimation if entropy_estimation else U.inference
        rv = fn(model, x, recon=(lambda xh, name=os.path.basename(f): reconstruct(xh, name, recon_path)) if recon_path else None)
        for k, v in rv.items():
            metrics[k] += v
    return {k: v / len(filepaths) for k, v in metrics.items()}


def setup_args() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog="icm_amd.eval_model")
    p.add_argument("-d", "--dataset", type=str, required=True, help="dataset path (a folder of images)")
    p.add_argument("-r", "--recon_path", type=str, default="", help="where to save reconstructed images (empty: do not save)")
    p.add_argument("-a", "--architecture", default="cnn", type=str, choices=sorted(models.keys()), help="model architecture")
    p.add_argument("-c", "--entropy-coder", choices=["ans"], default="ans", help="entropy coder (default: %(default)s)")
    p.add_argument("--cuda", dest="cuda", action="store_true", default=True, help="run on the GPU (required by the HIP path)")
    p.add_argument("--half", action="store_true", default=False, help="refused: the HIP path is f32")
    p.add_argument("--entropy-estimation", action="store_true", default=False,
                   help="use evaluated entropy estimation (no entropy coding)")
    p.add_argument("-v", "--verbose", action="store_true", help="verbose mode")
    p.add_argument("-p", "--path", dest="paths", type=str, default=None,
                   help="checkpoint path (default: the architecture's initial weights)")
    p.add_argument("--limit", type=int, default=0, help="evaluate only the first N images (0 = all)")
    return p


def main(argv) -> int:
    args = setup_args().parse_args(argv)
    if args.half:
        print("Error: --half is not supported (f32 path).", file=sys.stderr)
        return 2
    filepaths = collect_images(args.dataset)
    if args.limit > 0:
        filepaths = filepaths[:args.limit]
    if len(filepaths) == 0:
        print("Error: no images found in directory.", file=sys.stderr)
        return 1
    if not torch.cuda.is_available():
        print("Error: no GPU (the HIP path has no CPU fallback).", file=sys.stderr)
        return 3
    model = load_checkpoint(args.architecture, args.paths) if args.paths else models[args.architecture]().eval()
    model = model.to("cuda")
    model.update(force=True)
    if args.verbose:
        sys.stderr.write(f"Evaluating {args.paths or '<initial weights>'} on {len(filepaths)} images\n")
    metrics = eval_model(model, filepaths, args.entropy_estimation, args.recon_path)
    results = defaultdict(list)
    for k, v in metrics.items():
        results[k].append(v)
    description = "entropy estimation" if args.entropy_estimation else args.entropy_coder
    print(json.dumps({"name": args.architecture, "description": f"Inference ({description})", "results": results}, indent=2))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
