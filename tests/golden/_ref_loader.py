"""Loader for the upstream reference (THIS CONTAINER ONLY; never runs on the GPU box).

Imports /root/reference's ``compressai.models.cnn`` / ``.stf`` without executing the
package ``__init__`` files that pull in detectron2 / the cp38 ``.so`` files / timm
(SURVEY.md Appendix C).  Used only by ``make_golden.py`` to emit the committed
fixtures under ``tests/golden/``.  Nothing from the reference is copied: the stubs below
are inert stand-ins for *third-party* modules that are not installed (timm) and for the
two pre-built binaries that are never loaded (compressai._CXX, compressai.ans); none of
them is on the forward()/backward() path that the fixtures capture.
"""
import importlib
import sys
import types

import torch

REF = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _DropPath(torch.nn.Module):
    def __init__(self, p=0.0):
        super().__init__()
        self.p = float(p)

    def forward(self, x):
        if self.p == 0.0 or not self.training:
            return x
        keep = 1.0 - self.p
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return x * mask / keep


class _Inert:
    def __init__(self, *a, **k):
        pass


def load_reference():
    """Returns (cnn_module, stf_module) of the real reference code."""
    if "compressai.models.cnn" in sys.modules:
        return sys.modules["compressai.models.cnn"], sys.modules.get("compressai.models.stf")
    sys.dont_write_bytecode = True
    _mod("timm")
    _mod("timm.models")
    _mod(
        "timm.models.layers",
        DropPath=_DropPath,
        to_2tuple=lambda v: (v, v) if not isinstance(v, tuple) else v,
        trunc_normal_=lambda t, std=1.0, **k: torch.nn.init.trunc_normal_(t, std=std, a=-2.0, b=2.0),
    )
    pkg = _mod("compressai", get_entropy_coder=lambda: "ans", available_entropy_coders=lambda: ["ans"])
    pkg.__path__ = [REF + "/compressai"]
    mpkg = _mod("compressai.models")
    mpkg.__path__ = [REF + "/compressai/models"]
    _mod("compressai._CXX", pmf_to_quantized_cdf=lambda pmf, prec: [0] * (len(pmf) + 1))
    _mod("compressai.ans", RansEncoder=_Inert, RansDecoder=_Inert, BufferedRansEncoder=_Inert)
    cnn = importlib.import_module("compressai.models.cnn")
    try:
        stf = importlib.import_module("compressai.models.stf")
    except Exception as e:  # pragma: no cover
        print("stf import failed:", e)
        stf = None
    return cnn, stf
