"""Micro-benchmark of the implicit-GEMM / wgrad kernels on the WACNN layer shapes (B=16, 256x256).
Times every tile configuration (icm_debug_force_conv_cfg) per shape; prints TFLOP/s. GPU box only."""
import ctypes
import json
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "image-compression-for-machine_amd"))
import torch
from icm_amd import _lib, engine as E
from icm_amd.engine import VT

lib = _lib.lib()
lib.icm_debug_force_conv_cfg.argtypes = [ctypes.c_int]
lib.icm_debug_force_conv_cfg.restype = None
dev = torch.device("cuda:0")
NCFG = 13

# name, N, Cin, H, W, Cout, k, stride, transposed
SHAPES = [
    ("g_a.2 c5s2 192->192 @128", 16, 192, 128, 128, 192, 5, 2, False),
    ("g_a.5 c5s2 192->192 @64", 16, 192, 64, 64, 192, 5, 2, False),
    ("g_a.7 c5s2 192->320 @32", 16, 192, 32, 32, 320, 5, 2, False),
    ("g_a.0 c5s2 3->192 @256", 16, 3, 256, 256, 192, 5, 2, False),
    ("g_s.6 t5s2 192->192 @64", 16, 192, 64, 64, 192, 5, 2, True),
    ("g_s.8 t5s2 192->3 @128", 16, 192, 128, 128, 3, 5, 2, True),
    ("RU c3 96->96 @64", 16, 96, 64, 64, 96, 3, 1, False),
    ("RU c1 192->96 @64", 16, 192, 64, 64, 96, 1, 1, False),
    ("RU c1 96->192 @64", 16, 96, 64, 64, 192, 1, 1, False),
    ("qkv c1 192->576 @64", 16, 192, 64, 64, 576, 1, 1, False),
    ("gdn c1 192->192 @128", 16, 192, 128, 128, 192, 1, 1, False),
    ("cc c3 480->224 @16", 16, 480, 16, 16, 224, 3, 1, False),
    ("cc c3 224->176 @16", 16, 224, 16, 16, 176, 3, 1, False),
    ("cc c3 176->128 @16", 16, 176, 16, 16, 128, 3, 1, False),
    ("cc c3 128->64 @16", 16, 128, 16, 16, 64, 3, 1, False),
    ("cc c3 64->32 @16", 16, 64, 16, 16, 32, 3, 1, False),
    ("RU c3 160->160 @16", 16, 160, 16, 16, 160, 3, 1, False),
]


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    out = {}
    for name, N, Cin, H, W, Cout, k, s, tr in SHAPES:
        x = torch.randn(N, Cin, H, W, device=dev)
        w = (torch.randn(Cin, Cout, k, k, device=dev) if tr else torch.randn(Cout, Cin, k, k, device=dev)) * 0.05
        b = torch.zeros(Cout, device=dev)
        tape = E.Tape(need_grad=False)
        kw = dict(stride=s, pad=k // 2, transposed=tr, output_padding=(s - 1) if tr else 0)
        y = E.conv2d(tape, VT(x), w, b, **kw)
        OH, OW = y.shape[2], y.shape[3]
        flop = 2.0 * N * Cout * Cin * k * k * (H * W if tr else OH * OW)
        res = {}
        for cfg in list(range(NCFG)) + [-1]:
            lib.icm_debug_force_conv_cfg(cfg)
            try:
                ms = timeit(lambda: E.conv2d(tape, VT(x), w, b, out=y, **kw))
                res[cfg] = flop / ms / 1e9
            except Exception as e:  # unsupported (LDS) for this shape
                res[cfg] = 0.0
        lib.icm_debug_force_conv_cfg(-1)
        best = max((c for c in res if c >= 0), key=lambda c: res[c])
        print(f"{name:28s} " + " ".join(f"{c}:{res[c]:5.1f}" for c in range(NCFG)) + f" | auto {res[-1]:5.1f} best cfg{best} {res[best]:5.1f} TF")
        # wgrad
        if not tr:
            dy = torch.randn_like(y)
            gw = torch.empty_like(w)
            gb = torch.empty(Cout, device=dev)
            ms = timeit(lambda: E.wgrad_launch(tape, dy, x, gw, Ca=Cout, Cb=Cin, KH=k, KW=k, stride=s, pad=k // 2, dbias=gb))
            print(f"{'':28s} wgrad {flop / ms / 1e9:5.1f} TF ({ms*1e3:.0f} us)")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
