"""Micro-benchmark of the weight-gradient kernels on the WACNN launch shapes (B=16, 256x256): every kernel variant
(icm_debug_force_wgrad_cfg) per shape, grouped as the step groups them; prints TFLOP/s.  GPU box only."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "image-compression-for-machine_amd"))
import torch
from icm_amd import _lib, engine as E

lib = _lib.lib()
dev = torch.device("cuda:0")
NV = 15

# name, N, Cb(in), H, W, Ca(out), k, stride, group
SHAPES = [
    ("1x1 192->192 @128 (gdn)", 16, 192, 128, 128, 192, 1, 1, 1),
    ("1x1 192->192 @64", 16, 192, 64, 64, 192, 1, 1, 1),
    ("1x1 96->192 @64 x6", 16, 96, 64, 64, 192, 1, 1, 6),
    ("1x1 192->96 @64 x4", 16, 192, 64, 64, 96, 1, 1, 4),
    ("1x1 192->576 @64", 16, 192, 64, 64, 576, 1, 1, 1),
    ("1x1 75->192 @128", 16, 75, 128, 128, 192, 1, 1, 1),
    ("1x1 320->320 @16", 16, 320, 16, 16, 320, 1, 1, 1),
    ("1x1 160->320 @16 x6", 16, 160, 16, 16, 320, 1, 1, 6),
    ("1x1 320->960 @16", 16, 320, 16, 16, 960, 1, 1, 1),
    ("3x3 96->96 @64 x6", 16, 96, 64, 64, 96, 3, 1, 6),
    ("5x5s2 192->192 @128", 16, 192, 128, 128, 192, 5, 2, 1),
    ("5x5s2 192->192 @64", 16, 192, 64, 64, 192, 5, 2, 1),
    ("5x5s2 192->320 @32", 16, 192, 32, 32, 320, 5, 2, 1),
    ("5x5s2 192->192 @32 (g_s.2 as wgrad)", 16, 192, 32, 32, 192, 5, 2, 1),
    ("3x3 480->224 @16 x10", 16, 480, 16, 16, 224, 3, 1, 10),
    ("3x3 224->176 @16 x11", 16, 224, 16, 16, 176, 3, 1, 11),
    ("3x3 128->64 @16 x10", 16, 128, 16, 16, 64, 3, 1, 10),
    ("3x3 64->32 @16 x10", 16, 64, 16, 16, 32, 3, 1, 10),
    ("3x3 64->32 @16 x2", 16, 64, 16, 16, 32, 3, 1, 2),
    ("3x3 176->128 @16 x10", 16, 176, 16, 16, 128, 3, 1, 10),
    ("3x3 512->224 @16 x5", 16, 512, 16, 16, 224, 3, 1, 5),
    ("3x3 224->176 @16 x2", 16, 224, 16, 16, 176, 3, 1, 2),
    ("3x3 160->160 @16 x6", 16, 160, 16, 16, 160, 3, 1, 6),
    ("3x3 320->320 @16", 16, 320, 16, 16, 320, 3, 1, 1),
    ("3x3 576->224 @16 x12", 16, 576, 16, 16, 224, 3, 1, 12),
    ("3x3 48->3 @256 (stf end_conv.2)", 16, 48, 256, 256, 3, 3, 1, 1),
    ("5x5s1 48->192 @128 (stf end_conv.0)", 16, 48, 128, 128, 192, 5, 1, 1),
    ("1x1 48->144 @128 (stf qkv)", 16, 48, 128, 128, 144, 1, 1, 1),
    ("1x1 48->48 @128 (stf proj)", 16, 48, 128, 128, 48, 1, 1, 1),
    ("1x1 192->48 @128 (stf fc2)", 16, 192, 128, 128, 48, 1, 1, 1),
    # virtual-GELU inputs (what most 3x3 / RU layers see in the step): name ends with "gelu"
    ("3x3 96->96 @64 x6 gelu", 16, 96, 64, 64, 96, 3, 1, 6),
    ("3x3 224->176 @16 x11 gelu", 16, 224, 16, 16, 176, 3, 1, 11),
    ("3x3 176->128 @16 x10 gelu", 16, 176, 16, 16, 128, 3, 1, 10),
    ("1x1 96->192 @64 x6 gelu", 16, 96, 64, 64, 192, 1, 1, 6),
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
    only = sys.argv[1] if len(sys.argv) > 1 else None
    for name, N, Cb, H, W, Ca, k, s, grp in SHAPES:
        if only and only not in name:
            continue
        OH, OW = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
        xs = [torch.randn(N, Cb, H, W, device=dev) for _ in range(grp)]
        dys = [torch.randn(N, Ca, OH, OW, device=dev) for _ in range(grp)]
        gws = [torch.empty(Ca, Cb, k, k, device=dev) for _ in range(grp)]
        gbs = [torch.empty(Ca, device=dev) for _ in range(grp)]
        flop = 2.0 * grp * N * Ca * Cb * k * k * OH * OW
        tape = E.Tape(need_grad=True)

        def run():
            for i in range(grp):
                E.wgrad_defer(tape, dys[i], xs[i], gws[i], Ca=Ca, Cb=Cb, KH=k, KW=k, stride=s, pad=k // 2, dbias=gbs[i],
                              act_b=_lib.ACT_GELU if name.endswith("gelu") else _lib.ACT_NONE)
            E.flush_wgrads(tape)
        res = {}
        for v in list(range(NV)) + [-1]:
            lib.icm_debug_force_wgrad_cfg(v, -1)
            try:
                res[v] = flop / timeit(run) / 1e9
            except Exception:
                res[v] = 0.0
        lib.icm_debug_force_wgrad_cfg(-1, -1)
        print(f"{name:26s} " + " ".join(f"{v}:{res[v]:5.1f}" for v in range(NV)) + f" | auto {res[-1]:5.1f} TF", flush=True)


if __name__ == "__main__":
    main()
