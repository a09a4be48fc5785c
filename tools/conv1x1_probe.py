"""Time the 1x1 convolution launches of the training step (measurement tool; ICM_1X1_DEBUG is read by the library at
first use, so one process = one setting)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-compression-for-machine_amd"))
from icm_amd import engine as E  # noqa: E402

CASES = [  # name, N, Cin, H, W, Cout, members, materialise gelu (y2), residual
    ("RU 96->192 @64 x2 res+y2", 16, 96, 64, 64, 192, 2, True, True),
    ("RU 96->192 @64 x2 plain", 16, 96, 64, 64, 192, 2, False, False),
    ("RU 192->96 @64 x2 y2", 16, 192, 64, 64, 96, 2, True, False),
    ("RU 192->96 @64 x2 plain", 16, 192, 64, 64, 96, 2, False, False),
    ("gdn 192->192 @128", 16, 192, 128, 128, 192, 1, False, False),
    ("qkv 192->576 @64", 16, 192, 64, 64, 576, 1, False, False),
    ("proj 192->192 @64", 16, 192, 64, 64, 192, 1, False, False),
    ("stf fc2 768->192 @32", 16, 768, 32, 32, 192, 1, False, True),
    ("stf fc1 192->768 @32", 16, 192, 32, 32, 768, 1, False, False),
    ("stf proj 192->192 @32", 16, 192, 32, 32, 192, 1, False, True),
    ("stf qkv 192->576 @32", 16, 192, 32, 32, 576, 1, False, False),
    ("stf fc2 384->96 @64", 16, 384, 64, 64, 96, 1, False, True),
    ("stf proj 96->96 @64", 16, 96, 64, 64, 96, 1, False, True),
    ("stf fc2 1536->384 @16", 16, 1536, 16, 16, 384, 1, False, True),
    ("RU 160->320 @16 x2", 16, 160, 16, 16, 320, 2, True, True),
    ("RU 320->160 @16 x2", 16, 320, 16, 16, 160, 2, True, False),
]


def main():
    d = torch.device("cuda:0")
    for name, N, Cin, H, W, Cout, n, mat, res in CASES:
        xs = [torch.randn(N, Cin, H, W, device=d) for _ in range(n)]
        ws = [torch.randn(Cout, Cin, 1, 1, device=d) * 0.05 for _ in range(n)]
        bs = [torch.zeros(Cout, device=d) for _ in range(n)]
        ys = [torch.empty(N, Cout, H, W, device=d) for _ in range(n)]
        y2s = [torch.empty(N, Cout, H, W, device=d) for _ in range(n)] if mat else None
        rs = [torch.randn(N, Cout, H, W, device=d) for _ in range(n)] if res else None
        tape = E.Tape(need_grad=False)
        wps = [tape.pack(w, Cout, Cin, 1, 1, 1, 0, 1, 0) for w in ws]
        kw = dict(Cin=Cin, Cout=Cout, KH=1, KW=1, stride=1, pad=0, transposed=0, OH=H, OW=W,
                  epi=E.EPI_RES if res else E.EPI_NONE, y2s=y2s, ress=rs)
        run = lambda: E.conv_launch_grouped(tape, xs, wps, bs, ys, **kw)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        tf = 2.0 * N * H * W * Cin * Cout * n / us / 1e6
        mb = 4.0 * N * H * W * n * (Cin + Cout * (1 + (1 if mat else 0) + (1 if res else 0))) / 1e6
        print(f"{name:28s} {us:8.1f} us {tf:6.1f} TF  {mb / us:5.2f} TB/s ({mb:.0f} MB)", flush=True)


if __name__ == "__main__":
    main()
