"""Time single 3x3 stride-1 launches on the direct and the Winograd kernel (measurement tool; ICM_WINO_DEBUG /
ICM_WINO_TCO / ICM_WINO_PXFAST are read by the library at first use, so one process = one setting)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-compression-for-machine_amd"))
from icm_amd import engine as E  # noqa: E402

CASES = [  # name, N, Cin, H, W, Cout, members
    ("A 320->4256", 16, 320, 16, 16, 4256, 1),
    ("chain 224->176 x10", 16, 224, 16, 16, 176, 10),
    ("chain 224->176 x2", 16, 224, 16, 16, 176, 2),
    ("chain 224->176 x1", 16, 224, 16, 16, 176, 1),
    ("chain 176->128 x2", 16, 176, 16, 16, 128, 2),
    ("chain 128->64 x2", 16, 128, 16, 16, 64, 2),
    ("RU 96->96 @64 x2", 16, 96, 64, 64, 96, 2),
    ("RU 160->160 x2", 16, 160, 16, 16, 160, 2),
    ("dgradA 4480->320", 16, 4480, 16, 16, 320, 1),
    ("dgradB 3360->160", 16, 3360, 16, 16, 160, 1),
]


def main():
    d = torch.device("cuda:0")
    algos = [int(a) for a in os.environ.get("PROBE_ALGOS", "0,1").split(",")]
    for name, N, Cin, H, W, Cout, n in CASES:
        xs = [torch.randn(N, Cin, H, W, device=d) for _ in range(n)]
        ws = [torch.randn(Cout, Cin, 3, 3, device=d) * 0.05 for _ in range(n)]
        ys = [torch.empty(N, Cout, H, W, device=d) for _ in range(n)]
        out = []
        for algo in algos:
            tape = E.Tape(need_grad=False)
            wps = [tape.pack(w, Cout, Cin, 3, 3, 1, 0, 1, 1, wino=algo) for w in ws]
            kw = dict(Cin=Cin, Cout=Cout, KH=3, KW=3, stride=1, pad=1, transposed=0, OH=H, OW=W, algo=algo)
            run = lambda: E.conv_launch_grouped(tape, xs, wps, None, ys, **kw)
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
            tf = 2.0 * 9 * N * H * W * Cin * Cout * n / us / 1e6
            out.append(f"{'wino' if algo else 'direct'} {us:8.1f} us {tf:6.1f} TF")
        print(f"{name:22s} " + " | ".join(out), flush=True)


if __name__ == "__main__":
    main()
