"""Time batched 3x3 stride-1 weight-gradient launches on the direct and the Winograd kernel (measurement tool)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-compression-for-machine_amd"))
from icm_amd import engine as E  # noqa: E402

CASES = [  # name, N, Cb (in), H, W, Ca (out), problems
    ("first 320->224 x30", 16, 320, 16, 16, 224, 30),
    ("support 160->224 x15", 16, 160, 16, 16, 224, 15),
    ("chain 224->176 x10", 16, 224, 16, 16, 176, 10),
    ("chain 224->176 x3", 16, 224, 16, 16, 176, 3),
    ("chain 176->128 x10", 16, 176, 16, 16, 128, 10),
    ("chain 128->64 x10", 16, 128, 16, 16, 64, 10),
    ("chain 64->32 x10", 16, 64, 16, 16, 32, 10),
    ("RU 96->96 @64 x6", 16, 96, 64, 64, 96, 6),
    ("RU 160->160 x6", 16, 160, 16, 16, 160, 6),
    ("h_s 288->320 x2", 16, 288, 16, 16, 320, 2),
]


def main():
    d = torch.device("cuda:0")
    for name, N, Cb, H, W, Ca, n in CASES:
        xs = [torch.randn(N, Cb, H, W, device=d) for _ in range(n)]
        gs = [torch.randn(N, Ca, H, W, device=d) for _ in range(n)]
        dws = [torch.zeros(Ca, Cb, 3, 3, device=d) for _ in range(n)]
        dbs = [torch.zeros(Ca, device=d) for _ in range(n)]
        out = []
        for algo in (0, 1):
            tape = E.Tape(need_grad=True)

            def run():
                for i in range(n):
                    E.wgrad_defer(tape, gs[i], xs[i], dws[i], Ca=Ca, Cb=Cb, KH=3, KW=3, stride=1, pad=1, accum=0,
                                  dbias=dbs[i], accum_bias=0, algo=algo)
                E.flush_wgrads(tape)
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10
            e0.record()
            for _ in range(reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / reps * 1e3
            tf = 2.0 * 9 * N * H * W * Cb * Ca * n / us / 1e6
            out.append(f"{'wino' if algo else 'direct'} {us:8.1f} us {tf:6.1f} TF")
        print(f"{name:22s} " + " | ".join(out), flush=True)


if __name__ == "__main__":
    main()
