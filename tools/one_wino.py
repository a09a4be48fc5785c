"""Run ONE Winograd launch a few times (for rocprofv3 PMC / trace passes).
usage: one_wino.py conv    -> conv_wino_kernel on the slice chains' grouped second layer (224 -> 176 @16x16, 10 members)
       one_wino.py wgrad   -> wgrad_wino_kernel on the chains' first-layer weight gradients (512 -> 224 @16x16, 5 problems)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-compression-for-machine_amd"))
import torch
from icm_amd import engine as E

dev = torch.device("cuda:0")
what = sys.argv[1] if len(sys.argv) > 1 else "conv"
if what == "conv":
    N, Cin, H, W, Cout, n = 16, 224, 16, 16, 176, 10
    xs = [torch.randn(N, Cin, H, W, device=dev) for _ in range(n)]
    ws = [torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05 for _ in range(n)]
    ys = [torch.empty(N, Cout, H, W, device=dev) for _ in range(n)]
    tape = E.Tape(need_grad=False)
    wps = [tape.pack(w, Cout, Cin, 3, 3, 1, 0, 1, 1, wino=1) for w in ws]
    for _ in range(5):
        E.conv_launch_grouped(tape, xs, wps, None, ys, Cin=Cin, Cout=Cout, KH=3, KW=3, stride=1, pad=1, transposed=0, OH=H,
                              OW=W, algo=1)
    torch.cuda.synchronize()
    # algorithmic bytes: every member reads its input and its weights once and writes its output once
    print("done conv_wino: algorithmic bytes", n * 4 * (N * Cin * H * W + N * Cout * H * W + Cout * Cin * 9),
          "algorithmic flop", 2.0 * 9 * n * N * H * W * Cin * Cout)
else:
    N, Cb, H, W, Ca, n = 16, 512, 16, 16, 224, 5
    xs = [torch.randn(N, Cb, H, W, device=dev) for _ in range(n)]
    gs = [torch.randn(N, Ca, H, W, device=dev) for _ in range(n)]
    dws = [torch.zeros(Ca, Cb, 3, 3, device=dev) for _ in range(n)]
    dbs = [torch.zeros(Ca, device=dev) for _ in range(n)]
    tape = E.Tape(need_grad=True)
    for _ in range(5):
        for i in range(n):
            E.wgrad_defer(tape, gs[i], xs[i], dws[i], Ca=Ca, Cb=Cb, KH=3, KW=3, stride=1, pad=1, dbias=dbs[i], algo=1)
        E.flush_wgrads(tape)
    torch.cuda.synchronize()
    print("done wgrad_wino: algorithmic bytes", n * 4 * (N * Cb * H * W + N * Ca * H * W + Ca * Cb * 9),
          "algorithmic flop", 2.0 * 9 * n * N * H * W * Cb * Ca)
