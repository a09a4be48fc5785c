"""Same-box A/B of the conv kernels: run with ICM_LIB=<lib> ; prints auto-config time per shape (us)."""
import ctypes, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "image-compression-for-machine_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from icm_amd import _lib, engine as E
from icm_amd.engine import VT
from tune_conv import SHAPES, timeit
dev = torch.device("cuda:0")
out = []
for idx in range(len(SHAPES)):
    name, N, Cin, H, W, Cout, k, s, tr = SHAPES[idx]
    x = torch.randn(N, Cin, H, W, device=dev)
    w = (torch.randn(Cin, Cout, k, k, device=dev) if tr else torch.randn(Cout, Cin, k, k, device=dev)) * 0.05
    b = torch.zeros(Cout, device=dev)
    tape = E.Tape(need_grad=False)
    kw = dict(stride=s, pad=k // 2, transposed=tr, output_padding=(s - 1) if tr else 0)
    y = E.conv2d(tape, VT(x), w, b, **kw)
    ms = min(timeit(lambda: E.conv2d(tape, VT(x), w, b, out=y, **kw), iters=10) for _ in range(3))
    out.append(f"{ms*1e3:7.1f}")
print(os.path.basename(_lib.LIB_PATH).ljust(22), " ".join(out), flush=True)
out = []
for idx in range(len(SHAPES)):
    name, N, Cin, H, W, Cout, k, s, tr = SHAPES[idx]
    if tr:
        continue
    x = torch.randn(N, Cin, H, W, device=dev)
    OH = (H + 2 * (k // 2) - k) // s + 1
    dy = torch.randn(N, Cout, OH, OH, device=dev)
    gw = torch.empty(Cout, Cin, k, k, device=dev); gb = torch.empty(Cout, device=dev)
    tape = E.Tape(need_grad=False)
    ms = min(timeit(lambda: E.wgrad_launch(tape, dy, x, gw, Ca=Cout, Cb=Cin, KH=k, KW=k, stride=s, pad=k // 2, dbias=gb), iters=10) for _ in range(3))
    out.append(f"{ms*1e3:7.1f}")
print((os.path.basename(_lib.LIB_PATH) + " wgrad").ljust(22), " ".join(out), flush=True)
