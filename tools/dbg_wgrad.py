"""Time-breakdown experiment for wgrad_kernel: flags skip MFMA (1), final stores (2), staging (4), reduce kernel (8)."""
import ctypes, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "image-compression-for-machine_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from icm_amd import _lib, engine as E
from tune_conv import SHAPES, timeit
lib = _lib.lib()
lib.icm_debug_set_wgrad_flags.argtypes = [ctypes.c_int]
dev = torch.device("cuda:0")
for idx in [int(a) for a in sys.argv[1:]]:
    name, N, Cin, H, W, Cout, k, s, tr = SHAPES[idx]
    if tr: continue
    x = torch.randn(N, Cin, H, W, device=dev)
    OH = (H + 2 * (k // 2) - k) // s + 1
    dy = torch.randn(N, Cout, OH, OH, device=dev)
    gw = torch.empty(Cout, Cin, k, k, device=dev); gb = torch.empty(Cout, device=dev)
    tape = E.Tape(need_grad=False)
    flop = 2.0 * N * Cout * Cin * k * k * OH * OH
    out = []
    for fl in (0, 8, 9, 10, 12, 11, 13, 14, 15):
        lib.icm_debug_set_wgrad_flags(fl)
        ms = timeit(lambda: E.wgrad_launch(tape, dy, x, gw, Ca=Cout, Cb=Cin, KH=k, KW=k, stride=s, pad=k // 2, dbias=gb), iters=10)
        out.append(f"{fl}:{ms*1e3:6.1f}")
    lib.icm_debug_set_wgrad_flags(0)
    print(f"{name:28s} ({flop/1e9:6.2f} GF, ideal {flop/157.3e12*1e6:5.1f}us) " + " ".join(out), flush=True)
