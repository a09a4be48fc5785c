"""Time-breakdown experiment for conv_igemm_kernel: debug flags skip the MFMA loop (1), the epilogue (2), the staging (4).
usage: dbg_conv.py <shape_idx>..."""
import ctypes, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "image-compression-for-machine_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from icm_amd import _lib, engine as E
from icm_amd.engine import VT
from tune_conv import SHAPES, timeit
lib = _lib.lib()
lib.icm_debug_set_flags.argtypes = [ctypes.c_int]
lib.icm_debug_force_conv_cfg.argtypes = [ctypes.c_int]
dev = torch.device("cuda:0")
for idx in [int(a) for a in sys.argv[1:]]:
    name, N, Cin, H, W, Cout, k, s, tr = SHAPES[idx]
    x = torch.randn(N, Cin, H, W, device=dev)
    w = (torch.randn(Cin, Cout, k, k, device=dev) if tr else torch.randn(Cout, Cin, k, k, device=dev)) * 0.05
    b = torch.zeros(Cout, device=dev)
    tape = E.Tape(need_grad=False)
    kw = dict(stride=s, pad=k // 2, transposed=tr, output_padding=(s - 1) if tr else 0)
    y = E.conv2d(tape, VT(x), w, b, **kw)
    flop = 2.0 * N * Cout * Cin * k * k * (H * W if tr else y.shape[2] * y.shape[3])
    out = []
    for fl in (0, 1, 2, 4, 3, 5, 6, 7):
        lib.icm_debug_set_flags(fl)
        ms = timeit(lambda: E.conv2d(tape, VT(x), w, b, out=y, **kw), iters=10)
        out.append(f"{fl}:{ms*1e3:6.1f}us")
    lib.icm_debug_set_flags(0)
    print(f"{name:28s} ({flop/1e9:6.2f} GF, ideal {flop/157.3e12*1e6:5.1f}us) " + " ".join(out), flush=True)
