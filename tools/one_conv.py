"""Run one conv shape/config a few times (for rocprofv3 PMC passes). usage: one_conv.py <shape_idx> <cfg> [wgrad]"""
import ctypes, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "image-compression-for-machine_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from icm_amd import _lib, engine as E
from icm_amd.engine import VT
from tune_conv import SHAPES
lib = _lib.lib()
lib.icm_debug_force_conv_cfg.argtypes = [ctypes.c_int]
name, N, Cin, H, W, Cout, k, s, tr = SHAPES[int(sys.argv[1])]
cfg = int(sys.argv[2])
dev = torch.device("cuda:0")
x = torch.randn(N, Cin, H, W, device=dev)
w = (torch.randn(Cin, Cout, k, k, device=dev) if tr else torch.randn(Cout, Cin, k, k, device=dev)) * 0.05
b = torch.zeros(Cout, device=dev)
tape = E.Tape(need_grad=False)
kw = dict(stride=s, pad=k // 2, transposed=tr, output_padding=(s - 1) if tr else 0)
lib.icm_debug_force_conv_cfg(cfg)
y = E.conv2d(tape, VT(x), w, b, **kw)
for _ in range(3):
    E.conv2d(tape, VT(x), w, b, out=y, **kw)
if len(sys.argv) > 3:
    dy = torch.randn_like(y); gw = torch.empty_like(w); gb = torch.empty(Cout, device=dev)
    for _ in range(3):
        E.wgrad_launch(tape, dy, x, gw, Ca=Cout, Cb=Cin, KH=k, KW=k, stride=s, pad=k // 2, dbias=gb)
torch.cuda.synchronize()
print("done", name, cfg)
