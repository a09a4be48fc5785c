"""Run ONE weight-gradient shape a few times (for rocprofv3 PMC / trace passes).
usage: one_wgrad.py <substring of a tools/tune_wgrad.py SHAPES name>"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "image-compression-for-machine_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from icm_amd import _lib, engine as E
from tune_wgrad import SHAPES
dev = torch.device("cuda:0")
name, N, Cb, H, W, Ca, k, s, grp = next(sh for sh in SHAPES if sys.argv[1] in sh[0])
OH, OW = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
xs = [torch.randn(N, Cb, H, W, device=dev) for _ in range(grp)]
dys = [torch.randn(N, Ca, OH, OW, device=dev) for _ in range(grp)]
gws = [torch.empty(Ca, Cb, k, k, device=dev) for _ in range(grp)]
gbs = [torch.empty(Ca, device=dev) for _ in range(grp)]
tape = E.Tape(need_grad=True)
for _ in range(4):
    for i in range(grp):
        E.wgrad_defer(tape, dys[i], xs[i], gws[i], Ca=Ca, Cb=Cb, KH=k, KW=k, stride=s, pad=k // 2, dbias=gbs[i],
                      act_b=_lib.ACT_GELU if name.endswith("gelu") else _lib.ACT_NONE)
    E.flush_wgrads(tape)
torch.cuda.synchronize()
print("done", name, "algorithmic bytes", grp * 4 * (N * Cb * H * W + N * Ca * OH * OW + Ca * Cb * k * k))
