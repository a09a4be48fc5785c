"""eval-mode forward at the bench geometry: eager launches vs hipGraph replay (GraphedForward)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "image-compression-for-machine_amd"))
import bench  # noqa: E402
from icm_amd.graphs import GraphedForward  # noqa: E402


def timed(fn, warmup, steps):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    host = (time.perf_counter() - t0) / steps * 1e3
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps, host


def main():
    dev = torch.device("cuda:0")
    for name in ("cnn", "stf"):
        tr, x, _ = bench.make_workload(name, dev)
        net = tr.model.eval()
        with torch.no_grad():
            ms, host = timed(lambda: net(x), 3, 20)
            print(f"{name}: eager   {ms:7.3f} ms/fwd ({16 / ms * 1e3:7.1f} img/s), host issue time {host:6.3f} ms", flush=True)
            gf = GraphedForward(net)
            ms, host = timed(lambda: gf(x), 3, 20)
            print(f"{name}: graphed {ms:7.3f} ms/fwd ({16 / ms * 1e3:7.1f} img/s), host issue time {host:6.3f} ms", flush=True)
        del tr, net, gf


if __name__ == "__main__":
    main()
