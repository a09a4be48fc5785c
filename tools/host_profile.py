"""Host-side cost of one training step: cProfile of Trainer.step calls (GPU kept busy, no sync inside)."""
import cProfile, pstats, sys, os, time, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "image-compression-for-machine_amd"))
import torch
import bench
model = sys.argv[1] if len(sys.argv) > 1 else "cnn"
tr, x, _ = bench.make_workload(model, torch.device("cuda:0"))
for _ in range(3):
    tr.step(x)
torch.cuda.synchronize()
# pure host time: issue 3 steps back to back, time the issuing only, then sync
t0 = time.perf_counter()
for _ in range(3):
    tr.step(x)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{model}: host issue time {1e3 * (t1 - t0) / 3:.1f} ms/step, total {1e3 * (t2 - t0) / 3:.1f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    tr.step(x)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:5000])
