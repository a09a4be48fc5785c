"""Per-shape summary of a kernel trace pulled by tools/trace_step.sh: picks the last full step (the launch
sequence repeats), reports busy time, idle gaps and the top (kernel, grid, LDS) groups."""
import csv, gzip, sys, re
from collections import defaultdict
rows = list(csv.DictReader(gzip.open(sys.argv[1], "rt")))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
names = [re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void icm::", "").replace("icm::", "") for r in rows]
# step boundary: first kernel of the model = the conv on the 3-channel input; find launches of the rd/first marker
sig = [(n, r["Grid_Size_X"], r["LDS_Block_Size"]) for n, r in zip(names, rows)]
first = sig.index(next(s for s in sig if s[0].startswith("pack_weights")))
# find period: next occurrence where the following 50 signatures repeat
period = None
for p in range(200, len(sig) - first - 50):
    if sig[first + p:first + p + 50] == sig[first:first + 50]:
        period = p; break
print("rows", len(rows), "first", first, "period", period)
marks = [i for i, n in enumerate(names) if n.startswith("sqnorm_kernel")]
if len(marks) >= 2:   # train trace: one grad-norm kernel per step -> take the last full step (windowed packing makes step 1 differ)
    period = marks[-1] - marks[-2]
    a = marks[-2] + 4   # sqnorm, adam, eb_aux, adam_aux end the previous step
    b = marks[-1] + 4
else:
    a = first + period
    b = a + period if a + period <= len(rows) else len(rows)
step = list(zip(names[a:b], rows[a:b]))
t0, t1 = step[0][1]["s"], max(r["e"] for _, r in step)
busy = 0; cur_end = t0; gaps = 0
for n, r in step:
    if r["s"] > cur_end: gaps += r["s"] - cur_end
    cur_end = max(cur_end, r["e"])
print(f"step wall {(t1-t0)/1e6:.2f} ms, idle gaps {gaps/1e6:.2f} ms, launches {len(step)}")
g = defaultdict(lambda: [0, 0])
for n, r in step:
    k = (n, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Grid_Size_Y"], r["LDS_Block_Size"])
    g[k][0] += r["e"] - r["s"]; g[k][1] += 1
tot = sum(v[0] for v in g.values())
print(f"sum kernel time {tot/1e6:.2f} ms")
byname = defaultdict(int)
for k, v in g.items(): byname[k[0]] += v[0]
for n, v in sorted(byname.items(), key=lambda kv: -kv[1])[:14]: print(f"  {v/1e6:7.3f} ms {100*v/tot:5.1f}%  {n}")
print("top groups (kernel, workgroups, gridY, LDS): total ms, count, avg us")
for k, v in sorted(g.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"  {v[0]/1e6:7.3f} ms x{v[1]:4d} avg {v[0]/v[1]/1e3:8.1f} us  {k}")
if len(sys.argv) > 3:
    lo, hi = int(sys.argv[3]), int(sys.argv[4])
    for i, (n, r) in enumerate(step[lo:hi]):
        print(f"{lo+i:4d} {(r['s']-t0)/1e3:9.1f} us +{(r['e']-r['s'])/1e3:7.1f}  {n} wg={int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])} y={r['Grid_Size_Y']}")
