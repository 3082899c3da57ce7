"""Summarise a rocprofv3 kernel trace of tools/bench_train.py: per (kernel, grid) total / mean duration per iteration."""
import csv, glob, sys, re
import numpy as np
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
iters = int(sys.argv[2])
rows = list(csv.DictReader(open(path)))
acc = {}
for r in rows:
    name = re.sub(r"^void ", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name)
    g = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"])), int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_Z"])))
    acc.setdefault((name, g), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in acc.values()) / 1e6
print(f"sum of kernel durations {tot / iters:.2f} ms/iteration, {sum(len(v) for v in acc.values()) / iters:.0f} launches/iteration")
top = int(sys.argv[3]) if len(sys.argv) > 3 else 70
for (name, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print(f"{sum(v) / 1e6 / iters:8.3f} ms/it  {len(v) / iters:6.2f} x {np.mean(v) / 1e3:8.1f} us  grid {g}  {name[:90]}")
