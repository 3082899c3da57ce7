"""Summarise a rocprofv3 kernel trace of tools/trace_step_loop.py: per (kernel, grid) mean duration and launches per
step over the timed steps (the last NB_STEPS steps; a step = one launch of the mapping kernel per sub-batch chain)."""
import csv, glob, sys, re
import numpy as np
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 150
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
maps = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("mapping")]
first = maps[-steps] if len(maps) >= steps else maps[0]
rows = rows[first:]
nsteps = sum(1 for r in rows if r["Kernel_Name"].startswith("mapping"))
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
acc = {}
for r in rows:
    name = re.sub(r"^void ", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name)
    g = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"])), int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_Z"])))
    acc.setdefault((name, g), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in acc.values()) / 1e3
print(f"{nsteps} steps, span {span / nsteps:.1f} us/step, sum of kernel durations {tot / nsteps:.1f} us/step")
for (name, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{sum(v) / 1e3 / nsteps:8.1f} us/step  {len(v) / nsteps:5.2f} x {np.mean(v) / 1e3:7.1f} us  grid {g}  {name[:70]}")
