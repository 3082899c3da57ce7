"""One steady-state step of tools/trace_step_loop.py as a timeline (rocprofv3 kernel trace): start offset, duration and the idle gap in
front of every kernel (time since the latest end of any earlier kernel: > 0 = the chip had nothing running).  Median over the last steps."""
import csv, glob, sys, re
import numpy as np
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
maps = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("mapping")]
maps = maps[-(steps + 1):]
tl = []
for a, b in zip(maps[:-1], maps[1:]):
    seg = rows[a:b]
    t0 = int(seg[0]["Start_Timestamp"])
    line, busy_end = [], t0
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = re.sub(r"^void ", "", r["Kernel_Name"]); name = re.sub(r"\(.*", "", name)
        line.append((name[:44], (s - t0) / 1e3, (e - s) / 1e3, (s - busy_end) / 1e3))
        busy_end = max(busy_end, e)
    nxt = int(rows[b]["Start_Timestamp"])
    line.append(("(next step's first kernel)", (nxt - t0) / 1e3, 0.0, (nxt - busy_end) / 1e3))
    tl.append(line)
n = min(len(l) for l in tl)
tl = [l for l in tl if len(l) == n] or tl
print(f"{len(tl)} steps; columns: start offset (us), duration (us), idle gap in front (us; negative = overlaps an earlier kernel)")
idle = 0.0
for i in range(n):
    st = np.median([l[i][1] for l in tl]); du = np.median([l[i][2] for l in tl]); gp = np.median([l[i][3] for l in tl])
    idle += max(gp, 0.0)
    print(f"{st:9.1f} {du:8.1f} {gp:8.1f}   {tl[0][i][0]}")
print(f"sum of positive idle gaps: {idle:.1f} us per step")
