"""Summarise a rocprofv3 kernel trace of tools/trace_b1.py: per-kernel mean duration in launch order over the last
replays, the gaps between consecutive kernels, and the replay span."""
import csv, glob, sys
import numpy as np
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# one replay = the span between two consecutive launches of the mapping kernel
idx = [i for i, n in enumerate(names) if n.startswith("mapping")]
per = idx[-1] - idx[-2]
reps = [(a, a + per) for a in idx[-60:-1] if a + per <= len(rows)]
dur = np.zeros(per); gap = np.zeros(per); span = []
for a, b in reps:
    seg = rows[a:b]
    if [r["Kernel_Name"] for r in seg] != names[reps[0][0]:reps[0][1]]:
        continue
    st = np.array([int(r["Start_Timestamp"]) for r in seg]); en = np.array([int(r["End_Timestamp"]) for r in seg])
    dur += en - st; gap[1:] += st[1:] - en[:-1]; span.append(en[-1] - st[0])
k = len(span)
print(f"{k} replays, {per} kernels per replay, span {np.mean(span) / 1e3:.1f} us, sum of kernels {dur.sum() / k / 1e3:.1f} us, sum of gaps {gap.sum() / k / 1e3:.1f} us")
for i in range(per):
    print(f"{i:2d} {dur[i] / k / 1e3:7.1f} us  gap before {gap[i] / k / 1e3:5.1f} us  {names[reps[0][0] + i][:90]}")
