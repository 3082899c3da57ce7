"""GPU busy fraction from a rocprofv3 kernel trace of bench.py: union of kernel intervals / wall time over the
steady-state tail (last `frac` of the trace), plus the time only ONE kernel was resident (no overlap partner)."""
import csv, glob, sys
import numpy as np
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(path))]
rows.sort()
t_end = max(e for _, e in rows); t_beg = rows[0][0]
# window: the last `frac` of the trace, minus its final 5 % (tail effects)
hi = t_end - int((t_end - t_beg) * 0.05)
lo = hi - int((t_end - t_beg) * frac)
rows = [(s, min(e, hi)) for s, e in rows if s < hi]
t_end = hi
ev = []
for s, e in rows:
    if e <= lo: continue
    ev.append((max(s, lo), 1)); ev.append((e, -1))
ev.sort()
depth = 0; last = lo; busy = 0; single = 0
for t, d in ev:
    if depth > 0: busy += t - last
    if depth == 1: single += t - last
    depth += d; last = t
wall = t_end - lo
print(f"window {wall / 1e6:.2f} ms: busy {busy / wall:.3f}, exactly one kernel resident {single / wall:.3f}, idle {1 - busy / wall:.3f}")
