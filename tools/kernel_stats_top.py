"""Top kernels of a rocprofv3 --kernel-trace --stats run:  python tools/kernel_stats_top.py <dir with *kernel_stats.csv> [n]"""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time over the run: {tot / 1e6:.2f} ms")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[: int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    print(f"{float(r['TotalDurationNs']) / 1e6:9.2f} ms  {int(r['Calls']):6d} calls  avg {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:100]}")
