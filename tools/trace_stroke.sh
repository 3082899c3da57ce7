# device side of an interactive render_stroke call: kernel trace of tools/latency_stroke.py, per-kernel totals
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/strace -o s --output-format csv -- python3 tools/latency_stroke.py 2>&1 | grep "R="
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/strace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.1f} ms over 480 calls = {tot / 480 / 1e3:.0f} us per call")
for r in rows[:22]:
    print(f'{int(r["Calls"]):6d} {int(r["TotalDurationNs"]) / 480 / 1e3:7.1f} us/call  avg {float(r["AverageNs"]) / 1e3:7.1f} us  {r["Name"][:100]}')
PY
rm -rf gpurun_out/strace
