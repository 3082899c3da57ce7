"""Aggregate gpurun_out/train_trace_summary.txt (tools/trace_train.sh) by kernel name over all grids."""
import re, collections
rows = [l for l in open("gpurun_out/train_trace_summary.txt")][1:]
agg = collections.defaultdict(lambda: [0.0, 0.0])
for l in rows:
    m = re.match(r"\s*([\d.]+) ms/it\s+([\d.]+) x\s+([\d.]+) us\s+grid \(([^)]*)\)\s+(.*)", l)
    if not m:
        continue
    ms, calls, name = float(m[1]), float(m[2]), m[5]
    lib = name.startswith("at::") or "Cijk" in name or "rocclr" in name
    key = ("torch: " + name[:70]) if lib else name[:48]
    agg[key][0] += ms; agg[key][1] += calls
own = sum(v[0] for k, v in agg.items() if not k.startswith("torch: "))
lib = sum(v[0] for k, v in agg.items() if k.startswith("torch: "))
print(f"own kernels {own:.2f} ms/it, library (torch / hipBLASLt / runtime fills) {lib:.2f} ms/it, "
      f"{sum(v[1] for k, v in agg.items() if k.startswith('torch: ')):.0f} library launches/it")
for k, (ms, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:50]:
    print(f"{ms:7.2f} ms/it {c:7.1f} calls  {k}")
