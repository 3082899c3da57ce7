"""Per (kernel, grid) mean duration per encoder pass over the last NB_STEPS passes of a rocprofv3 kernel trace of tools/trace_encoder.py."""
import csv, glob, sys, re, os
import numpy as np
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
steps = int(os.environ.get("NB_STEPS", "40"))
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a pass starts with the stem kernel or, for f8 batches, with the launch that computes the stem itself (enc_conv3x3_h3_kernel<2, 5, 1, true, true>)
first = lambda r: "stem" in r["Kernel_Name"] or re.search(r"enc_conv3x3_h3_kernel<2, 5, 1, (true|1), (true|1)>", r["Kernel_Name"]) is not None
stems = [i for i, r in enumerate(rows) if first(r)]
rows = rows[stems[-steps]:]
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
acc, order = {}, []
for r in rows:
    name = re.sub(r"\(.*", "", re.sub(r"^void ", "", r["Kernel_Name"]))
    g = tuple(int(r["Grid_Size_" + a]) // max(1, int(r["Workgroup_Size_" + a])) for a in "XYZ")
    if (name, g) not in acc: order.append((name, g))
    acc.setdefault((name, g), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print(f"{steps} passes, span {span / steps:.1f} us/pass, sum of kernel durations {sum(sum(v) for v in acc.values()) / 1e3 / steps:.1f} us/pass")
for k in order:
    v = acc[k]
    print(f"{sum(v) / 1e3 / steps:8.1f} us/pass  {len(v) / steps:5.2f} x {np.mean(v) / 1e3:7.1f} us  grid {k[1]}  {k[0][:80]}")
