"""Per-kernel memory-side summary of rocprofv3 --pmc passes (TCC_HIT/MISS, FETCH_SIZE, WRITE_SIZE), averaged per
dispatch and grouped by (kernel, grid).  FETCH_SIZE is doubled (gfx950 correction, MI355X_MICROARCH.md "HBM").

    python tools/pmc_mem_summary.py gpurun_out/pmc_hit gpurun_out/pmc_fetch gpurun_out/pmc_write [out.json]
"""
import collections
import csv
import glob
import json
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in sys.argv[1:4]:
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        meta = {}
        for r in csv.DictReader(open(f)):
            key = (r["Kernel_Name"], r["Grid_Size"], int(r["Dispatch_Id"]))
            per[key][r["Counter_Name"]] += float(r["Counter_Value"])
            meta[key] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        for (k, g, _), v in per.items():
            for c, val in v.items():
                rows[(k, g)][c].append(val)
            dur[(k, g)].append(meta[(k, g, _)])
out = {}
for (k, g), v in sorted(rows.items(), key=lambda t: -sum(dur[t[0]])):
    if not any(s in k for s in ("modconv", "pack_h2", "torgb", "enc_", "canvas", "noise", "styles")):
        continue
    m = {c: sum(x) / len(x) for c, x in v.items()}
    hit, miss = m.get("TCC_HIT_sum"), m.get("TCC_MISS_sum")
    fetch = m.get("FETCH_SIZE")
    write = m.get("WRITE_SIZE")
    line = f"{k[:60]:60s} grid {g:>9s} {sum(dur[(k, g)]) / len(dur[(k, g)]) / 1e3:8.1f}us"
    rec = {"grid": g, "launches": len(dur[(k, g)])}
    if hit is not None and miss is not None and hit + miss > 0:
        line += f"  L2 hit {hit / (hit + miss) * 100:5.1f}%"
        rec["l2_hit"] = hit / (hit + miss)
    if fetch is not None:
        line += f"  read {2 * fetch / 1e3:8.1f} MB"          # FETCH_SIZE is in KB
        rec["hbm_read_mb"] = 2 * fetch / 1e3
    if write is not None:
        line += f"  write {write / 1e3:8.1f} MB"
        rec["hbm_write_mb"] = write / 1e3
    print(line)
    out.setdefault(k, []).append(rec)
if len(sys.argv) > 4:
    json.dump(out, open(sys.argv[4], "w"), indent=1)
