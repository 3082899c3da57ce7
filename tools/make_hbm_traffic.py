"""profiles/hbm_traffic.json from a tools/pmc_mem_summary.py JSON: mean HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, the
gfx950 correction of MI355X_MICROARCH.md) of every conv kernel's launches, keyed the way bench.py names kernels, stamped
with the digest of the kernel sources it was measured on (bench.py reports `traffic: null` for another digest).

    python tools/make_hbm_traffic.py gpurun_out/r02/pmc_mem.json profiles/hbm_traffic.json
"""
import json, os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import build

src, dst = sys.argv[1], sys.argv[2]
data = json.load(open(src))
out, detail = {}, {}
for kname, recs in data.items():
    m = re.match(r"(?:void )?(modconv3x3_up[12]_h3_kernel|modconv3x3_up1_small_h3_kernel|modconv3x3_up[12]_kernel)(<[^>]*>)?", kname)
    if not m:
        continue
    key = m.group(1)
    if key == "modconv3x3_up1_h3_kernel" and m.group(2):
        key += "<%s>" % m.group(2)[1:].split(",")[0].strip()          # bench.py keys the up=1 kernel by its MW parameter
    for r in recs:
        if "hbm_read_mb" in r and "hbm_write_mb" in r:
            detail.setdefault(key, []).append({"grid": r["grid"], "mb": round(r["hbm_read_mb"] + r["hbm_write_mb"], 1),
                                               "l2_hit": round(r.get("l2_hit", float("nan")), 3)})
for key, lst in detail.items():
    out[key] = int(sum(d["mb"] for d in lst) / len(lst) * 1e6)
head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
out["_stamp"] = {"source_digest": build.source_digest(), "git_head": head,
                 "what": "mean over the kernel's launches of (2 x FETCH_SIZE + WRITE_SIZE) per launch, separate rocprofv3 --pmc passes "
                         "of `bench.py --steps 2` (launches carry one sub-batch of 16 patches; the calibration / isolated passes add "
                         "a few 32-patch launches to the mean)", "per_grid": detail}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "_stamp"}))
