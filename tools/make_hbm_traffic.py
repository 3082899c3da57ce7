"""profiles/hbm_traffic.json from tools/pmc_mem_summary.py JSONs, one per arithmetic mode: HBM bytes per launch
(2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md) of every conv kernel, averaged over the
kernel's launches (weighted by launch count: a kernel that runs four layers of a step is the mean of those four), keyed
the way bench.py names kernels, stamped with the digest of the kernel sources it was measured on (bench.py reports
`traffic: null` for another digest) and the commit (passed in: there is no git on the GPU box).

    python tools/make_hbm_traffic.py profiles/hbm_traffic.json <git head> f8=gpurun_out/r03/pmc_mem_f8.json h3=... f32=...
"""
import json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from brushstroke_engine_amd import build

dst, head = sys.argv[1], sys.argv[2]
modes, per_grid = {}, {}
for arg in sys.argv[3:]:
    mode, src = arg.split("=", 1)
    if not os.path.exists(src):
        continue
    data = json.load(open(src))
    detail = {}
    for kname, recs in data.items():
        m = re.match(r"(?:void )?(modconv3x3_up[12]_h3_kernel|modconv3x3_up2[vw]_kernel|modconv3x3_up1_small_h3_kernel|modconv_small_chain_kernel|modconv3x3_up[12]_kernel)(<[^>]*>)?", kname)
        if not m:
            continue
        key = m.group(1)
        if key == "modconv3x3_up1_h3_kernel" and m.group(2):
            key += "<%s>" % m.group(2)[1:].split(",")[0].strip()          # bench.py keys the up=1 kernel by its MW parameter
        elif key in ("modconv3x3_up1_kernel", "modconv3x3_up2_kernel") and m.group(2):
            key += m.group(2)                            # fp32 kernels: the full variant name (nb_modconv3x3_variant)
        for r in recs:
            if "hbm_read_mb" in r and "hbm_write_mb" in r:
                detail.setdefault(key, []).append({"grid": r["grid"], "launches": r.get("launches", 1),
                                                   "read_mb": round(r["hbm_read_mb"], 1), "write_mb": round(r["hbm_write_mb"], 1),
                                                   "mb": round(r["hbm_read_mb"] + r["hbm_write_mb"], 1),
                                                   "l2_hit": round(r.get("l2_hit", float("nan")), 3)})
    modes[mode] = {k: int(sum(d["mb"] * d["launches"] for d in lst) / sum(d["launches"] for d in lst) * 1e6) for k, lst in detail.items()}
    per_grid[mode] = detail
out = {"modes": modes,
       "_stamp": {"source_digest": build.source_digest(), "git_head": head,
                  "what": "per arithmetic mode and kernel: launch-count-weighted mean of (2 x FETCH_SIZE + WRITE_SIZE) per launch over "
                          "separate rocprofv3 --pmc passes of `bench.py --modes primary --conv-mode <mode> --steps 2` (batch 32 at R=256 "
                          "runs as one chain: every launch carries the whole batch of 32 patches)",
                  "per_grid": per_grid}}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(modes))
