"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel (last dispatch) MFMA-busy fraction etc."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
meta = {}
for r in csv.DictReader(open(f)):
    key = (r["Kernel_Name"], int(r["Dispatch_Id"]))
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
    meta[key] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"], r["Grid_Size"], r["Workgroup_Size"])
last = {}
for (k, d), v in agg.items():
    if "modconv" in k or "torgb" in k or "pack_h2" in k:
        last[(k, meta[(k, d)][4])] = (d, v, meta[(k, d)])
for (k, g), (d, v, m) in sorted(last.items(), key=lambda t: -t[1][2][0]):
    dur, vg, ag, lds, grid, wg = m
    cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8
    simd = 1024
    out = f"{k[5:45]:40s} grid {grid:>9s} {dur/1e3:8.1f}us clk {cyc/dur:5.2f}GHz vgpr {vg}+{ag} lds {lds}"
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and cyc:
        out += f" mfma_busy {v['SQ_VALU_MFMA_BUSY_CYCLES']/simd/cyc*100:5.1f}%"
    if "SQ_WAVE_CYCLES" in v and v["SQ_WAVE_CYCLES"]:
        wc = v["SQ_WAVE_CYCLES"]
        out += f" waves/simd {wc*4/simd/cyc:4.2f} wait_any {v.get('SQ_WAIT_ANY',0)/wc*100:4.1f}% wait_inst {v.get('SQ_WAIT_INST_ANY',0)/wc*100:4.1f}% active {v.get('SQ_ACTIVE_INST_ANY',0)/wc*100:4.1f}%"
    for c in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS", "SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_VMEM"):
        if c in v:
            out += f" {c[3:]} {v[c]:.3g}"
    print(out)
