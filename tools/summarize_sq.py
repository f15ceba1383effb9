#!/usr/bin/env python3
"""Per-kernel means of the SQ counter passes of tools/pmc_arith.sh + derived figures.
usage: summarize_sq.py <dir with g*/.../*_counter_collection.csv and *_kernel_trace.csv> <tag>

Units (MI355X_MICROARCH.md, cycle constants): SQ_BUSY_CYCLES, SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count
quad-cycles (x 4 = shader cycles) summed over all SEs / waves; GRBM_GUI_ACTIVE counts cycles summed over the 8 XCDs.
Derived:
  valu_issue_util  = 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8
  waves_per_simd   = 4 * SQ_WAVE_CYCLES / (1024 * kernel cycles)          (time-averaged resident waves)
  wait / issue-stall / active shares of a wave's lifetime = SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES
  lds_conflict     = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
"""
import collections, csv, glob, sys
root, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pfa::" in r["Kernel_Name"] or "pfft" in r["Kernel_Name"]:
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size", "LDS_Block_Size", "Workgroup_Size", "Grid_Size"):
                if k in r and r[k] != "":
                    agg[r["Kernel_Name"]][k].append(float(r[k]))
dur = collections.defaultdict(list)
for f in glob.glob(root + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pfa::" in r["Kernel_Name"] or "pfft" in r["Kernel_Name"]:
            dur[r["Kernel_Name"]].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3)
print("== %s" % tag)
for k, cs in sorted(agg.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    name = k.replace("pfa::", "")
    print(name[:170])
    print("   launches %d, mean duration %.1f us (profiled passes run at a lower clock: guide, DVFS item 2)" % (len(dur.get(k, [])) , (sum(dur[k]) / len(dur[k])) if dur.get(k) else float("nan")))
    print("   vgpr %d agpr %d sgpr %d scratch %d B lds %d B wg %d grid %d" % tuple(int(m.get(x, 0)) for x in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size", "LDS_Block_Size", "Workgroup_Size", "Grid_Size")))
    raw = ["%s=%.4g" % (c, m[c]) for c in sorted(m) if c.startswith("SQ_") or c.startswith("GRBM")]
    print("   " + "  ".join(raw))
    if m.get("GRBM_GUI_ACTIVE") and m.get("SQ_WAVE_CYCLES"):
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0
        simds = 1024.0
        d = {}
        d["valu_issue_util"] = 4.0 * m.get("SQ_ACTIVE_INST_VALU", 0) / (simds * cyc)
        d["waves_per_simd"] = 4.0 * m["SQ_WAVE_CYCLES"] / (simds * cyc)
        for nm, c in (("wait_share", "SQ_WAIT_ANY"), ("issue_stall_share", "SQ_WAIT_INST_ANY"), ("active_share", "SQ_ACTIVE_INST_ANY"),
                      ("valu_share_of_lifetime", "SQ_ACTIVE_INST_VALU"), ("lds_share_of_lifetime", "SQ_ACTIVE_INST_LDS")):
            if c in m:
                d[nm] = m[c] / m["SQ_WAVE_CYCLES"]
        if m.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_conflict"] = m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_LDS_IDX_ACTIVE"]
        d["valu_insts_per_wave"] = m.get("SQ_INSTS_VALU", 0) / max(m.get("SQ_WAVES", 1), 1)
        d["kernel_cycles"] = cyc
        print("   derived: " + "  ".join("%s=%.3g" % kv for kv in d.items()))
