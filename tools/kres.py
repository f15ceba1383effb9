#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks: one line per kernel (demangled, shortened).
usage: hipcc ... -Rpass-analysis=kernel-resource-usage 2>&1 | python tools/kres.py [filter]"""
import re, subprocess, sys
flt = sys.argv[1] if len(sys.argv) > 1 else ""
cur = {}
rows = []
for line in sys.stdin:
    if "error" in line and "remark" not in line:
        print(line.rstrip())
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|SGPRs Spill|LDS Size \[bytes/block\]):\s+(\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    else:
        cur[k] = v
names = [r["name"] for r in rows]
if names:
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
else:
    dem = []
for r, d in zip(rows, dem):
    d = d.replace("pfa::", "").replace("radix_list", "rl")
    if flt and flt not in d:
        continue
    print("vgpr %-4s sgpr %-4s occ %-2s spill %-3s scratch %-4s %s" % (r.get("VGPRs"), r.get("SGPRs"), r.get("Occupancy [waves/SIMD]"), r.get("VGPRs Spill"), r.get("ScratchSize [bytes/lane]"), d[:150]))
