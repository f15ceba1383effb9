#!/bin/bash
# usage: tools/kres_diff.sh <file.hip> [extra hipcc flags]: one line per kernel -- VGPRs, scratch bytes, VGPR spills, occupancy
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
python3 -c '
import re,sys,subprocess
name=None; rec={}
out=[]
for l in sys.stdin:
    m=re.search(r"Function Name: (\S+)",l)
    if m:
        name=m.group(1); rec={}; out.append((name,rec)); continue
    m=re.search(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill|Occupancy \[waves/SIMD\]): (\d+)",l)
    if m and name: rec[m.group(1).split()[0]+("S" if "Spill" in m.group(1) else "")]=int(m.group(2))
names=[n for n,_ in out]
dem=subprocess.run(["c++filt"],input="\n".join(names),capture_output=True,text=True).stdout.split("\n")
for (n,r),d in zip(out,dem):
    d=re.sub(r"pfa::wg_cfg<(\w+), pfa::radix_list<([\d, ]+)>, (\d+), (\d+)[^>]*>", lambda m:"cfg<%s %s wg%s f%s>"%(m.group(1),m.group(2).replace(", ","."),m.group(3),m.group(4)), d)
    d=d.replace("void pfa::","").replace("(pfa::xcd_args)","").replace("(pfa::strided_args)","")
    print("%-110s v%3d scr%4d vsp%3d ssp%3d occ%d"%(d[:110],r.get("VGPRs",-1),r.get("ScratchSize",-1),r.get("VGPRsS",-1),r.get("SGPRsS",-1),r.get("Occupancy",-1)))
'
