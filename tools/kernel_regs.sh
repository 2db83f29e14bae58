#!/bin/bash
# VGPRs / SGPR spills / occupancy of the engine's kernels whose name matches $1 (a grep -E pattern), from the compiler's remarks
pat=${1:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I/root/repo/include \
  -I/root/repo/ellp_amd/csrc/engine -c /root/repo/ellp_amd/csrc/engine/ellp_engine.hip -o /tmp/ellp_engine_regs.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys,re
cur=None; rows={}
for l in sys.stdin:
    m=re.search(r"Function Name: (\S+)",l)
    if m: cur=m.group(1); rows[cur]={}; continue
    m=re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)",l)
    if m and cur: rows[cur][m.group(1).strip()]=m.group(2)
import subprocess
for k,v in rows.items():
    d=subprocess.run(["c++filt",k],capture_output=True,text=True).stdout.strip()
    d=re.sub(r"\(anonymous namespace\)::","",d); d=re.sub(r"\(.*","",d)
    print("%-44s vgpr %3s agpr %3s sspill %3s vspill %3s occ %s lds %s"%(d,v.get("VGPRs"),v.get("AGPRs"),v.get("SGPRs Spill"),v.get("VGPRs Spill"),v.get("Occupancy"),v.get("LDS Size")))
' | grep -E "$pat"
