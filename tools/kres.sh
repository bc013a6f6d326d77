#!/bin/bash
# register / spill / occupancy table of the kernels in one csrc file:  tools/kres.sh pnrf_mlp_kernels.hip [filter]
cd "$(dirname "$0")/.." && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed -I include \
  -c pronerf_amd/csrc/$1 -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur=None; rows={}
for l in sys.stdin:
    m=re.search(r'remark: [^ ]+ +(?:Function )?Name: (\S+)',l) or re.search(r'Name: (\S+)',l)
    if m: cur=m.group(1); rows[cur]={}; continue
    m=re.search(r'(VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill): (\d+)',l)
    if m and cur: rows[cur][m.group(1).split(' [')[0]]=int(m.group(2))
import subprocess
for k,v in rows.items():
    name=subprocess.run(['c++filt',k],capture_output=True,text=True).stdout.strip()[:70]
    if len(sys.argv)>1 and sys.argv[1] not in name: continue
    print(f\"{name:70s} V{v.get('VGPRs')} A{v.get('AGPRs')} S{v.get('TotalSGPRs')} scr{v.get('ScratchSize')} occ{v.get('Occupancy')} vspill{v.get('VGPRs Spill')} sspill{v.get('SGPRs Spill')}\")
" $2
