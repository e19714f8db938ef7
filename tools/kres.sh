#!/bin/bash
# kernel resource summary of one .hip file: name | VGPRs | spills | occupancy | SGPRs   (usage: tools/kres.sh hawq_amd/csrc/fused_wp.hip [filter])
f=$1; flt=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -Wall -Wno-unused-function \
  -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/kres.o 2>&1 | python3 -c "
import sys,re,subprocess
cur=None; rows=[]
for l in sys.stdin:
    if 'error' in l or 'warning' in l: print(l.rstrip())
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur={'name':m.group(1)}; rows.append(cur)
    for k in ('VGPRs:','VGPRs Spill:','Occupancy \[waves/SIMD\]:','TotalSGPRs:','ScratchSize \[bytes/lane\]:'):
        m=re.search(r'    '+k+r' (\d+)',l)
        if m and cur is not None: cur[k]=m.group(1)
names=subprocess.run(['/usr/bin/c++filt']+[r['name'] for r in rows],capture_output=True,text=True).stdout.split('\n')
for r,n in zip(rows,names):
    if re.search(sys.argv[1],n): print(n[:110].ljust(110), 'vgpr',r.get('VGPRs:'),'spill',r.get('VGPRs Spill:'),'occ',r.get('Occupancy \[waves/SIMD\]:'),'sgpr',r.get('TotalSGPRs:'),'scratch',r.get('ScratchSize \[bytes/lane\]:'))
" "$flt"
