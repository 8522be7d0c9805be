#!/bin/bash
# per-kernel register / LDS footprint (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel
cd "$(dirname "$0")/../graphsage-simple_amd/csrc"
for f in ${@:-sage_gather sage_dense sage_fused sage_sample sage_pipeline sage_linear sage_backward}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -c $f.hip -o /tmp/kr_$f.o -Rpass-analysis=kernel-resource-usage 2>&1 \
  | python3 -c "
import sys,re,subprocess
cur=None;rows=[]
for ln in sys.stdin:
    m=re.search(r'Function Name: (\S+)',ln)
    if m: cur={'name':m.group(1)}; rows.append(cur); continue
    m=re.search(r'remark:\s+([A-Za-z][\w ]*?)\s*(?:\[[^\]]*\])?: (\d+)',ln)
    if m and cur is not None: cur[m.group(1)]=int(m.group(2))
for r in rows:
    n=subprocess.run(['/usr/bin/c++filt',r['name']],capture_output=True,text=True).stdout.strip()
    n=re.sub(r'\(anonymous namespace\)::','',n); n=re.sub(r'\(.*','',n)[:70]
    print('%-72s VGPR %3d AGPR %3d SGPR %3d LDS %6d occ %d scratch %d'%(n,r.get('VGPRs',-1),r.get('AGPRs',-1),r.get('TotalSGPRs',-1),r.get('LDS Size',-1),r.get('Occupancy',-1),r.get('ScratchSize',-1)))
"
done
