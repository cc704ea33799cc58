#!/bin/bash
# dev tool: VGPRs / scratch / occupancy of every kernel of one csrc file (cross-compiles; no GPU needed).   usage: tools/kres.sh depthwise [filter]
cd "$(dirname "$0")/.."
b=$1; extra=""
case $b in pwdirect|pointwise|tail|depthwise) extra="-mllvm -amdgpu-mfma-vgpr-form=1";; postprocess) extra="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-gpu-rdc $extra -c demonet_amd/csrc/$b.hip -o /tmp/kres_$b.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re,subprocess
cur=None; rows={}
for l in sys.stdin:
    m=re.search(r' Name: (\S+)',l)
    if m: cur=m.group(1); rows[cur]={}
    for key,nm in (('VGPRs','vgpr'),('AGPRs','agpr'),('SGPRs','sgpr'),('ScratchSize \[bytes/lane\]','scratch'),('Occupancy \[waves/SIMD\]','occ'),('LDS Size \[bytes/block\]','lds')):
        m=re.search(key+r': (\d+)',l)
        if m and cur: rows[cur][nm]=int(m.group(1))
names=subprocess.run(['c++filt'],input='\n'.join(rows),capture_output=True,text=True).stdout.split('\n')
flt=sys.argv[1] if len(sys.argv)>1 else ''
for (k,v),d in zip(rows.items(),names):
    d=d.replace('(anonymous namespace)::','').replace('void ','')
    d=d.split('(')[0]
    if flt in d: print('%-60s'%d[:60], v)
" "$2"
rm -f /tmp/kres_$b.o
