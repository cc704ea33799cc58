#!/bin/bash
# dev tool: which kernels spill to scratch (spilled registers are HBM write traffic: the expand+depthwise+project kernel of the 80 x 80
# maps wrote 97 MB per launch for a 9.8 MB output until its 72-wide variant stopped being used for 64-channel blocks).
#   usage: tools/scratch_scan.sh            (cross-compiles every csrc/*.hip with -Rpass-analysis=kernel-resource-usage; no GPU needed)
cd "$(dirname "$0")/.."
for f in demonet_amd/csrc/*.hip; do
  b=$(basename $f .hip); extra=""
  case $b in pwdirect|pointwise|tail) extra="-mllvm -amdgpu-mfma-vgpr-form=1";; esac
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-gpu-rdc $extra -c $f -o /tmp/scan_$b.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur=None; rows={}
for l in sys.stdin:
    m=re.search(r' Name: (\S+)',l)
    if m: cur=m.group(1); rows[cur]={}
    for key,nm in (('VGPRs','vgpr'),('AGPRs','agpr'),('ScratchSize \[bytes/lane\]','scratch'),('Occupancy \[waves/SIMD\]','occ')):
        m=re.search(key+r': (\d+)',l)
        if m and cur: rows[cur][nm]=int(m.group(1))
bad=[(k,v) for k,v in rows.items() if v.get('scratch',0)>0]
print('$b:', len(rows),'kernels,',len(bad),'with scratch')
for k,v in bad: print('   ',k[:110],v)
"
  rm -f /tmp/scan_$b.o
done
