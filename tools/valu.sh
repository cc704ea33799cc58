#!/bin/bash
# dev tool: per-kernel dynamic instruction counts of one eager bench pass (rocprofv3 --pmc, kernel-trace only).
# The chain's big kernels are VALU-issue bound (a wave64 VALU instruction occupies its SIMD for 4 cycles), so
#   valu_us = SQ_INSTS_VALU * 4 cycles / 1024 SIMDs / 2.4 GHz
# is the floor of a launch however the latencies overlap.
#   usage: tools/valu.sh <tag> <name> [bench args]   -> gpurun_out/<tag>/valu_<name>.txt
TAG=$1; NAME=$2; shift 2
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
rm -rf /tmp/valu_$NAME
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --kernel-trace --output-format csv -d /tmp/valu_$NAME -o v -- python3 bench.py --steps 3 --warmup 2 --eager --chains 1 --no-cpu-baseline --no-roofline --no-latency --no-configs "$@" > $OUT/valu_$NAME.log 2>&1
python3 - <<PY
import csv, glob, re
acc = {}
for path in glob.glob('/tmp/valu_$NAME/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']); k = re.sub(r'\(.*', '', k).replace('void ', '').replace('.kd', '')
        if k.startswith('__amd') or 'at::' in k or 'elementwise' in k: continue
        a = acc.setdefault(k, {}); c = a.setdefault(r['Counter_Name'], [0.0, 0]); c[0] += float(r['Counter_Value']); c[1] += 1
rows = []
for k, a in acc.items():
    n = a['SQ_INSTS_VALU'][1]
    per = {c: v[0] / v[1] for c, v in a.items()}
    rows.append((per['SQ_INSTS_VALU'] * n / 5, k, n / 5, per))
tot = 0
with open('$OUT/valu_$NAME.txt', 'w') as f:
    for t, k, n, per in sorted(rows, reverse=True):
        w = max(per.get('SQ_WAVES', 1), 1)
        us = per['SQ_INSTS_VALU'] * 4 / 1024 / 2400
        tot += us * n
        f.write(f"{k[:58]:58s} launches/step {n:5.1f} waves {w:8.0f} VALU/wave {per['SQ_INSTS_VALU'] / w:7.1f} SALU/wave {per.get('SQ_INSTS_SALU', 0) / w:6.1f} LDS/wave {per.get('SQ_INSTS_LDS', 0) / w:6.1f} valu_us/launch {us:6.1f} per-step {us * n:7.1f}\n")
    f.write(f"VALU floor per step {tot:.1f} us\n")
print(open('$OUT/valu_$NAME.txt').read())
PY
