#!/bin/bash
# dev tool: rocprofv3 kernel stats of a short bench run -> gpurun_out/qs/summary.txt   usage: tools/quick_stats.sh [bench args]
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/qs
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o qs -- python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline "$@" > $OUT/log.txt 2>&1
python3 - <<PY
import csv,re
rows=list(csv.DictReader(open('$OUT/qs_kernel_stats.csv')))
steps=50.0
out=open('$OUT/summary.txt','w')
tot=0
for r in rows:
    n=re.sub(r'\(anonymous namespace\)::','',r['Name']); n=re.sub(r'\(.*','',n).replace('void ','')
    calls=int(r['Calls']); avg=float(r['AverageNs'])/1e3; t=float(r['TotalDurationNs'])/1e3
    if calls < 40: continue
    tot+=t/steps
    out.write(f"{n[:52]:52s} calls/step {calls/steps:5.1f} avg {avg:7.1f} us  per-step {t/steps:7.1f} us\n")
out.write(f"sum per step {tot:.1f} us\n")
PY
rm -f $OUT/qs_kernel_trace.csv
cat $OUT/summary.txt
