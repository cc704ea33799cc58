#!/bin/bash
# dev tool: rocprofv3 kernel stats of a short bench run -> gpurun_out/<tag>/stats_<name>.txt
#   usage: tools/kstats.sh <tag> <name> [env K=V ...] -- [bench args]
TAG=$1; NAME=$2; shift 2
ENVS=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do ENVS+=("$1"); shift; done
shift
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
for kv in "${ENVS[@]}"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$NAME -o p -- python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-latency --no-configs "$@" > $OUT/log_$NAME.txt 2>&1
python3 - <<PY
import csv,re,glob,json
f=glob.glob('/tmp/ks_$NAME/**/p_kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=0
out=open('$OUT/stats_$NAME.txt','w')
for r in rows:
    n=re.sub(r'\(anonymous namespace\)::','',r['Name']); n=re.sub(r'\(.*','',n).replace('void ','')
    calls=int(r['Calls']); avg=float(r['AverageNs'])/1e3; t=float(r['TotalDurationNs'])/1e3
    if calls < 40 or 'copyBuffer' in n: continue
    tot+=t/50
    out.write(f"{n[:60]:60s} calls/step {calls/50:5.1f} avg {avg:7.1f} min {float(r['MinNs'])/1e3:6.1f} us  per-step {t/50:7.1f} us\n")
out.write(f"sum per step {tot:.1f} us\n")
try:
    d=json.loads([l for l in open('$OUT/log_$NAME.txt').read().splitlines() if l.startswith('{')][-1])
    out.write(f"bench: {d['value']} img/s {d['ms_per_step']} ms/step\n")
except Exception as e:
    out.write(f"bench line unreadable: {e}\n")
PY
rm -rf /tmp/ks_$NAME
cat $OUT/stats_$NAME.txt
