#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for o in 0 2; do
  export DN_CONV_PLAIN_ORDER=$o
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/ab_$o/$c -o $c -- python3 bench.py --steps 3 --warmup 2 --eager --chains 1 --no-cpu-baseline --no-roofline --no-latency --model ssd512_vgg16 --batch 32 > gpurun_out/ab/log_$o_$c.txt 2>&1
  done
  PMC_CMD="ab" python3 tools/pmc_traffic.py /tmp/ab_$o gpurun_out/ab/traffic_$o.json > /dev/null
  python3 - $o <<'PY'
import json,sys
d=json.load(open('gpurun_out/ab/traffic_%s.json'%sys.argv[1]))
for k,v in d.items():
    if isinstance(v,dict):
        for kk,vv in v.items():
            if 'conv_halo' in kk: print(sys.argv[1],kk,vv)
PY
  python3 bench.py --model ssd512_vgg16 --batch 32 --no-cpu-baseline --no-latency 2>&1 | tail -1 | cut -c1-260
done
