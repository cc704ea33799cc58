#!/bin/bash
# One GPU-box pass over everything the round is judged on: the -m gpu suite, smoke(), the default bench line and the three other
# configurations, per-kernel stats of the headline configuration.      usage: gpurun -- 'bash tools/gpu_check.sh <tag>'
TAG=${1:-check}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
timeout 1200 python3 -m pytest tests -m gpu -q > gpurun_out/$TAG/pytest.txt 2>&1; tail -3 gpurun_out/$TAG/pytest.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
B="python3 bench.py --no-cpu-baseline --no-roofline --no-latency --steps 100 --warmup 20"
for i in 1 2; do for bs in 64 32; do
  $B --batch $bs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $bs', d['value'], d['ms_per_step'])"
done; done
for m in "ssd512_vgg16 --batch 32" "ssd300_vgg16 --batch 64" "ssd_lite_mobilenet_v2 --image-size 300 --batch 128"; do
python3 bench.py --no-cpu-baseline --no-roofline --no-latency --steps 30 --warmup 5 --model $m 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$m', d['value'], d['ms_per_step'])"
done
bash tools/kstats.sh $TAG b64 -- --batch 64 | tail -32
