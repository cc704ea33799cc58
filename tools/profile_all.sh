#!/bin/bash
# Regenerate everything under profiles/ that comes from the GPU box (run through gpurun, then copy from gpurun_out/):
#   gpurun -- 'bash tools/profile_all.sh'
cd $GRAFT_REPO_ROOT
bash tools/round_profile.sh r02 > /dev/null 2>&1
bash tools/round_profile.sh r02_vgg512 --model ssd512_vgg16 --batch 32 --steps 20 --warmup 5 > /dev/null 2>&1
bash tools/round_profile.sh r02_vgg300 --model ssd300_vgg16 --batch 64 --steps 20 --warmup 5 > /dev/null 2>&1
bash tools/valu.sh r02 valu --batch 64 > /dev/null 2>&1
timeout 900 python3 tools/layer_errors.py ssdlite320_mobilenet_v3_large ssd_lite_mobilenet_v2 ssd_lite_mobilenet_v2:300 ssd300_vgg16 ssd512_vgg16 --out gpurun_out/r02/layer_errors.txt > gpurun_out/r02/layer_errors.log 2>&1
mkdir -p gpurun_out/r02b
python3 bench.py > gpurun_out/r02b/bench.json 2> gpurun_out/r02b/bench.err
python3 bench.py --model ssd512_vgg16 --batch 32 --steps 20 --warmup 5 > gpurun_out/r02b/vgg512_bench.json 2>/dev/null
python3 bench.py --model ssd300_vgg16 --batch 64 --steps 20 --warmup 5 > gpurun_out/r02b/vgg300_bench.json 2>/dev/null
python3 bench.py --model ssd_lite_mobilenet_v2 --image-size 300 --batch 128 --steps 20 --warmup 5 > gpurun_out/r02b/v2_300_bench.json 2>/dev/null
python3 bench.py --batch 32 > gpurun_out/r02b/b32_bench.json 2>/dev/null
rm -f gpurun_out/r02b/batch_sweep.txt
for bs in 1 8 16 32 64 128 256; do for fl in 3 1; do python3 bench.py --no-cpu-baseline --no-roofline --no-latency --steps 200 --warmup 20 --batch $bs --inflight $fl 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $bs  forwards in flight $fl  %.1f img/s  %.4f ms/step' % (d['value'], d['ms_per_step']))" >> gpurun_out/r02b/batch_sweep.txt; done; done
cat gpurun_out/r02b/batch_sweep.txt
