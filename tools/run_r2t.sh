cd $GRAFT_REPO_ROOT
bash tools/round_profile.sh r02 > gpurun_out/r02_log.txt 2>&1; tail -5 gpurun_out/r02_log.txt
bash tools/round_profile.sh r02_vgg512 --model ssd512_vgg16 --batch 32 --steps 20 --warmup 5 > gpurun_out/r02_vgg512_log.txt 2>&1; tail -3 gpurun_out/r02_vgg512_log.txt
bash tools/round_profile.sh r02_vgg300 --model ssd300_vgg16 --batch 64 --steps 20 --warmup 5 > gpurun_out/r02_vgg300_log.txt 2>&1; tail -3 gpurun_out/r02_vgg300_log.txt
timeout 900 python3 tools/layer_errors.py ssdlite320_mobilenet_v3_large ssd_lite_mobilenet_v2 ssd_lite_mobilenet_v2:300 ssd300_vgg16 ssd512_vgg16 --out gpurun_out/r02/layer_errors.txt > gpurun_out/r02/layer_errors.log 2>&1
timeout 900 python3 -m pytest tests -m gpu -q > gpurun_out/r02/pytest.txt 2>&1; tail -3 gpurun_out/r02/pytest.txt
B="python3 bench.py --no-cpu-baseline --no-roofline --no-latency --steps 100 --warmup 20"
for bs in 1 8 16 32 64 128 256; do
  $B --batch $bs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $bs', d['value'], d['ms_per_step'])" >> gpurun_out/r02/batch_sweep.txt
done
$B --model ssd_lite_mobilenet_v2 --image-size 300 --batch 128 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3 v2@300 batch 128', d['value'], d['ms_per_step'])" >> gpurun_out/r02/batch_sweep.txt
$B --input u8 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('u8 input batch 64', d['value'], d['ms_per_step'])" >> gpurun_out/r02/batch_sweep.txt
$B --weights worstcase 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('worstcase batch 64', d['value'], d['ms_per_step'])" >> gpurun_out/r02/batch_sweep.txt
DN_BENCH_FORCE_DIST=1 $B 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('dist path world 1 batch 64', d['value'], d['ms_per_step'])" >> gpurun_out/r02/batch_sweep.txt
cat gpurun_out/r02/batch_sweep.txt
