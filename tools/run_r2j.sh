cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2j
timeout 900 python3 -m pytest tests -m gpu -q -s -k "map_on_fixed or v2_at_300 or heads_match or other_model" > gpurun_out/r2j/pytest_sel.txt 2>&1; grep -E "mAP|max\|err|passed|failed|ground truth" gpurun_out/r2j/pytest_sel.txt
timeout 900 python3 tools/layer_errors.py ssdlite320_mobilenet_v3_large ssd_lite_mobilenet_v2 ssd_lite_mobilenet_v2:300 ssd300_vgg16 ssd512_vgg16 --out gpurun_out/r2j/layer_errors.txt > gpurun_out/r2j/layer_errors.log 2>&1
grep -E "^##|^head|logits: max err" gpurun_out/r2j/layer_errors.txt
B="python3 bench.py --no-cpu-baseline --no-roofline --no-latency --steps 100 --warmup 20"
for i in 1 2; do for bs in 64 32; do
  $B --batch $bs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $bs', d['value'], d['ms_per_step'])" >> gpurun_out/r2j/sweep.txt
done; done
cat gpurun_out/r2j/sweep.txt
