cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2z
timeout 1200 python3 -m pytest tests -m gpu -q > gpurun_out/r2z/pytest.txt 2>&1; tail -3 gpurun_out/r2z/pytest.txt
B="python3 bench.py --no-cpu-baseline --no-roofline --no-latency --steps 100 --warmup 20"
for i in 1 2; do for bs in 64 32; do
  $B --batch $bs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $bs', d['value'], d['ms_per_step'])"
done; done
for m in "ssd512_vgg16 --batch 32" "ssd300_vgg16 --batch 64" "ssd_lite_mobilenet_v2 --image-size 300 --batch 128"; do
python3 bench.py --no-cpu-baseline --no-roofline --no-latency --steps 30 --warmup 5 --model $m 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$m', d['value'], d['ms_per_step'])"
done
bash tools/kstats.sh r2z b64 -- --batch 64 | tail -25
