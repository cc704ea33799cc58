cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2p
timeout 1500 python3 -m pytest tests -m gpu -q -s -k "map_on_fixed" > gpurun_out/r2p/pytest.txt 2>&1; grep -E "mAP|passed|failed|classes whose|DEBUG" gpurun_out/r2p/pytest.txt | cut -c1-3000 | head -30
timeout 900 python3 -m pytest tests -m gpu -q -x -k "other_model or vgg or full_size or dense" > gpurun_out/r2p/pytest_vgg.txt 2>&1; tail -3 gpurun_out/r2p/pytest_vgg.txt
for m in "ssd512_vgg16 --batch 32" "ssd300_vgg16 --batch 64"; do
python3 bench.py --no-cpu-baseline --no-roofline --no-latency --steps 30 --warmup 5 --model $m 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$m', d['value'], d['ms_per_step'])"
DN_CONV_POOL=0 python3 bench.py --no-cpu-baseline --no-roofline --no-latency --steps 30 --warmup 5 --model $m 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('nopoolfuse $m', d['value'], d['ms_per_step'])"
done
