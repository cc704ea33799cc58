cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2n
timeout 1200 python3 -m pytest tests -m gpu -q -s > gpurun_out/r2n/pytest.txt 2>&1; grep -E "mAP|passed|failed|FAILED|ground truth|classes whose" gpurun_out/r2n/pytest.txt
bash tools/kstats.sh r2n vgg512 -- --model ssd512_vgg16 --batch 32 | grep -E "select|merge|tau|softmax|sum|bench"
python3 bench.py --no-cpu-baseline --no-roofline --no-latency --steps 30 --warmup 5 --model ssd512_vgg16 --batch 32 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ssd512', d['value'], d['ms_per_step'])"
python3 bench.py --no-cpu-baseline --no-roofline --no-latency --steps 100 --warmup 10 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('v3', d['value'], d['ms_per_step'])"
