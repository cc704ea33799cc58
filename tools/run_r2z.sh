cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2z
timeout 900 python3 -m pytest tests/test_gpu_model.py -m gpu -q -x -k "postprocess or golden or cap" > gpurun_out/r2z/pytest_m.txt 2>&1; tail -5 gpurun_out/r2z/pytest_m.txt
B="python3 bench.py --no-cpu-baseline --no-roofline --no-latency --steps 100 --warmup 20"
for bs in 64 64 32 32; do $B --batch $bs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $bs', d['value'], d['ms_per_step'])"; done
python3 bench.py --no-cpu-baseline --no-roofline --no-latency --steps 30 --warmup 5 --model ssd512_vgg16 --batch 32 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ssd512', d['value'], d['ms_per_step'])"
bash tools/kstats.sh r2z sel -- --batch 64 | grep -E "select|merge|tau|softmax|sum"
