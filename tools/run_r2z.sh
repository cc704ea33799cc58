cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2z
timeout 900 python3 -m pytest tests/test_gpu_model.py -m gpu -q -x -k "postprocess or golden or cap or packed or num_classes" > gpurun_out/r2z/pytest_m.txt 2>&1; tail -3 gpurun_out/r2z/pytest_m.txt
B="python3 bench.py --no-cpu-baseline --no-roofline --no-latency --steps 100 --warmup 20"
for bs in 64 64 32 32 16; do $B --batch $bs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $bs', d['value'], d['ms_per_step'])"; done
bash tools/kstats.sh r2z b16 -- --batch 16 | grep -E "merge|sum"
