cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2z
timeout 900 python3 -m pytest tests/test_gpu_model.py -m gpu -q -x -k "other_model or vgg or full_size" > gpurun_out/r2z/pytest_m.txt 2>&1; tail -3 gpurun_out/r2z/pytest_m.txt
for pf in 0 1; do for m in "ssd512_vgg16 --batch 32" "ssd300_vgg16 --batch 64"; do for i in 1 2; do
DN_CONV_SMALL_PF=$pf python3 bench.py --no-cpu-baseline --no-roofline --no-latency --steps 30 --warmup 5 --model $m 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pf $pf $m', d['value'], d['ms_per_step'])"
done; done; done
python3 bench.py --model ssd512_vgg16 --batch 32 --steps 20 --warmup 5 --no-cpu-baseline --no-latency --per-op gpurun_out/r2z/per_op_vgg512.txt > gpurun_out/r2z/bench_vgg512.json 2>/dev/null
cut -c1-60,100-210 gpurun_out/r2z/per_op_vgg512.txt | sed -n 22,31p; tail -3 gpurun_out/r2z/per_op_vgg512.txt | cut -c1-100
