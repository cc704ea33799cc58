cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
python3 bench.py > gpurun_out/r02b/bench.json 2> gpurun_out/r02b/bench.err
python3 bench.py --model ssd512_vgg16 --batch 32 --steps 20 --warmup 5 > gpurun_out/r02b/vgg512_bench.json 2>/dev/null
python3 bench.py --model ssd300_vgg16 --batch 64 --steps 20 --warmup 5 > gpurun_out/r02b/vgg300_bench.json 2>/dev/null
python3 bench.py --model ssd_lite_mobilenet_v2 --image-size 300 --batch 128 --steps 20 --warmup 5 > gpurun_out/r02b/v2_300_bench.json 2>/dev/null
python3 bench.py --batch 32 > gpurun_out/r02b/b32_bench.json 2>/dev/null
rm -f gpurun_out/r02b/batch_sweep.txt
for bs in 1 8 16 32 64 128 256; do python3 bench.py --no-cpu-baseline --no-roofline --no-latency --steps 60 --warmup 10 --batch $bs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $bs  %.1f img/s  %.4f ms/step' % (d['value'], d['ms_per_step']))" >> gpurun_out/r02b/batch_sweep.txt; done
cat gpurun_out/r02b/batch_sweep.txt
