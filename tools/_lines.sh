mkdir -p gpurun_out/lines
python3 bench.py > gpurun_out/lines/bench.json 2> gpurun_out/lines/bench.err
python3 bench.py --batch 32 > gpurun_out/lines/b32_bench.json 2>/dev/null
python3 bench.py --model ssd512_vgg16 --batch 32 --steps 20 --warmup 5 > gpurun_out/lines/vgg512_bench.json 2>/dev/null
python3 bench.py --model ssd300_vgg16 --batch 64 --steps 20 --warmup 5 > gpurun_out/lines/vgg300_bench.json 2>/dev/null
python3 bench.py --model ssd_lite_mobilenet_v2 --image-size 300 --batch 128 --steps 20 --warmup 5 > gpurun_out/lines/v2_300_bench.json 2>/dev/null
python3 bench.py --input u8 --no-cpu-baseline --no-roofline > gpurun_out/lines/u8.json 2>/dev/null
python3 bench.py --weights worstcase --no-cpu-baseline --no-roofline > gpurun_out/lines/worst.json 2>/dev/null
DN_BENCH_FORCE_DIST=1 python3 bench.py --no-cpu-baseline --no-roofline --batch 32 > gpurun_out/lines/dist32.json 2>/dev/null
