mkdir -p gpurun_out/lines6
python3 bench.py > gpurun_out/lines6/bench.json 2> gpurun_out/lines6/bench.err
python3 bench.py --batch 32 > gpurun_out/lines6/b32_bench.json 2>/dev/null
python3 bench.py --model ssd512_vgg16 --batch 32 --steps 40 --warmup 5 > gpurun_out/lines6/vgg512_bench.json 2>/dev/null
python3 bench.py --model ssd300_vgg16 --batch 64 --steps 40 --warmup 5 > gpurun_out/lines6/vgg300_bench.json 2>/dev/null
python3 bench.py --model ssd_lite_mobilenet_v2 --image-size 300 --batch 128 --steps 100 --warmup 10 > gpurun_out/lines6/v2_300_bench.json 2>/dev/null
