cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
python3 bench.py > gpurun_out/r02b/bench.json 2> gpurun_out/r02b/bench.err; tail -c 1800 gpurun_out/r02b/bench.json; echo
python3 bench.py --model ssd512_vgg16 --batch 32 --steps 20 --warmup 5 > gpurun_out/r02b/vgg512_bench.json 2>/dev/null
python3 bench.py --model ssd300_vgg16 --batch 64 --steps 20 --warmup 5 > gpurun_out/r02b/vgg300_bench.json 2>/dev/null
python3 bench.py --model ssd_lite_mobilenet_v2 --image-size 300 --batch 128 --steps 20 --warmup 5 > gpurun_out/r02b/v2_300_bench.json 2>/dev/null
python3 bench.py --batch 32 > gpurun_out/r02b/b32_bench.json 2>/dev/null
for f in vgg512 vgg300 v2_300 b32; do tail -c 300 gpurun_out/r02b/${f}_bench.json | head -c 300; echo; done
