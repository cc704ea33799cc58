mkdir -p gpurun_out/ab10
run() { echo "$1 | $2" >> gpurun_out/ab10/ab.txt; env $1 python bench.py $2 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-latency 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])" >> gpurun_out/ab10/ab.txt; }
for e in DN_CONV_HEAD_BIG_MIN=40 DN_CONV_HEAD_BIG_MIN=100 DN_CONV_HEAD_BIG_MIN=200 DN_CONV_HEAD_BIG_MIN=400 DN_CONV_HEAD_BIG_MIN=40; do
run $e "--model ssd512_vgg16 --batch 32"; run $e "--model ssd300_vgg16 --batch 64"
done
