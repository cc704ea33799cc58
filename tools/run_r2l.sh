cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2l
timeout 1200 python3 -m pytest tests -m gpu -q -s -k "map_on_fixed" > gpurun_out/r2l/pytest.txt 2>&1; grep -E "mAP|passed|failed|FAILED|ground truth|classes whose" gpurun_out/r2l/pytest.txt
bash tools/kstats.sh r2l vgg512 -- --model ssd512_vgg16 --batch 32 | tail -30
for kv in "DN_PP_FAST=0" "DN_PP_WANT=2" "DN_PP_WANT=8"; do
  env $kv python3 bench.py --no-cpu-baseline --no-roofline --no-latency --steps 30 --warmup 5 --model ssd512_vgg16 --batch 32 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$kv', d['value'], d['ms_per_step'])"
done
