cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/r2c/pytest.txt 2>&1
tail -15 gpurun_out/r2c/pytest.txt
B="python3 bench.py --no-cpu-baseline --no-roofline --no-latency --steps 100 --warmup 20"
for kv in "DN_XCD=1" "DN_XCD=0"; do
  for bs in 64 32 16 128; do
    echo "== $kv batch $bs" >> gpurun_out/r2c/sweep.txt
    env $kv $B --batch $bs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['launch'])" >> gpurun_out/r2c/sweep.txt
  done
done
for kv in "DN_SPLIT=1" "DN_SPLIT=3" "DN_SPLIT=4"; do
  for bs in 64 32; do
    echo "== $kv batch $bs" >> gpurun_out/r2c/sweep.txt
    env $kv $B --batch $bs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['launch'])" >> gpurun_out/r2c/sweep.txt
  done
done
cat gpurun_out/r2c/sweep.txt
