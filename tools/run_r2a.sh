cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
timeout 300 gpurun_out/xcd_probe > gpurun_out/r2a/xcd_probe.txt 2>&1
B="python3 bench.py --no-cpu-baseline --no-roofline --no-latency --steps 100 --warmup 20"
for bs in 64 32 16 128 256; do
  echo "== batch $bs" >> gpurun_out/r2a/sweep.txt
  $B --batch $bs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['launch'])" >> gpurun_out/r2a/sweep.txt
done
for kv in "DN_SPLIT=1" "DN_SPLIT=3" "DN_EXPDW_MINHW=1600" "DN_EXPDW_MINHW=400" "DN_EXPDW_MINHW=100" "DN_EXPDW=0" "DN_TAIL=0" "DN_PW_XS=0"; do
  for bs in 64 32; do
    echo "== $kv batch $bs" >> gpurun_out/r2a/sweep.txt
    env $kv $B --batch $bs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['launch'])" >> gpurun_out/r2a/sweep.txt
  done
done
echo "== split1 batch 32 forced split 2" >> gpurun_out/r2a/sweep.txt
cat gpurun_out/r2a/sweep.txt
