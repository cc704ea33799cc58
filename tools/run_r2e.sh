cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
timeout 300 python3 tools/probe_pw_stamps.py > gpurun_out/r2e/pw_stamps.txt 2>&1
B="python3 bench.py --no-cpu-baseline --no-roofline --no-latency --steps 100 --warmup 20"
for kv in "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "DN_XCD=1"; do
  for bs in 64 32; do
    echo "== $kv batch $bs" >> gpurun_out/r2e/sweep.txt
    env $kv $B --batch $bs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['launch'])" >> gpurun_out/r2e/sweep.txt
  done
done
cat gpurun_out/r2e/pw_stamps.txt gpurun_out/r2e/sweep.txt
