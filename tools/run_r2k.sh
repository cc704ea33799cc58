cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2k
timeout 1200 python3 -m pytest tests -m gpu -q -s > gpurun_out/r2k/pytest.txt 2>&1; grep -E "mAP|passed|failed|FAILED|ground truth|V2 at 300|reproduced" gpurun_out/r2k/pytest.txt
B="python3 bench.py --no-cpu-baseline --no-roofline --no-latency --steps 100 --warmup 20"
for i in 1 2; do for kv in "DN_WS_REUSE=1" "DN_WS_REUSE=0"; do for bs in 64 32; do
  env $kv $B --batch $bs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$kv batch $bs', d['value'], d['ms_per_step'])" >> gpurun_out/r2k/sweep.txt
done; done; done
cat gpurun_out/r2k/sweep.txt
