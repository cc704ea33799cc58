cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2h
timeout 600 python3 -m pytest tests -m gpu -q > gpurun_out/r2h/pytest.txt 2>&1; tail -4 gpurun_out/r2h/pytest.txt
B="python3 bench.py --no-cpu-baseline --no-roofline --no-latency --steps 100 --warmup 20"
for kv in "DN_SPLIT=2" "DN_SPLIT=4" "DN_SPLIT=3" "DN_SPLIT=2 DN_GRAPH_BRANCHES=1" "DN_SPLIT=1"; do
  for bs in 64 32 128; do
    echo "== $kv batch $bs" >> gpurun_out/r2h/sweep.txt
    env $kv $B --batch $bs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['launch'])" >> gpurun_out/r2h/sweep.txt
  done
done
cat gpurun_out/r2h/sweep.txt
timeout 600 python3 tools/layer_errors.py ssd_lite_mobilenet_v2:300 --out gpurun_out/r2h/layer_errors_v2_300.txt > gpurun_out/r2h/layer_errors.log 2>&1
tail -22 gpurun_out/r2h/layer_errors_v2_300.txt
