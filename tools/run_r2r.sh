cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2r
timeout 900 python3 -m pytest tests -m gpu -q -s -x -k "trunk_kernel" > gpurun_out/r2r/pytest_trunk.txt 2>&1; grep -E "trunk vs|passed|failed|Error|error" gpurun_out/r2r/pytest_trunk.txt | head -20
B="python3 bench.py --no-cpu-baseline --no-roofline --no-latency --steps 100 --warmup 20"
for kv in "DN_TRUNK=1" "DN_TRUNK=0" "DN_TRUNK=1 DN_SPLIT=1"; do for bs in 64 32 16; do
  env $kv timeout 300 $B --batch $bs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$kv batch $bs', d['value'], d['ms_per_step'])"
done; done
timeout 900 python3 -m pytest tests -m gpu -q -x > gpurun_out/r2r/pytest.txt 2>&1; tail -4 gpurun_out/r2r/pytest.txt
