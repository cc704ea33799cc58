cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2u
timeout 900 python3 -m pytest tests -m gpu -q -s -x -k "se_fold or heads_match or xcd_grouping or sub_batch" > gpurun_out/r2u/pytest_se.txt 2>&1; grep -E "SE fold|passed|failed|Error|error" gpurun_out/r2u/pytest_se.txt | head
B="python3 bench.py --no-cpu-baseline --no-roofline --no-latency --steps 100 --warmup 20"
for i in 1 2; do for kv in "DN_SE_FOLD=1" "DN_SE_FOLD=0"; do for bs in 64 32; do
  env $kv $B --batch $bs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$kv batch $bs', d['value'], d['ms_per_step'])"
done; done; done
DN_BENCH_FORCE_DIST=1 $B 2>gpurun_out/r2u/dist.err | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('dist path world 1 batch 64', d['value'], d['ms_per_step'])"; grep -v "^[EWI]2026" gpurun_out/r2u/dist.err | tail -3
