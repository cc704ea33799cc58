cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
timeout 200 python bench.py --no-cpu-baseline --steps 400 --per-op gpurun_out/po.txt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'], d.get('one_at_a_time'), d['kernels']['dw_kernel'])"
done
grep "k5s" gpurun_out/po.txt | cut -c1-8,60-175
timeout 200 python bench.py --no-cpu-baseline --no-roofline --steps 400 --batch 32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('batch 32', d['value'], d['ms_per_step'], d.get('one_at_a_time'))"
