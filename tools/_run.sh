cd $GRAFT_REPO_ROOT
for l in 1 0 1 0; do
DN_DW_LDS=$l timeout 200 python bench.py --no-cpu-baseline --no-roofline --steps 400 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('DN_DW_LDS=$l', d['value'], d['ms_per_step'], d.get('one_at_a_time'))"
done
