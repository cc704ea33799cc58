cd $GRAFT_REPO_ROOT
B="python3 bench.py --no-cpu-baseline --no-roofline --no-latency --steps 100 --warmup 20"
run() { for bs in 64 64 32 256; do env "$@" $B --batch $bs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', 'batch $bs', d['value'], d['ms_per_step'])"; done; }
for c in 0 5 6 8 9; do run DN_PW_GROUP_TILE=$c; done
