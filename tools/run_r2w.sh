cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2w
for t in 0 48 88; do echo "== DN_EXPDW_TILE=$t" >> gpurun_out/r2w/expdw_proj.txt; DN_EXPDW_TILE=$t timeout 300 python3 tools/probe_expdw_proj.py 32 2>&1 | grep -v amdgpu >> gpurun_out/r2w/expdw_proj.txt; done
cat gpurun_out/r2w/expdw_proj.txt
B="python3 bench.py --no-cpu-baseline --no-roofline --no-latency --steps 100 --warmup 20"
for t in 0 48 88; do for bs in 64 32; do
  DN_EXPDW_TILE=$t $B --batch $bs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('tile $t batch $bs', d['value'], d['ms_per_step'])"
done; done
