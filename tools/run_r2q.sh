cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2q
timeout 1500 python3 -m pytest tests -m gpu -q -s -k "map_on_fixed" > gpurun_out/r2q/pytest.txt 2>&1; grep -E "mAP|passed|failed|classes whose" gpurun_out/r2q/pytest.txt | cut -c1-600 | head
DN_SPLIT=1 timeout 300 python3 tools/probe_pp_fast.py > gpurun_out/r2q/pp_fast.txt 2>&1; tail -12 gpurun_out/r2q/pp_fast.txt
