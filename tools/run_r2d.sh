cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
timeout 900 python3 -m pytest tests -m gpu -q > gpurun_out/r2d/pytest.txt 2>&1
tail -5 gpurun_out/r2d/pytest.txt
bash tools/kstats.sh r2d x1_b32s1 DN_XCD=1 DN_SPLIT=1 -- --batch 32 > /dev/null
bash tools/kstats.sh r2d x0_b32s1 DN_XCD=0 DN_SPLIT=1 -- --batch 32 > /dev/null
bash tools/kstats.sh r2d x1_b64 DN_XCD=1 -- --batch 64 > /dev/null
bash tools/kstats.sh r2d x0_b64 DN_XCD=0 -- --batch 64 > /dev/null
timeout 600 python3 tools/layer_errors.py --out gpurun_out/r2d/layer_errors.txt > gpurun_out/r2d/layer_errors.log 2>&1
tail -3 gpurun_out/r2d/layer_errors.log
