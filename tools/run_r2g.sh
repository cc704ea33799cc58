cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/queue_probe.hip -o /tmp/queue_probe 2> gpurun_out/r2g/compile.txt
timeout 300 /tmp/queue_probe 2>&1 | grep -v "branches x" > gpurun_out/r2g/queue_probe2.txt
cat gpurun_out/r2g/queue_probe2.txt
timeout 600 python3 tools/layer_errors.py --out gpurun_out/r2g/layer_errors.txt > gpurun_out/r2g/layer_errors.log 2>&1
tail -3 gpurun_out/r2g/layer_errors.log
timeout 600 python3 -m pytest tests -m gpu -q -x > gpurun_out/r2g/pytest.txt 2>&1; tail -3 gpurun_out/r2g/pytest.txt
