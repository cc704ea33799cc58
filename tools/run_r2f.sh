cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/queue_probe.hip -o /tmp/queue_probe 2> gpurun_out/r2f/compile.txt
timeout 300 /tmp/queue_probe > gpurun_out/r2f/queue_probe.txt 2>&1
cat gpurun_out/r2f/queue_probe.txt
timeout 600 python3 tools/layer_errors.py --out gpurun_out/r2f/layer_errors.txt > gpurun_out/r2f/layer_errors.log 2>&1
tail -3 gpurun_out/r2f/layer_errors.log
