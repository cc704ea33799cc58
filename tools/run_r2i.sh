cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2i
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/xcd_probe2.hip -o /tmp/xcd_probe2 2> gpurun_out/r2i/compile.txt
timeout 120 /tmp/xcd_probe2 > gpurun_out/r2i/xcd_probe2.txt 2>&1
cat gpurun_out/r2i/xcd_probe2.txt
