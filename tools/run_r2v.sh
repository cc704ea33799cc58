cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2v
timeout 300 python3 tools/probe_expdw_proj.py 32 > gpurun_out/r2v/expdw_proj.txt 2>&1; tail -3 gpurun_out/r2v/expdw_proj.txt
