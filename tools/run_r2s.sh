cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2s
timeout 300 python3 tools/probe_trunk.py 8 > gpurun_out/r2s/trunk_stamps.txt 2>&1; tail -6 gpurun_out/r2s/trunk_stamps.txt
