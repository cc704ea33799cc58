cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2z
timeout 1200 python3 -m pytest tests -m gpu -q > gpurun_out/r2z/pytest.txt 2>&1; tail -3 gpurun_out/r2z/pytest.txt
