cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2z
timeout 600 python3 -m pytest tests/test_loss.py -m gpu -q > gpurun_out/r2z/pytest_loss.txt 2>&1; tail -5 gpurun_out/r2z/pytest_loss.txt
