cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2x
timeout 1200 python3 -m pytest tests -m gpu -q > gpurun_out/r2x/pytest.txt 2>&1; tail -3 gpurun_out/r2x/pytest.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python3 bench.py > gpurun_out/r2x/bench_default.json 2> gpurun_out/r2x/bench_default.err; tail -c 600 gpurun_out/r2x/bench_default.json; echo
