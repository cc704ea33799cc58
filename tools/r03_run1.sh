#!/bin/bash
# round-3 GPU session 1: parity suite, headline bench, and the one-row depthwise hunt (tools/hunt_dw_rows.py)
cd $GRAFT_REPO_ROOT; O=gpurun_out/s1; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" ; tail -3 $O/pytest.txt
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
for cfg in "1 0 32 3 10" "1 1 32 3 10" "1 0 64 3 6" "0 1 32 3 6"; do
  set -- $cfg
  echo "=== DN_DW_ROWS=$1 DN_SE_SMALL=$2 batch $3 depth $4 rounds $5"
  DN_DW_ROWS=$1 DN_SE_SMALL=$2 timeout 300 python tools/hunt_dw_rows.py $3 $4 $5 2>&1 | tail -40
done > $O/hunt.txt 2>&1
cat $O/hunt.txt
