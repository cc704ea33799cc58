#!/bin/bash
# Round artifacts on the GPU box: headline bench line, per-op table, rocprofv3 kernel stats of the same command,
# FETCH_SIZE / WRITE_SIZE passes (separate, kernel-trace only).   usage: tools/round_profile.sh <tag>
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 bench.py --per-op $OUT/per_op.txt > $OUT/bench.json 2> $OUT/bench.err
tail -c 2500 $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 bench.py --no-cpu-baseline > $OUT/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc/$c -o $c -- python3 bench.py --steps 3 --warmup 2 --eager --no-cpu-baseline --no-roofline > $OUT/pmc_$c.log 2>&1
done
find $OUT -name "*kernel_stats.csv" | head -2
python3 tools/pmc_traffic.py $OUT/pmc $OUT/hbm_traffic.json
# keep the merge-back small: the raw traces are large
find $OUT -name "*kernel_trace.csv" -size +20M -delete
