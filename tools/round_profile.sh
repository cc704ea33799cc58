#!/bin/bash
# Round artifacts on the GPU box for one bench configuration: the rocprofv3 kernel stats of the bench command (forwards in flight) and of one
# single-chain forward at a time, FETCH_SIZE / WRITE_SIZE passes (separate, kernel-trace only), one SQ pass for the matrix-core counters, and
# LAST the bench line (+ per-op table), whose `roofline.rocprof` / `traffic` fields then read the summaries of THIS build.
#   usage: tools/round_profile.sh <tag> [bench args]        e.g.  tools/round_profile.sh r03            (headline config)
#                                                                  tools/round_profile.sh r03_vgg512 --model ssd512_vgg16 --batch 32
TAG=${1:-r03}; shift
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$TAG/stats -o stats -- python3 bench.py --no-cpu-baseline --no-latency --no-roofline --no-configs "$@" > $OUT/stats.log 2>&1
cp $(find /tmp/rp_$TAG/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
# the same kernels with ONE forward in flight (single chain): what bench.py's event pass measures; with three forwards in flight a launch
# shares the chip and its duration in the summary above stretches (3x for the MFMA-bound VGG convs)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$TAG/stats1 -o stats -- python3 bench.py --no-cpu-baseline --no-latency --no-roofline --no-configs --inflight 1 --chains 1 "$@" > $OUT/stats1.log 2>&1
cp $(find /tmp/rp_$TAG/stats1 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_one_forward.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/rp_$TAG/pmc/$c -o $c -- python3 bench.py --steps 3 --warmup 2 --eager --chains 1 --no-cpu-baseline --no-roofline --no-latency --no-configs "$@" > $OUT/pmc_$c.log 2>&1
done
PMC_CMD="bench.py --steps 3 --warmup 2 --eager --chains 1 $*" python3 tools/pmc_traffic.py /tmp/rp_$TAG/pmc $OUT/hbm_traffic.json
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY --kernel-trace --output-format csv -d /tmp/rp_$TAG/sq -o sq -- python3 bench.py --steps 3 --warmup 2 --eager --chains 1 --no-cpu-baseline --no-roofline --no-latency --no-configs "$@" > $OUT/pmc_sq.log 2>&1
python3 tools/pmc_mfma.py /tmp/rp_$TAG/sq $OUT/mfma_util.json $OUT/kernel_stats_one_forward.csv
rm -rf /tmp/rp_$TAG
# the summaries of this build become the committed ones the bench line quotes
for f in kernel_stats.csv kernel_stats_one_forward.csv hbm_traffic.json mfma_util.json; do cp $OUT/$f profiles/${TAG}_$f; done
python3 bench.py --per-op $OUT/per_op.txt "$@" > $OUT/bench.json 2> $OUT/bench.err
tail -c 1500 $OUT/bench.json; echo
ls $OUT
