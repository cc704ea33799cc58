#!/bin/bash
# copies the summaries of tools/round_profile.sh runs (merged back under gpurun_out/<tag>/) into the tracked profiles/ directory
cd "$(dirname "$0")/.."
for tag in "$@"; do
  for f in kernel_stats.csv kernel_stats_one_forward.csv hbm_traffic.json mfma_util.json per_op.txt bench.json; do
    [ -f gpurun_out/$tag/$f ] && cp gpurun_out/$tag/$f profiles/${tag}_$f
  done
done
ls profiles | grep -c "^r06"
