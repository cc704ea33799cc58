cd $GRAFT_REPO_ROOT
bash tools/round_profile.sh r02 > /dev/null 2>&1
bash tools/valu.sh r02 valu --batch 64 > /dev/null 2>&1
mkdir -p profiles_tmp
for f in kernel_stats.csv kernel_stats_one_forward.csv hbm_traffic.json mfma_util.json per_op.txt; do cp gpurun_out/r02/$f profiles/r02_$f; done
bash tools/_lines.sh
