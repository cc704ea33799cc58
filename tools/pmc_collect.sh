#!/bin/bash
# Collect PMC counters for the headline bench in separate passes (MI355X_MICROARCH.md: SQ 8 slots, TCC 4; FETCH_SIZE and
# WRITE_SIZE cannot share a pass). Kernel trace + counters only (no sys/hip trace domains).
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
mkdir -p $OUT
ARGS="bench.py --steps 3 --warmup 2 --eager --no-cpu-baseline --no-roofline"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT TCC_MISS TCP_TOTAL_CACHE_ACCESSES" "GRBM_GUI_ACTIVE GRBM_TA_BUSY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- python3 $ARGS > $OUT/p$i.log 2>&1
  ls $OUT/p$i | head -3
done
