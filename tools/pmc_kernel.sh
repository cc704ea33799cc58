#!/bin/bash
# dev tool: hardware counters of the kernels whose name matches a regex, one rocprofv3 --pmc pass per counter group (kernel trace only,
# program directly after `--`), averaged per launch -> gpurun_out/<tag>/pmc_<name>.txt
#   usage: tools/pmc_kernel.sh <tag> <name> <kernel regex> [env K=V ...] -- [bench args]
# Groups: SQ issue / wait split, SQ pipes, LDS, L1 (TCP) and L2 (TCC) requests. Counter names that this rocprofv3 does not know are
# reported by the pass that asked for them (log_pmc_<name>_<i>.txt) and simply missing from the table. TA_* counters are left out: they abort rocprofv3 here.
TAG=$1; NAME=$2; REGEX=$3; shift 3
ENVS=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do ENVS+=("$1"); shift; done
shift
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
for kv in "${ENVS[@]}"; do export "$kv"; done
GROUPS_=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SMEM SQ_WAVE32_INSTS"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
 "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITE_sum"
 "GRBM_GUI_ACTIVE GRBM_COUNT"
 "TD_TD_BUSY_sum TD_TC_STALL_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
 "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"
 "TCC_BUSY_avr TCC_TAG_STALL_sum TCC_CYCLE_sum TCC_EA0_RDREQ_32B_sum"
)
i=0
for g in "${GROUPS_[@]}"; do
  # (every pass under its own timeout: a TA_* group aborted rocprofv3 on this image and its finalization then hung for the whole call)
  timeout 180 rocprofv3 --pmc $g --kernel-trace --output-format csv -d /tmp/pk_$NAME/$i -o p -- python3 bench.py --steps 3 --warmup 2 --eager --chains 1 --no-cpu-baseline --no-roofline --no-latency --no-configs "$@" > $OUT/log_pmc_${NAME}_$i.txt 2>&1
  i=$((i+1))
done
python3 - "$REGEX" /tmp/pk_$NAME $OUT/pmc_$NAME.txt <<'PY'
import csv, glob, os, re, sys
rx, root, out = re.compile(sys.argv[1]), sys.argv[2], sys.argv[3]
acc = {}
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); k = re.sub(r"^void ", "", k); k = re.sub(r"\(.*$", "", k)
            if not rx.search(k): continue
            c = acc.setdefault(k, {}).setdefault(r["Counter_Name"], [0.0, 0])
            c[0] += float(r["Counter_Value"]); c[1] += 1
with open(out, "w") as f:
    for k, a in sorted(acc.items()):
        f.write(k + "\n")
        for name, v in a.items():
            f.write("    %-34s %16.1f   (%d launches)\n" % (name, v[0] / max(v[1], 1), v[1]))
print(open(out).read())
PY
rm -rf /tmp/pk_$NAME
