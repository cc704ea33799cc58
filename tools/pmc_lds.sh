#!/bin/bash
# dev: LDS bank-conflict share per kernel of one eager forward (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE).   usage on the GPU box: tools/pmc_lds.sh <out.txt> [bench args]
OUT=$1; shift
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/rp_lds
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS --kernel-trace --output-format csv -d /tmp/rp_lds -o lds -- python3 bench.py --steps 2 --warmup 1 --eager --chains 1 --no-cpu-baseline --no-roofline --no-latency --no-configs "$@" > $OUT.log 2>&1
f=$(find /tmp/rp_lds -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY' > $OUT
import csv, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    k = re.sub(r"^void ", "", k).split("(")[0][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0))[:24]:
    act = v.get("SQ_LDS_IDX_ACTIVE", 0); bc = v.get("SQ_LDS_BANK_CONFLICT", 0)
    print("%-70s LDS cycles %.3e of which bank conflicts %.3e (%.1f %%), LDS instructions %.3e" % (k, act, bc, 100 * bc / max(act, 1), v.get("SQ_INSTS_LDS", 0)))
PY
cat $OUT
