"""Summarise one rocprofv3 SQ counter pass into per-kernel matrix-core figures (profiles/<tag>_mfma_util.json).

    python tools/pmc_mfma.py <dir of the --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES ... pass> out.json [kernel_stats.csv]

Per kernel and launch (averages over the sampled launches): SQ_INSTS_MFMA (wave-level MFMA instructions), SQ_VALU_MFMA_BUSY_CYCLES
(cycles a SIMD's matrix pipe is busy; 32 per v_mfma_f32_32x32x16_f16: MI355X_MICROARCH.md), SQ_BUSY_CYCLES, SQ_WAVE_CYCLES (quad-cycles),
SQ_WAIT_ANY. Derived:
  mfma_pipe_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * launch duration * 2.4 GHz)   -- share of the chip's matrix-pipe cycles
                   in use while the kernel runs (duration = rocprofv3 --kernel-trace --stats average of bench.py --inflight 1 --chains 1, i.e. one
                   single-chain forward at a time like the eager counter pass;
                   2.4 GHz is the maximum clock, so this is a lower bound when the chip clocks down under load);
  mfma_cycles_per_inst = SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_MFMA (a consistency check: 32 for 32x32x16 fp16, 64 for 32x32x2 f32).
"""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*$", "", name)
    name = name.replace(", ", ",").replace("(bool)1", "true").replace("(bool)0", "false")
    return name.replace(".kd", "").strip()


def main():
    root, out = sys.argv[1], sys.argv[2]
    dur = {}
    if len(sys.argv) > 3 and os.path.exists(sys.argv[3]):
        with open(sys.argv[3]) as f:
            for r in csv.DictReader(f):
                dur[short(r["Name"])] = float(r["AverageNs"]) * 1e-9
    acc = {}
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                k = short(r["Kernel_Name"])
                if k.startswith("__amd") or "at::" in k or "elementwise" in k:
                    continue
                a = acc.setdefault(k, {})
                c = a.setdefault(r["Counter_Name"], [0.0, 0])
                c[0] += float(r["Counter_Value"])
                c[1] += 1
    res = {}
    for k, a in sorted(acc.items()):
        e = {name: round(v[0] / max(v[1], 1), 1) for name, v in a.items()}
        e["launches_sampled"] = max(v[1] for v in a.values())
        busy, inst = e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), e.get("SQ_INSTS_MFMA", 0.0)
        if inst > 0:
            e["mfma_cycles_per_inst"] = round(busy / inst, 2)
        if k in dur and busy > 0:
            e["avg_launch_us"] = round(dur[k] * 1e6, 2)
            e["mfma_pipe_util"] = round(busy / (1024 * dur[k] * 2.4e9), 4)
        res[k] = e
    with open(out, "w") as f:
        json.dump({"meta": {"source": "rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY "
                                      "--kernel-trace (one pass, no other trace domains) on the eager bench",
                            "formulas": __doc__.split("Derived:")[1].strip()}, "kernels": res}, f, indent=1)
    print(f"{len(res)} kernels -> {out}")


if __name__ == "__main__":
    main()
