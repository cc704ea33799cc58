"""Summarise rocprofv3 --pmc passes into per-kernel HBM traffic per launch (profiles/<tag>_hbm_traffic.json).

    python tools/pmc_traffic.py gpurun_out/pmc profiles/r01_hbm_traffic.json [--skip-launches N]

Inputs: the FETCH_SIZE and WRITE_SIZE passes written by tools/pmc_collect.sh (separate passes: the two counters do not
fit the TCC's 4 slots together). Corrections per /opt/skills/guides/MI355X_MICROARCH.md "HBM": both counters are in KB;
on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, i.e. reports half of a wide coalesced read, so
    bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
Kernel names are reduced to the same spelling bench.py prints ("pw_kernel<128,64,4,1,false,32>").
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
    name = re.sub(r"\(.*$", "", name)                  # argument list
    name = name.replace(", ", ",").replace("(bool)1", "true").replace("(bool)0", "false")
    return name.replace(".kd", "").strip()


def collect(root, counter):
    rows = {}
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] != counter:
                    continue
                k = short(r["Kernel_Name"])
                a = rows.setdefault(k, [0.0, 0])
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    return rows


def main():
    root, out = sys.argv[1], sys.argv[2]
    fetch, write = collect(root, "FETCH_SIZE"), collect(root, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fetch) | set(write)):
        if k.startswith("__amd") or "at::" in k or "elementwise" in k:
            continue
        f, nf = fetch.get(k, [0.0, 0])
        w, nw = write.get(k, [0.0, 0])
        fpl = f / nf if nf else 0.0
        wpl = w / nw if nw else 0.0
        res[k] = {"launches_sampled": max(nf, nw), "fetch_kb_raw_per_launch": round(fpl, 1), "write_kb_per_launch": round(wpl, 1),
                  "hbm_bytes_per_launch": round((2.0 * fpl + wpl) * 1024.0)}
    meta = {"formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch, averaged over the sampled launches",
            "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on: " + os.environ.get("PMC_CMD", "bench.py --steps 3 --warmup 2 --eager")}
    with open(out, "w") as f:
        json.dump({"meta": meta, "kernels": res}, f, indent=1)
    print(f"{len(res)} kernels -> {out}")


if __name__ == "__main__":
    main()
