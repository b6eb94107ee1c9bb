"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `python3 bench.py --no-cpu-baseline --no-graph --steps 3 --warmup 1`
into per-kernel HBM-side bytes per launch (MI355X_MICROARCH.md, HBM section: bytes = counter * 1024; FETCH_SIZE doubled on gfx950).
Usage: python profiles/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> [workload label] [steps recorded]
[sources sha16 (profiles/tools/source_hash.py)]"""
import collections
import csv
import glob
import json
import os
import re
import sys


def load(d, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
                acc[name][0] += float(r["Counter_Value"])
                acc[name][1] += 1
    return acc


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {"workload": sys.argv[4] if len(sys.argv) > 4 else "C2 [256,128,88,5] bf16, eager launches, 3 steps + 1 warm-up",
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes (MI355X_MICROARCH.md HBM section): "
                     "bytes = counter * 1024; FETCH_SIZE doubled (gfx950 tallies 128-B read requests at 64 B)",
           "steps_recorded": int(sys.argv[5]) if len(sys.argv) > 5 else 4,
           "sources_sha16": sys.argv[6] if len(sys.argv) > 6 else None,
           "kernels": {}}
    for name in sorted(fetch, key=lambda k: -(fetch[k][0] + write.get(k, [0, 0])[0])):
        calls = fetch[name][1]
        if calls == 0 or name.startswith("at::") or "rocclr" in name:
            continue
        fb = fetch[name][0] * 1024.0
        wb = write.get(name, [0.0, 0])[0] * 1024.0
        out["kernels"][name] = {"calls": calls, "fetch_size_bytes_raw": fb / calls, "fetch_bytes_x2_corrected": 2 * fb / calls,
                                "write_bytes": wb / calls, "hbm_side_bytes_per_launch": (2 * fb + wb) / calls}
    # optimiser steps the passes really contain (bench.py runs a few eager steps for its per-call timing besides --steps / --warmup): one
    # clip_adam_kernel launch per step
    for name, k in out["kernels"].items():
        if name.startswith("clip_adam_kernel"):
            out["steps_recorded"] = k["calls"]
    out["hbm_side_bytes_per_step"] = sum(k["hbm_side_bytes_per_launch"] * k["calls"] for k in out["kernels"].values()) / out["steps_recorded"]
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print("HBM-side bytes per step (all kernels of the library): %.2f GB" % (out["hbm_side_bytes_per_step"] / 1e9))
    for k, v in list(out["kernels"].items())[:8]:
        print("%-40s %8.1f MB / launch" % (k[:40], v["hbm_side_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
