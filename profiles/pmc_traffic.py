"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `python3 bench.py --no-cpu-baseline --no-graph --steps 3 --warmup 1`
into per-kernel HBM-side bytes per launch (MI355X_MICROARCH.md, HBM section: bytes = counter * 1024; FETCH_SIZE doubled on gfx950).
Usage: python profiles/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> [workload label]"""
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
           "kernels": {}}
    for name in sorted(fetch, key=lambda k: -(fetch[k][0] + write.get(k, [0, 0])[0])):
        calls = fetch[name][1]
        if calls == 0 or name.startswith("at::") or "rocclr" in name:
            continue
        fb = fetch[name][0] * 1024.0
        wb = write.get(name, [0.0, 0])[0] * 1024.0
        out["kernels"][name] = {"calls": calls, "fetch_size_bytes_raw": fb / calls, "fetch_bytes_x2_corrected": 2 * fb / calls,
                                "write_bytes": wb / calls, "hbm_side_bytes_per_launch": (2 * fb + wb) / calls}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in list(out["kernels"].items())[:8]:
        print("%-40s %8.1f MB / launch" % (k[:40], v["hbm_side_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
