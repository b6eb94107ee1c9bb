"""Summarise one rocprofv3 --pmc pass of SQ / GRBM counters over `python3 bench.py --no-cpu-baseline --no-sampling --no-graph --steps 3
--warmup 1` into per-kernel averages per launch and two derived ratios:
  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8)   (MI355X_MICROARCH.md: the counter is in
                   cycles, 32 per v_mfma_f32_32x32x16_bf16, summed over the chip's SIMDs; GRBM_GUI_ACTIVE comes back summed over the 8
                   XCDs -- checked on the persistent backward: 14.35 M = 8 x (757 us x 2.37 GHz), and its MFMA cycles are exactly
                   32 x the 3 670 016 MFMAs of a launch)
  parked_frac    = SQ_WAIT_ANY / SQ_WAVE_CYCLES  (waves parked on s_waitcnt / barriers), issue_stall_frac = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
Usage: python profiles/sq_counters.py <dir of the pass> <out.json> [workload label] [sources sha16 (profiles/tools/source_hash.py)]"""
import collections, csv, glob, json, os, re, sys


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
            acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[name].add(r["Dispatch_Id"])
    out = {"workload": sys.argv[3] if len(sys.argv) > 3 else "C2 [256,128,88,5] bf16, eager launches, 3 steps + 1 warm-up",
           "sources_sha16": sys.argv[4] if len(sys.argv) > 4 else None, "kernels": {}}
    for name, c in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0.0)):
        n = len(calls[name])
        if n == 0 or name.startswith("at::") or "rocclr" in name:
            continue
        k = {"calls": n}
        for cn, v in c.items():
            k[cn + "_per_launch"] = v / n
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        if gui > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            k["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * gui / 8.0)
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if wc > 0:
            k["parked_frac"] = c.get("SQ_WAIT_ANY", 0.0) / wc
            k["issue_stall_frac"] = c.get("SQ_WAIT_INST_ANY", 0.0) / wc
            k["active_frac"] = c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
        out["kernels"][name] = k
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    for name, k in list(out["kernels"].items())[:9]:
        print("%-34s calls %3d  mfma_busy %5.3f  parked %5.3f  issue_stall %5.3f  active %5.3f" % (
            name[:34], k["calls"], k.get("mfma_busy_frac", float("nan")), k.get("parked_frac", float("nan")),
            k.get("issue_stall_frac", float("nan")), k.get("active_frac", float("nan"))))


if __name__ == "__main__":
    main()
