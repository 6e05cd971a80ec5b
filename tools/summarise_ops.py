#!/usr/bin/env python
"""Summarise gpurun_out/ops_r02 (tools/collect_ops_r02.sh): per kernel of this library, mean duration from the
--kernel-trace --stats pass and mean counter values per dispatch from the separate --pmc passes.
FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE x2 is the gfx950 correction for wide (16 B / lane) streaming reads
(MI355X_MICROARCH.md, HBM) and is reported next to the raw value, because the narrower loads of some of these
kernels are not covered by that calibration."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

KEEP = ("onesweep_kernel", "radix_hist_kernel", "unique_tile_kernel", "window_kernel", "stripwin_kernel", "nthash_strip_kernel", "kway_kernel",
        "kw_compact_kernel", "setop_tile_kernel")


def short(name):
    for k in KEEP:
        if k in name:
            i = name.index(k)
            return name[i:].split("(")[0][:90]
    return None


def main():
    root = sys.argv[1]
    out = defaultdict(dict)
    for d in sorted(glob.glob(os.path.join(root, "*"))):
        if not os.path.isdir(d):
            continue
        tag = os.path.basename(d)
        for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                s = short(row["Name"])
                if s:
                    out[s]["mean_ms"] = float(row["AverageNs"]) / 1e6
                    out[s]["dispatches"] = int(row["Calls"])
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            agg = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
            for row in csv.DictReader(open(f)):
                s = short(row["Kernel_Name"])
                if s:
                    agg[s][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
            for s, cs in agg.items():
                for cn, per in cs.items():
                    out[s][cn] = sum(per.values()) / len(per)
    for s, v in out.items():
        if "FETCH_SIZE" in v:
            v["FETCH_GB_raw"] = v["FETCH_SIZE"] * 1024 / 1e9
            v["FETCH_GB"] = 2 * v["FETCH_GB_raw"]
        if "WRITE_SIZE" in v:
            v["WRITE_GB"] = v["WRITE_SIZE"] * 1024 / 1e9
        if "SQ_WAVE_CYCLES" in v and v["SQ_WAVE_CYCLES"]:
            v["valu_active_per_wave_cycle"] = v.get("SQ_ACTIVE_INST_VALU", 0) / v["SQ_WAVE_CYCLES"]
            v["wait_any"] = v.get("SQ_WAIT_ANY", 0) / v["SQ_WAVE_CYCLES"]
            if v.get("SQ_BUSY_CYCLES"):
                # SQ_BUSY_CYCLES is summed over the SEs; VALU busy ~ active VALU quad-cycles per SIMD per busy cycle
                v["valu_busy"] = v.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (v["SQ_BUSY_CYCLES"] / 32 * 1024) if v["SQ_BUSY_CYCLES"] else None
        if v.get("SQ_LDS_IDX_ACTIVE"):
            v["lds_conflict"] = v.get("SQ_LDS_BANK_CONFLICT", 0) / v["SQ_LDS_IDX_ACTIVE"]
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
