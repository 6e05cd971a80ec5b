#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSV output: per kernel (name substring filter), mean counter
values per dispatch and mean duration.  Usage: pmc_summary.py DIR [DIR...] [--filter setop_tile]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
    filt = "setop_tile"
    for a in sys.argv[1:]:
        if a.startswith("--filter="):
            filt = a.split("=", 1)[1]
    for d in dirs:
        for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
            agg = defaultdict(lambda: defaultdict(list))
            dur = defaultdict(dict)
            for row in csv.DictReader(open(f)):
                name = row["Kernel_Name"]
                if filt not in name:
                    continue
                short = name.split("(")[1] if name.startswith("void (") else name
                short = name[name.index(filt):][:60]
                agg[short][row["Counter_Name"]].append((row["Dispatch_Id"], float(row["Counter_Value"])))
                dur[short][row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
                meta = (row["VGPR_Count"], row["SGPR_Count"], row["LDS_Block_Size"], row["Grid_Size"], row["Workgroup_Size"])
            for k in agg:
                ds = dur[k]
                print("== %s  [%s]  dispatches=%d mean_ms=%.3f" % (k, os.path.basename(d), len(ds), sum(ds.values()) / len(ds)))
                for cn, vals in sorted(agg[k].items()):
                    per = defaultdict(float)
                    for did, v in vals:
                        per[did] += v
                    print("   %-28s %.6g" % (cn, sum(per.values()) / len(per)))
                print("   meta(vgpr,sgpr,lds,grid,wg) =", meta)


if __name__ == "__main__":
    main()
