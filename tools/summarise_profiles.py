#!/usr/bin/env python
"""Turn gpurun_out/prof_<tag>/ (written by tools/collect_profiles.sh on the GPU box) into the
committed summaries under profiles/: kernel stats, PMC summary, bench lines, traffic.json."""
import csv
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pmc(dirpath, filt):
    f = os.path.join(dirpath, "bench_counter_collection.csv")
    agg = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
    dur = defaultdict(dict)
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if filt not in name:
            continue
        short = name[name.index(filt):].split("(")[0]
        agg[short][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
        dur[short][row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
    out = {}
    for k in agg:
        out[k] = {"dispatches": len(dur[k]), "mean_ms": sum(dur[k].values()) / len(dur[k])}
        for cn, per in agg[k].items():
            out[k][cn] = sum(per.values()) / len(per)
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    # 1. kernel stats (rocprofv3 --kernel-trace --stats): keep our kernels + the top of the rest
    rows = list(csv.DictReader(open(os.path.join(src, "trace", "bench_kernel_stats.csv"))))
    keep = [r for r in rows if "anonymous namespace" in r["Name"]] + [r for r in rows if "anonymous namespace" not in r["Name"]][:8]
    with open(os.path.join(dst, tag + "_bench_kernel_stats.csv"), "w") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in keep:
            r = dict(r)
            r["Name"] = r["Name"][:160]
            w.writerow(r)
    # 2. PMC summaries (separate passes)
    summary = {}
    for d in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2"):
        p = os.path.join(src, d)
        if os.path.isdir(p):
            summary[d] = pmc(p, "setop_tile_kernel")
    json.dump(summary, open(os.path.join(dst, tag + "_bench_pmc.json"), "w"), indent=1)
    # 3. bench lines of the profiled runs (first = the --kernel-trace --stats run)
    lines = [json.loads(l) for l in open(os.path.join(src, "bench_lines.jsonl")) if l.strip()]
    json.dump(lines, open(os.path.join(dst, tag + "_bench_lines_under_rocprof.json"), "w"), indent=1)
    # 4. HBM traffic per launch of the dominant kernel, with the gfx950 correction of
    #    MI355X_MICROARCH.md §HBM: FETCH_SIZE counts 128-B requests of a wide coalesced stream
    #    at 64 B -> x2; WRITE_SIZE as reported.  Both counters are in KiB.
    def find(d, op, cn):
        for k, v in summary.get(d, {}).items():
            if k.startswith("setop_tile_kernel<%d," % op):
                return v.get(cn)
        return None
    import subprocess
    try:
        head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], text=True).strip()
    except Exception:
        head = None
    tr = {"n": int(lines[0]["config"]["per_gpu_set_size"]), "tag": tag, "head": head,
          "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE, one pass each, over `python bench.py --steps 5 "
                    "--warmup 2 --cpu-sample 0` (tools/collect_profiles.sh)",
          "note": "FETCH_SIZE x2 (gfx950 wide-stream correction), WRITE_SIZE as reported; KiB -> bytes; "
                  "separate --pmc passes; mean per launch over the run's dispatches"}
    for op, nm in ((0, "union"), (1, "inter")):
        f, w = find("pmc_fetch", op, "FETCH_SIZE"), find("pmc_write", op, "WRITE_SIZE")
        if f is not None and w is not None:
            tr[nm + "_fetch_bytes_per_launch"] = 2 * f * 1024
            tr[nm + "_write_bytes_per_launch"] = w * 1024
            tr[nm + "_traffic_bytes_per_launch"] = 2 * f * 1024 + w * 1024
    json.dump(tr, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    print(json.dumps(tr, indent=1))
    for r in keep[:4]:
        print(r["Name"][:80], r["Calls"], r["AverageNs"])
    print("bench (trace run): union kernel_ms", lines[0]["roofline"]["kernel_ms"], "inter", lines[0]["roofline_inter"]["kernel_ms"])


if __name__ == "__main__":
    main()
