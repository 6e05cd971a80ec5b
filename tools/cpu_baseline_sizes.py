#!/usr/bin/env python
""""time the CPU at 1e8 and 1e9/8 and state the size used" -- bench.py's cpu_baseline leg (the oracle's C
restatement of the reference's hash-map union + 2-pointer inter, one thread, and the all-cores merge beside it) at those two
sizes, once, on the GPU box's host cores.  The driver-run bench keeps its 2 x 2e7 sample (run time); its `sample` string cites
the file this writes.  usage: python tools/cpu_baseline_sizes.py [OUT.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

out = {}
for n in (2e7, 1e8, 1.25e8):
    r = bench.cpu_baseline((4 * int(n) + 2) // 3, 32)
    out["2x%g" % n] = r
    print("2 x %g: 1 thread %.3g k-mers/s (union %.1f s, inter %.1f s); all cores (%d) %.3g k-mers/s" % (
        n, r["value"], r["union_s"], r["inter_s"], r["allcores_sorted_merge"]["cores"], r["allcores_sorted_merge"]["value"]), flush=True)
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r06", "cpu_baseline_sizes.json")
os.makedirs(os.path.dirname(path), exist_ok=True)
json.dump(out, open(path, "w"), indent=1)
