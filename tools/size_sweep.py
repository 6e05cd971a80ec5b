#!/usr/bin/env python
"""Set-op kernel rate against input size (device-resident, bench.py's generator)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from unikmer_amd import lib

dev = torch.device("cuda", 0)
ctx = lib.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
res = {}
for n in (1e5, 1e6, 1e7, 1e8, 3e8, 1e9, 2e9):
    n = int(n)
    A, B = bench.gen_sets_device((4 * n + 2) // 3, 30 if n <= 1e9 else 29, 0, bench.SEED, dev)
    out = torch.empty(A.numel() + B.numel(), dtype=torch.int64, device=dev)
    row = {}
    for name, op in (("union", lib.OP_UNION), ("inter", lib.OP_INTER)):
        ks, cs = [], []
        for _ in range(5):
            r = ctx.setop2(op, A, B, out=out)
            ks.append(ctx.last_kernel_ms()); cs.append(ctx.last_call_ms())
        byt = 8 * (A.numel() + B.numel()) + 8 * r.numel()
        row[name] = {"kernel_ms": round(min(ks), 4), "call_ms": round(min(cs), 4), "TBps_kernel": round(byt / min(ks) / 1e9, 3),
                     "kmers_per_s_call": round((A.numel() + B.numel()) / min(cs) * 1e3, 0)}
    res["%.0e" % n] = row
    del A, B, out
print(json.dumps(res, indent=1))
