#!/usr/bin/env python
"""developer tool: the rate of random reads of a small table (what a per-record taxid costs: clade8[taxid] is one byte of a
2.4 MB table, euler[taxid] four bytes of a 9.6 MB one) measured WITHOUT this library -- torch's own gather kernel
(table[idx], int32 indices read with coalesced loads, one value written per index).  The kernels with per-record taxids are
priced against this rate in DESIGN.md (round 6).  usage: python tools/gather_ceiling.py [N=5e8]"""
import sys, torch
dev = torch.device("cuda", 0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 500_000_000
T = sum(8 ** d for d in range(8))
g = torch.Generator(device=dev); g.manual_seed(1)
for name, dtype, entries in (("u8 x 2.4e6 (clade8)", torch.uint8, T + 1), ("u32 x 2.4e6 (euler)", torch.int32, T + 1),
                             ("u8 x 16384 (fits L1)", torch.uint8, 16384), ("u8 x 3.4e7 (34 MB: beyond L2)", torch.uint8, 34_000_000)):
    table = torch.randint(0, 200, (entries,), device=dev, generator=g).to(dtype)
    idx = torch.randint(0, entries, (n,), dtype=torch.int32, device=dev, generator=g)
    seq = torch.arange(n, dtype=torch.int32, device=dev) % entries
    out = torch.empty(n, dtype=dtype, device=dev)
    res = {}
    for kind, ix in (("random", idx), ("sequential", seq)):
        best = 1e9
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); torch.index_select(table, 0, ix, out=out); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        res[kind] = "%.2f ms = %.0f G reads/s" % (best, n / best / 1e6)
    print(name, res, flush=True)
    del table, idx, seq, out
