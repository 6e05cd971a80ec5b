#!/usr/bin/env python
"""developer tool: the 2-way kernel with per-record taxids against the set size (round-5 review: 12 x the time for 10 x the
data).  usage: [KIND=random|file|runs] [DIFF=1] python tools/tax_scaling.py [sizes...]   (with a -DUKM_PROFILE_PHASES build the library prints cycles per tile
and phase to stderr)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
from unikmer_amd import lib
dev = torch.device("cuda", 0)
ctx = lib.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
T = sum(8 ** d for d in range(8))
child = np.arange(1, T + 1, dtype=np.uint32)
parent = ((child.astype(np.int64) - 2) // 8 + 1).astype(np.uint32); parent[0] = 1
ctx.taxonomy_load(child, parent)
KIND = os.environ.get("KIND", "random")
sizes = [int(float(x)) for x in sys.argv[1:]] or [100_000_000, 300_000_000, 1_000_000_000]
for n in sizes:
    A, B = bench.gen_sets_device((4 * n + 2) // 3, 32, 0, bench.SEED, dev)
    na, nb = A.numel(), B.numel()
    if KIND == "file":      # one taxid per file, handed over as arrays
        ta = torch.full((na,), 123457, dtype=torch.int32, device=dev)
        tb = torch.full((nb,), 2345678, dtype=torch.int32, device=dev)
    elif KIND == "runs":    # runs of 4096 records with one taxid (clustered taxa)
        ta = (1 + (bench.splitmix64_torch((torch.arange(na, device=dev) >> 12) ^ bench._i64(bench.SEED + 2)) & ((1 << 40) - 1)) % T).to(torch.int32)
        tb = (1 + (bench.splitmix64_torch((torch.arange(nb, device=dev) >> 12) ^ bench._i64(bench.SEED + 3)) & ((1 << 40) - 1)) % T).to(torch.int32)
    else:                   # uniformly random per record (SURVEY 8(d)'s generator)
        ta = (1 + (bench.splitmix64_torch(A ^ bench._i64(bench.SEED + 2)) & ((1 << 40) - 1)) % T).to(torch.int32)
        tb = (1 + (bench.splitmix64_torch(B ^ bench._i64(bench.SEED + 3)) & ((1 << 40) - 1)) % T).to(torch.int32)
    out = torch.empty(na + nb, dtype=torch.int64, device=dev); tout = torch.empty(na + nb, dtype=torch.int32, device=dev)
    res = {}
    ops = [("union_tax", lib.OP_UNION, True, 0), ("inter_tax", lib.OP_INTER, True, 0), ("union", lib.OP_UNION, False, 0), ("inter", lib.OP_INTER, False, 0)]
    if os.environ.get("DIFF"):
        ops += [("diff_tax", lib.OP_DIFF, True, 0), ("diff_t_tax", lib.OP_DIFF, True, lib.F_CMP_TAXID)]
    for name, op, tx, fl in ops:
        best = 1e9
        for _ in range(4):
            r = ctx.setop2(op, A, B, ta, tb, flags=fl, out=out, out_taxids=tout) if tx else ctx.setop2(op, A, B, out=out)
            best = min(best, ctx.last_kernel_ms())
        res[name] = round(best, 3)
    print("n=%d" % n, res, "ns per input record:", {k: round(v * 1e6 / (na + nb), 4) for k, v in res.items()}, flush=True)
    del A, B, ta, tb, out, tout
    torch.cuda.empty_cache()
