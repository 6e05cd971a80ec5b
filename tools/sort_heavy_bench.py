"""developer tool: radix sort with a few dozen heavy top-bit buckets (gathered and sorted by one general call)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unikmer_amd import lib
dev = torch.device("cuda:0")
ctx = lib.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
g = torch.Generator(device=dev); g.manual_seed(5)
for n, ntops, share in ((16_000_000, 40, 0.6), (100_000_000, 20, 0.1), (100_000_000, 30, 0.02), (100_000_000, 3, 0.5)):
    hi = torch.randint(0, 1 << 31, (n,), device=dev, generator=g, dtype=torch.int64)
    lo = torch.randint(0, 1 << 31, (n,), device=dev, generator=g, dtype=torch.int64)
    x = ((hi << 31) ^ lo) & ((1 << 62) - 1)
    m = torch.rand(n, device=dev, generator=g) < share
    tops = torch.randint(0, 1 << 16, (ntops,), device=dev, generator=g, dtype=torch.int64)
    pick = tops[torch.randint(0, ntops, (n,), device=dev, generator=g)]
    x = torch.where(m, (x & ((1 << 46) - 1)) | (pick << 46), x)
    exp = torch.sort(x).values
    res = {}
    for knob in ("1", "0"):
        ctx.set_option("sort_local", int(knob))   # (contexts read the environment once, at creation)
        best = 1e9
        for _ in range(3):
            w = x.clone(); torch.cuda.synchronize(); t0 = time.perf_counter(); ctx.sort_u64(w, 62); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) * 1e3)
        assert torch.equal(w, exp)
        res[knob] = round(best, 3)
    print("n=%d, %d heavy buckets holding %.0f %% of the keys: bucket route %.3f ms, all passes %.3f ms" % (n, ntops, share * 100, res["1"], res["0"]))
