"""developer tool: radix sort of 1e8 keys on a fresh context, after a crowded input (sample guard active), and on the crowded input itself"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unikmer_amd import lib
dev = torch.device("cuda:0")
ctx = lib.Context(0)
n = 100_000_000
g = torch.Generator(device=dev); g.manual_seed(1)
hi = torch.randint(0, 1 << 31, (n,), device=dev, generator=g, dtype=torch.int64)
lo = torch.randint(0, 1 << 31, (n,), device=dev, generator=g, dtype=torch.int64)
uni = ((hi << 31) ^ lo) & ((1 << 62) - 1)
crowd = (uni & ((1 << 52) - 1)) | (3 << 52)   # 64 top-16-bit buckets only
def t(x, label):
    best = 1e9
    for _ in range(4):
        w = x.clone(); torch.cuda.synchronize(); t0 = time.perf_counter(); ctx.sort_u64(w, 62); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) * 1e3)
    print(label, round(best, 3), "ms")
t(uni, "even keys, fresh context")
t(crowd, "crowded keys (first call pays the wasted attempt; the minimum over 4 is with the guard)")
t(uni, "even keys after the context has met crowded keys (sample first)")
