import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bench
from unikmer_amd import lib
dev = torch.device("cuda", 0)
ctx = lib.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
mode = sys.argv[2] if len(sys.argv) > 2 else "all"
A, B = bench.gen_sets_device((4 * n + 2) // 3, 30, 0, bench.SEED, dev)
cat = torch.cat([A, B])
print("cat", cat.numel(), flush=True)
ref = torch.sort(cat)[0]
nu = torch.unique_consecutive(ref).numel()
torch.cuda.synchronize()
if mode in ("all", "sort"):
    for it in range(4):
        w = cat.clone()
        torch.cuda.synchronize()
        ctx.sort_u64(w, 62)
        torch.cuda.synchronize()
        print("sort", it, bool((w == ref).all()), flush=True)
if mode in ("all", "unique"):
    out = torch.empty(cat.numel(), dtype=torch.int64, device=dev)
    for it in range(8):
        u = ctx.unique(ref, out=out)
        torch.cuda.synchronize()
        print("unique", it, u.numel() == nu, ctx.last_call_ms(), flush=True)
if mode in ("all", "mixed"):
    out = torch.empty(cat.numel(), dtype=torch.int64, device=dev)
    w = cat.clone()
    ctx.sort_u64(w, 62)
    for it in range(6):
        u = ctx.unique(w, out=out)
        print("mixed unique", it, u.numel() == nu, flush=True)
print("done")
