import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bench
from unikmer_amd import lib
dev = torch.device("cuda", 0)
ctx = lib.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
A, B = bench.gen_sets_device((4 * n + 2) // 3, 30, 0, bench.SEED, dev)
cat = torch.cat([A, B])
print("cat", cat.numel(), flush=True)
ref = torch.sort(cat)[0]
torch.cuda.synchronize()
print("torch sort done", flush=True)
ctx.sort_u64(cat, 62)
torch.cuda.synchronize()
print("ukm sort done", bool((cat == ref).all()), flush=True)
out = torch.empty(cat.numel(), dtype=torch.int64, device=dev)
u = ctx.unique(cat, out=out)
torch.cuda.synchronize()
print("unique done", u.numel(), torch.unique_consecutive(ref).numel(), flush=True)
