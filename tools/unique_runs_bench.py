"""developer tool: `unique` (-u with taxids, LCA fold per run) call time against the run length.
usage: python tools/unique_runs_bench.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, bench
from unikmer_amd import lib
from conftest import synth_tree
dev = torch.device("cuda:0")
ctx = lib.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
child, parent = synth_tree(7, 8)
ctx.taxonomy_load(child, parent)
T = len(child)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
i = torch.arange(n, dtype=torch.int64, device=dev)
tax = (1 + (bench.splitmix64_torch(i ^ bench._i64(12345)) & ((1 << 40) - 1)) % T).to(torch.int32)
out = torch.empty(n, dtype=torch.int64, device=dev)
outt = torch.empty(n, dtype=torch.int32, device=dev)
for L in (1, 2, 4, 8, 9, 12, 16, 24, 32, 48, 64, 65, 100, 200, 900, 5000):
    keys = (i // L) * 977 + 5
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = ctx.unique(keys, tax, lib.UNIQUE, out=out, out_taxids=outt)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("run length %5d: %8.2f ms  (%d out)" % (L, min(ts), r[0].numel()))
