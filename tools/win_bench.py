"""Kernel-only times of the window kernels (encode k=31, ntHash k=51) on n bases in `nrec` records.
Usage: python tools/win_bench.py [--n 1e8] [--nrec 100]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from unikmer_amd import lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=float, default=1e8)
ap.add_argument("--nrec", type=int, default=100)
ap.add_argument("--k", type=int, default=31)
ap.add_argument("--kh", type=int, default=51)
a = ap.parse_args()
n = int(a.n)
dev = torch.device("cuda:0")
ctx = L.Context(0)
i = torch.arange(n, dtype=torch.int64, device=dev)
w = bench.splitmix64_torch((i >> 5) ^ bench._i64(bench.SEED))
bases = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)[(w >> (2 * (i & 31))) & 3]
del i, w
off = torch.tensor([n * r // a.nrec for r in range(a.nrec + 1)], dtype=torch.int64, device=dev)
out = torch.empty(n, dtype=torch.int64, device=dev)
res = {}
for name, fn in (("encode_k31", lambda: ctx.encode_kmers(bases, off, a.k, out=out)),
                 ("nthash_k51", lambda: ctx.nthash(bases, off, a.kh, out=out))):
    ks, cs = [], []
    for _ in range(7):
        fn()
        ks.append(ctx.last_kernel_ms())
        cs.append(ctx.last_call_ms())
    res[name] = {"kernel_ms": min(ks), "call_ms": min(cs), "GBps": 9 * n / min(ks) / 1e6}
print(json.dumps(res))
