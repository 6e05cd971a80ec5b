"""n-way `common` / `merge` / `union` call times over config-4-shaped files (p = 0.9 draws over one universe).
usage: python tools/common_bench.py NFILES PER_FILE [tax]"""
import os, sys, time, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import bench
from unikmer_amd import lib
import numpy as np
dev = torch.device("cuda", 0)
ctx = lib.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
nfiles, per = int(sys.argv[1]), int(float(sys.argv[2]))
tax = len(sys.argv) > 3 and sys.argv[3] == "tax"
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import synth_tree
child, parent = synth_tree(7, 8); ctx.taxonomy_load(child, parent); T = len(child)
P = float(os.environ.get("CB_P", "0.9"))   # share of the universe a file holds (overlap between files)
nu = int(per / P)
j = torch.arange(nu, dtype=torch.int64, device=dev)
gaps = 1 + (bench.splitmix64_torch(j ^ bench._i64(bench.SEED)) & ((1 << 32) - 1))
U = torch.cumsum(gaps, 0)
thr = int(P * (1 << 20))
files, taxs = [], []
for f in range(nfiles):
    h = bench.splitmix64_torch(j ^ bench._i64(bench.SEED + 1000 * (f + 1)))
    k = U[((h >> 11) & ((1 << 20) - 1)) < thr]
    files.append(k)
    taxs.append((1 + (bench.splitmix64_torch(k ^ bench._i64(bench.SEED + 2 + f)) & ((1 << 40) - 1)) % T).to(torch.int32))
total = sum(x.numel() for x in files)
ok = torch.empty(total + 8, dtype=torch.int64, device=dev)
ot = torch.empty(total + 8, dtype=torch.int32, device=dev)
def wall(fn, reps=3):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return ["%.1f" % t for t in ts], out
tx = taxs if tax else None
kw = dict(out=ok, out_taxids=ot) if tax else dict(out=ok)
print("files", nfiles, "total", total, "tax", tax)
print("common thr=S-1", wall(lambda: ctx.common(files, nfiles - 1, tx, **kw))[0])
print("common thr=S  ", wall(lambda: ctx.common(files, nfiles, tx, **kw))[0])
print("merge_k       ", wall(lambda: ctx.merge_k(files, tx, **kw))[0])
print("union         ", wall(lambda: ctx.union(files, tx, **kw))[0])
