"""developer tool: config-4 diff call time against its kernel time (where the rest of the call goes)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, bench
from unikmer_amd import lib
from conftest import synth_tree
dev = torch.device("cuda:0")
ctx = lib.Context(0)
child, parent = synth_tree(7, 8)
ctx.taxonomy_load(child, parent)
nfiles, per = 1000, 1_000_000
nu = int(per / 0.9)
j = torch.arange(nu, dtype=torch.int64, device=dev)
U = torch.cumsum(1 + (bench.splitmix64_torch(j ^ bench._i64(bench.SEED)) & ((1 << 24) - 1)), 0)
files, taxs = [], []
for f in range(nfiles):
    h = bench.splitmix64_torch(j ^ bench._i64(bench.SEED + 1000 * (f + 1)))
    m = ((h >> 11).double() / float(1 << 53)) < 0.9
    k = U[m]
    files.append(k)
    taxs.append((1 + (bench.splitmix64_torch(k ^ bench._i64(7 + f)) % len(child)).abs()).to(torch.int32))
out = torch.empty(nu, dtype=torch.int64, device=dev); outt = torch.empty(nu, dtype=torch.int32, device=dev)
for name, fn in (("diff", lambda: ctx.diff(files, taxs, out=out, out_taxids=outt)), ("diff_plain", lambda: ctx.diff(files, out=out))):
    fn(); torch.cuda.synchronize()
    ts, ks = [], []
    for _ in range(5):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3); ks.append(ctx.last_kernel_ms())
    print(name, "call ms", round(min(ts), 3), "kernel ms", round(min(ks), 3), "lib call ms", round(ctx.last_call_ms(), 3))
