"""developer tool: config-4 (core shape) inter / diff call time against kernel time and library call time: where the rest goes.
usage: python tools/c4_host_overhead.py      (C4_TAX=file: every record of a file carries that file's taxid instead of a random one)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench
from unikmer_amd import lib
from conftest import synth_tree
dev = torch.device("cuda:0")
ctx = lib.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
child, parent = synth_tree(7, 8)
ctx.taxonomy_load(child, parent)
T = len(child)
nfiles, per = 1000, 1_000_000
nu = int(per / 0.9)
j = torch.arange(nu, dtype=torch.int64, device=dev)
U = torch.cumsum(1 + (bench.splitmix64_torch(j ^ bench._i64(bench.SEED)) & ((1 << 32) - 1)), 0)
thr = int(0.9 * (1 << 20))
core = ((bench.splitmix64_torch(j ^ bench._i64(bench.SEED + 77)) >> 11) & ((1 << 20) - 1)) < int(0.3 * (1 << 20))
files, taxs = [], []
for f in range(nfiles):
    h = bench.splitmix64_torch(j ^ bench._i64(bench.SEED + 1000 * (f + 1)))
    m = (((h >> 11) & ((1 << 20) - 1)) < thr) | core
    k = U[m]
    if f == 0:
        k = torch.cat([k, U[-1] + 1 + torch.arange(per // 10, dtype=torch.int64, device=dev) * 3])
    files.append(k)
    if os.environ.get("C4_TAX") == "file":
        taxs.append(torch.full((k.numel(),), T - 8 ** 7 + 1 + (f * 7919) % (8 ** 7), dtype=torch.int32, device=dev))
    else:
        taxs.append((1 + (bench.splitmix64_torch(k ^ bench._i64(bench.SEED + 2 + f)) & ((1 << 40) - 1)) % T).to(torch.int32))
out = torch.empty(files[0].numel() + 8, dtype=torch.int64, device=dev)
outt = torch.empty(files[0].numel() + 8, dtype=torch.int32, device=dev)
for name, fn in (("inter+tax", lambda: ctx.inter(files, taxs, out=out, out_taxids=outt)), ("diff+tax", lambda: ctx.diff(files, taxs, out=out, out_taxids=outt)),
                 ("diff -t", lambda: ctx.diff(files, taxs, compare_taxid=True, out=out, out_taxids=outt)), ("diff plain", lambda: ctx.diff(files, out=out))):
    fn(); torch.cuda.synchronize()
    ts, ks, ls = [], [], []
    for _ in range(5):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        ks.append(ctx.last_kernel_ms()); ls.append(ctx.last_call_ms())
    print("%-10s wall %.3f ms   device work of the call %.3f ms   probe kernel %.3f ms" % (name, min(ts), min(ls), min(ks)), flush=True)
