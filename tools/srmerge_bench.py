"""many-stream merge / union call times (and UKM_SRMERGE_DEBUG phases on stderr) over config-4-shaped files.
usage: python tools/srmerge_bench.py NFILES PER_FILE P [tax] [merge|union|both|common] [reps]
(common: threshold = half the expected copies of a code, NFILES * P / 2, at least 2)"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from unikmer_amd import lib
from conftest import synth_tree
dev = torch.device("cuda", 0)
ctx = lib.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
nfiles, per, P = int(sys.argv[1]), int(float(sys.argv[2])), float(sys.argv[3])
tax = "tax" in sys.argv[4:]
what = "union" if "union" in sys.argv[4:] else ("both" if "both" in sys.argv[4:] else ("common" if "common" in sys.argv[4:] else "merge"))
reps = int(sys.argv[-1]) if sys.argv[-1].isdigit() and len(sys.argv) > 4 else 3
child, parent = synth_tree(7, 8); ctx.taxonomy_load(child, parent); T = len(child)
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
    if tax:
        taxs.append((1 + (bench.splitmix64_torch(k ^ bench._i64(bench.SEED + 2 + f)) & ((1 << 40) - 1)) % T).to(torch.int32))
del j, gaps, U
total = sum(x.numel() for x in files)
if os.environ.get("SRB_ONEBUF"):   # all files as views of ONE allocation (TLB experiment)
    big = torch.cat(files); off = 0; views = []
    for x in files:
        views.append(big[off:off + x.numel()]); off += x.numel()
    files = views
    if tax:
        bigt = torch.cat(taxs); off = 0; tv = []
        for x in taxs:
            tv.append(bigt[off:off + x.numel()]); off += x.numel()
        taxs = tv
    torch.cuda.empty_cache()
ok = torch.empty(total + 8, dtype=torch.int64, device=dev)
ot = torch.empty(total + 8, dtype=torch.int32, device=dev) if tax else None
tx = taxs if tax else None
kw = dict(out=ok, out_taxids=ot) if tax else dict(out=ok)
def wall(fn):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return ["%.1f" % t for t in ts]
print("files", nfiles, "total", total, "P", P, "tax", tax, flush=True)
if what in ("merge", "both"):
    print("merge_k", wall(lambda: ctx.merge_k(files, tx, **kw)), "route", ctx.last_route(), flush=True)
    if os.environ.get("SRB_CHECK"):
        g = ctx.merge_k(files, tx, **kw)
        gk = g[0] if tax else g
        want = torch.sort(torch.cat(files)).values   # codes < 2^62: signed order = unsigned order
        print("merge_k keys equal torch.sort of the concatenation:", bool(torch.equal(gk, want)), flush=True)
        del want
if what in ("union", "both"):
    print("union  ", wall(lambda: ctx.union(files, tx, **kw)), "route", ctx.last_route(), flush=True)
if what == "common":
    cthr = max(2, int(nfiles * P / 2))
    r = [None]
    def f():
        r[0] = ctx.common(files, cthr, tx, **kw)
    print("common threshold", cthr, wall(f), "route", ctx.last_route(), "out", (r[0][0] if tax else r[0]).numel(), flush=True)
